// list_icp.hpp -- ICP on point LISTS (height == 1 handles): the reference's actual operating point.
//
// readimage (src/GraphicEnd.cpp:279-295) turns a frame into an UNORGANIZED cloud of ~15 k points (PassThrough + VoxelGrid 0.03) and
// hands THAT on (:158).  Round 5 aligned such lists with the full N x M scan on the matrix cores: three dependent launches per
// iteration (28 + 5 + 13 us), 0.24 G pairs each.  A list has no image tiles to prune with -- but it has space: this file
//   (1) sorts both lists into the cells of a fixed 64 x 64 x 32 grid over the sensor's frustum box, cells in Morton order (counting
//       sort: atomic ranks, a two-level prefix sum, a scatter; the order INSIDE a cell is whatever the atomics dealt, no result
//       depends on it), and cuts the sorted list into TILES of at most 64 points (one wavefront) that never leave a 2 x 2 x 2 block
//       of cells (a "super-cell", 0.44 m at z_filter 7 m): a tile's axis-aligned box, taken from the data, is then a few decimetres
//       wide.  (First build: every 64 consecutive sorted points -- tiles that straddled a jump of the Morton curve were metres wide,
//       their owners tested and scanned most of the target, and the slowest block set the iteration: 61 us.)
//   (2) runs ALL iterations of a run in ONE persistent launch: a block owns source tiles; per iteration it transforms its 64 points,
//       bounds every query by the distance to its previous match, tests the boxes of ALL target tiles (64 per step) against the
//       wave's box, scans the surviving tiles exhaustively with the canonical distance and the (d2 bits << 32 | index) key minimum
//       -- the four waves of a block share the candidate tiles of one source tile and merge through ds_min_u64 --, forms the
//       integer row vectors, Gram-accumulates them on the fp64 matrix cores (tile_accumulate) and meets the other blocks at a grid
//       barrier; every block then solves the SAME totals for itself (no pose broadcast, one barrier per iteration).
// Exactness: a candidate can win or tie only if its tile's box is within sqrt(U) of the query, U >= d2(NN) being the distance to a
// target that exists (or the gate); only such tiles are skipped.  Indices, d2, sums and poses are bit-identical to the oracle's
// (tests/test_unorganized.py), for any cell size, any order inside a cell and any grid width.
//
// Co-residency: the grid barrier needs every block of a launch resident.  A launch has at most LS_MAX_BLOCKS = 256 blocks of 256
// threads (<= 128 VGPRs, amdgpu_waves_per_eu(4, 4), ~4 KB of LDS: four blocks fit a CU, 1,024 the chip), so the four launches the
// runtime's four hardware queues can run side by side are always all resident -- two persistent launches can never starve each
// other of slots.  tests/test_isa_regressions.py holds the register budget this rests on.
#pragma once

namespace s3d {

constexpr int LS_NCELL = 1 << 17;                       // 64 x 64 x 32 cells (x, y in [-zmax, zmax], z in (0, zmax]), Morton order
constexpr int LS_GROUP = 128;                           // cells per group of the two-level prefix sum
constexpr int LS_NGROUP = LS_NCELL / LS_GROUP;          // 1,024
constexpr int LS_TASKS = 16;                            // (frame, role) lists sorted per launch sequence
constexpr int LS_WAVES = 4;                             // waves per block of the persistent kernel: they share a source tile's candidates
constexpr int LS_MAX_BLOCKS = 256;                      // blocks per persistent launch (see "Co-residency" above)

// Pointers that come out of the pair table are generic to the compiler: a load through them is a flat_load + s_waitcnt vmcnt(0) -- one memory
// round trip per CANDIDATE in the scan loop (first build: 75 us per iteration).  The sorted target list and its boxes are written by
// earlier launches only, and every address in the scan is wave-uniform: read through the constant address space they become s_load_dwordx4/x16.
typedef float ls_f4 __attribute__((ext_vector_type(4)));
typedef const ls_f4 __attribute__((address_space(4))) *ls_cptr4;
typedef const ls_f4 __attribute__((address_space(1))) *ls_gptr4;
typedef int __attribute__((address_space(1))) *ls_gptri;
typedef int ls_i2 __attribute__((ext_vector_type(2)));
typedef const ls_i2 __attribute__((address_space(4))) *ls_cptri2;
__device__ __forceinline__ float4 ls_ld(ls_cptr4 p, int k) { const ls_f4 v = p[k]; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 ls_ld(ls_gptr4 p, int k) { const ls_f4 v = p[k]; return make_float4(v.x, v.y, v.z, v.w); }

constexpr int LS_NSUPER = LS_NCELL / 8;                 // super-cells (2 x 2 x 2 cells: eight consecutive Morton codes)
// A tile load the compiler does not know to be one: issued here, waited for by hand (ls_wait).  hipcc puts s_waitcnt vmcnt(0) in front
// of the first use of ANY loaded register inside these loops, so a load issued one tile ahead was waited for at once -- one load latency
// per scanned tile (2.5 us per tile; stamps in profiles/r06_list_by_iteration.md).  Loads return in order: with N younger loads in
// flight, vmcnt(N) is exactly "mine has arrived"; whatever else the compiler has in flight only makes the wait longer, never shorter.
__device__ __forceinline__ void ls_issue(ls_f4 &r, ls_gptr4 p)
{
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
}
template <int N> __device__ __forceinline__ void ls_wait(ls_f4 &r)
{
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N) : "memory");
}

struct ListTask {
    const float4 *cloud;           // the frame's cloud (N records, invalid = NaN)
    const float4 *nrm;             // its normals (target role with use_normals: only points with a normal are targets)
    float4 *pts;                   // out: the sorted list, (x, y, z, original index), followed by 64 points at infinity
    int2 *tile;                    // out: tile t = sorted points [tile[t].x, tile[t].x + tile[t].y), y <= 64; in no particular order
    float4 *box;                   // out: [2 t] = lo, [2 t + 1] = hi of tile t
    int *n;                        // out: [0] number of points in the list, [1] number of tiles
    int which, use_normals, i_begin, i_end;
};
struct ListTasks { ListTask t[LS_TASKS]; };

__device__ __forceinline__ int ls_cell(float x, float y, float z, float zmax)
{
    const float s = 32.0f / zmax;
    int ix = (int)floorf((x + zmax) * s), iy = (int)floorf((y + zmax) * s), iz = (int)floorf(z * s);
    ix = min(max(ix, 0), 63); iy = min(max(iy, 0), 63); iz = min(max(iz, 0), 31);
    unsigned int c = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k)
        c |= (((unsigned int)ix >> k) & 1u) << (3 * k) | (((unsigned int)iy >> k) & 1u) << (3 * k + 1) | (((unsigned int)iz >> k) & 1u) << (3 * k + 2);
    c |= (((unsigned int)ix >> 5) & 1u) << 15 | (((unsigned int)iy >> 5) & 1u) << 16;
    return (int)c;
}

// (1a) grid (ceil(N / 256), ntasks): every kept point takes a rank in its cell.  cnt / grp are zero on entry (self-cleaning: k_list_scan
// zeroes the cells it read, k_list_boxes the groups).
__global__ __launch_bounds__(256) void k_list_bin(ListTasks a, int *__restrict__ cnt, int *__restrict__ grp, int2 *__restrict__ cr, int N, float zmax)
{
    const int t = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const ListTask &T = a.t[t];
    float4 q;
    const bool ok = i >= T.i_begin && compact_keep(T.cloud, T.nrm, i, T.i_end, T.which, T.use_normals, zmax, q);
    int2 o = make_int2(-1, 0);
    if (ok) {
        const int c = ls_cell(q.x, q.y, q.z, zmax);
        o = make_int2(c, atomicAdd(cnt + (size_t)t * LS_NCELL + c, 1));
        atomicAdd(grp + t * LS_NGROUP + (c >> 7), 1);
    }
    cr[(size_t)t * N + i] = o;
}

// (1b) grid (LS_NGROUP, ntasks), block 64: the start of every cell of a non-empty group = points in the groups in front (each wave
// sums them itself: <= 1,024 integers) + points in the group's cells in front; the counts it read are zeroed for the next run.
__global__ __launch_bounds__(64) void k_list_scan(ListTasks a, int *__restrict__ cnt, int *__restrict__ cstart, const int *__restrict__ grp, int2 *__restrict__ super)
{
    const int g = blockIdx.x, t = blockIdx.y, lane = threadIdx.x;
    const int *__restrict__ G = grp + t * LS_NGROUP;
    const int mine = G[g];
    if (mine == 0 && g != 0) return;
    int pre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < LS_NGROUP / 64 / 4; ++k) {
        const int4 v = reinterpret_cast<const int4 *>(G)[lane * (LS_NGROUP / 64 / 4) + k];
        const int i0 = (lane * (LS_NGROUP / 64 / 4) + k) * 4;
        tot += (v.x + v.y) + (v.z + v.w);
        pre += (i0 < g ? v.x : 0) + (i0 + 1 < g ? v.y : 0) + (i0 + 2 < g ? v.z : 0) + (i0 + 3 < g ? v.w : 0);
    }
    for (int o = 32; o >= 1; o >>= 1) { pre += __shfl_xor(pre, o); tot += __shfl_xor(tot, o); }
    if (g == 0) {
        if (lane == 0) { a.t[t].n[0] = tot; a.t[t].n[1] = 0; }
        const float inf = __int_as_float(0x7f800000);
        a.t[t].pts[tot + lane] = make_float4(inf, inf, inf, __int_as_float(-1));      // a scan may run up to 63 records past its tile: beyond the list it meets these
    }
    if (mine == 0) return;
    int2 *__restrict__ c2p = reinterpret_cast<int2 *>(cnt + (size_t)t * LS_NCELL + g * LS_GROUP) + lane;
    const int2 c2 = *c2p;
    const int s = c2.x + c2.y;
    int inc = s;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
    const int st = pre + inc - s;
    reinterpret_cast<int2 *>(cstart + (size_t)t * LS_NCELL + g * LS_GROUP)[lane] = make_int2(st, st + c2.x);
    *c2p = make_int2(0, 0);
    // super-cell = 8 cells = 4 lanes: (start, count)
    const int st0 = __shfl(st, lane & ~3), end3 = __shfl(pre + inc, lane | 3);
    if ((lane & 3) == 0) super[(size_t)t * LS_NSUPER + g * (LS_GROUP / 8) + (lane >> 2)] = make_int2(st0, end3 - st0);
}

// (1c) grid (ceil(N / 256), ntasks)
__global__ __launch_bounds__(256) void k_list_scatter(ListTasks a, const int *__restrict__ cstart, const int2 *__restrict__ super, const int2 *__restrict__ cr, int N)
{
    const int t = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int2 o = cr[(size_t)t * N + i];
    if (o.x < 0) return;
    const float4 q = a.t[t].cloud[i];
    const int pos = cstart[(size_t)t * LS_NCELL + o.x] + o.y;
    a.t[t].pts[pos] = make_float4(q.x, q.y, q.z, __int_as_float(i));
    // the point that opens a tile (every 64th of its super-cell's run) enters it in the tile table
    const int2 sc = super[(size_t)t * LS_NSUPER + (o.x >> 3)];
    const int rk = pos - sc.x;
    if ((rk & 63) == 0) a.t[t].tile[atomicAdd(a.t[t].n + 1, 1)] = make_int2(pos, min(64, sc.y - rk));
}

// (1d) grid (ceil(tile capacity / 4), ntasks), block 256: one wave per tile -- its box from the data.  Zeroes the group counts for the next run.
__global__ __launch_bounds__(256) void k_list_boxes(ListTasks a, int *__restrict__ grp)
{
    const int t = blockIdx.y, lane = threadIdx.x & 63, tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int z = blockIdx.x * 256 + threadIdx.x;
    if (z < LS_NGROUP) grp[t * LS_NGROUP + z] = 0;
    const ListTask &T = a.t[t];
    if (tile >= T.n[1]) return;
    const int2 td = T.tile[tile];
    const float inf = __int_as_float(0x7f800000);
    const bool v = lane < td.y;
    float4 p = make_float4(inf, inf, inf, 0.0f);
    if (v) p = T.pts[td.x + lane];
    const float lx = wave_min(v ? p.x : inf), ly = wave_min(v ? p.y : inf), lz = wave_min(v ? p.z : inf);
    const float hx = wave_max(v ? p.x : -inf), hy = wave_max(v ? p.y : -inf), hz = wave_max(v ? p.z : -inf);
    if (lane == 0) { T.box[2 * tile] = make_float4(lx, ly, lz, 0.0f); T.box[2 * tile + 1] = make_float4(hx, hy, hz, 0.0f); }
}

// (2) the persistent launch: grid (G <= LS_MAX_BLOCKS / B, B), block 64 x LS_WAVES.  `ticket[b]` (zeroed by k_pair_init) is the pair's
// barrier counter: iteration `it` is complete when it reads (it + 1) * G.
//
// Everything here is bound by LATENCY -- a dependent access to L2 costs 1-2 us, and the first build paid a dozen of them per iteration
// (61 us; stamps of block 0: search 25, barrier 13, solve 18).  So:
//   * the target's tile boxes and tile table live in LDS for the whole launch (up to LS_LDS_TILES tiles; larger lists read them from L2);
//   * a block's FIRST source tile stays in registers, and its previous matches (point + index) in LDS, across the iterations;
//   * a wave first collects the candidate tiles that pass its lanes' box tests (LDS only), then streams them: the points of tile k + 1 are
//     in flight while tile k is scanned out of a 1 KB LDS stage (one coalesced load per tile instead of four dependent scalar trips);
//   * the previous match's coordinates are stored, not gathered again.
// The svd estimator's step (solve_step_one(sums, T, 1): the same operations in the same order, so the same bits) with every array in
// LDS: inlined into the persistent kernel the register form (H, g, v, dR, T: ~60 doubles) pushed the loop-carried values of the
// search out to scratch -- 33 spilled VGPRs whose reloads (one L2 trip each) cost ~10 us at the head of every iteration.
// ws: 48 doubles of LDS; T: the pose (LDS), updated in place.  One lane.  Returns 1 (updated) or 0 (fewer than 3 correspondences).
__device__ __noinline__ int list_solve_svd(const double *sums, double *T, double *ws)
{
    const double n = sums[27];
    if (n < 3.0) return 0;
    double *g = ws, *v = ws + 9, *R = ws + 18, *pq = ws + 27;        // g[r * 3 + c], v[r * 3 + c], R (dR), pm[3] qm[3] dt[3], Tn[12]
    for (int k = 0; k < 3; ++k) { pq[k] = sums[k] / n; pq[3 + k] = sums[3 + k] / n; }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { g[r * 3 + c] = sums[6 + r * 3 + c] - (n * pq[r]) * pq[3 + c]; v[r * 3 + c] = r == c ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 12; ++sweep) {
        bool rotated = false;
        for (int k = 0; k < 3; ++k) {
            const int p = k == 2 ? 1 : 0, q = k == 0 ? 1 : 2;
            const double g0p = g[p], g1p = g[3 + p], g2p = g[6 + p], g0q = g[q], g1q = g[3 + q], g2q = g[6 + q];
            const double al = (g0p * g0p + g1p * g1p) + g2p * g2p;
            const double be = (g0q * g0q + g1q * g1q) + g2q * g2q;
            const double ga = (g0p * g0q + g1p * g1q) + g2p * g2q;
            if (ga * ga <= 0x1p-100 * (al * be)) continue;
            rotated = true;
            const double zeta = (be - al) / (2.0 * ga);
            double t = 1.0 / (fabs(zeta) + sqrt(zeta * zeta + 1.0));
            if (zeta < 0.0) t = -t;
            const double c = 1.0 / sqrt(t * t + 1.0);
            const double s = c * t;
            for (int m = 0; m < 3; ++m) {
                const double gp = g[m * 3 + p], gq = g[m * 3 + q];
                g[m * 3 + p] = c * gp - s * gq;
                g[m * 3 + q] = s * gp + c * gq;
                const double vp = v[m * 3 + p], vq = v[m * 3 + q];
                v[m * 3 + p] = c * vp - s * vq;
                v[m * 3 + q] = s * vp + c * vq;
            }
        }
        if (!rotated) break;
    }
    double sg[3];
    for (int k = 0; k < 3; ++k) sg[k] = sqrt((g[k] * g[k] + g[3 + k] * g[3 + k]) + g[6 + k] * g[6 + k]);
    int i0 = 0, i1 = 1, i2 = 2, tmp;
    if (sg[i1] > sg[i0]) { tmp = i0; i0 = i1; i1 = tmp; }
    if (sg[i2] > sg[i1]) { tmp = i1; i1 = i2; i2 = tmp; }
    if (sg[i1] > sg[i0]) { tmp = i0; i0 = i1; i1 = tmp; }
    for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if ((sg[i0] > 0.0) && (sg[i1] > 1e-14 * sg[i0])) {
        const double s0 = sg[i0], s1 = sg[i1];
        const double u00 = g[i0] / s0, u01 = g[3 + i0] / s0, u02 = g[6 + i0] / s0;
        const double u10 = g[i1] / s1, u11 = g[3 + i1] / s1, u12 = g[6 + i1] / s1;
        const double v00 = v[i0], v01 = v[3 + i0], v02 = v[6 + i0], v10 = v[i1], v11 = v[3 + i1], v12 = v[6 + i1];
        const double u20 = u01 * u12 - u02 * u11, u21 = u02 * u10 - u00 * u12, u22 = u00 * u11 - u01 * u10;
        const double v20 = v01 * v12 - v02 * v11, v21 = v02 * v10 - v00 * v12, v22 = v00 * v11 - v01 * v10;
        R[0] = (v00 * u00 + v10 * u10) + v20 * u20; R[1] = (v00 * u01 + v10 * u11) + v20 * u21; R[2] = (v00 * u02 + v10 * u12) + v20 * u22;
        R[3] = (v01 * u00 + v11 * u10) + v21 * u20; R[4] = (v01 * u01 + v11 * u11) + v21 * u21; R[5] = (v01 * u02 + v11 * u12) + v21 * u22;
        R[6] = (v02 * u00 + v12 * u10) + v22 * u20; R[7] = (v02 * u01 + v12 * u11) + v22 * u21; R[8] = (v02 * u02 + v12 * u12) + v22 * u22;
    }
    for (int r = 0; r < 3; ++r) pq[6 + r] = pq[3 + r] - ((R[r * 3 + 0] * pq[0] + R[r * 3 + 1] * pq[1]) + R[r * 3 + 2] * pq[2]);
    double *Tn = ws + 36;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Tn[r * 4 + c] = (R[r * 3 + 0] * T[0 * 4 + c] + R[r * 3 + 1] * T[1 * 4 + c]) + R[r * 3 + 2] * T[2 * 4 + c];
        Tn[r * 4 + 3] = ((R[r * 3 + 0] * T[3] + R[r * 3 + 1] * T[7]) + R[r * 3 + 2] * T[11]) + pq[6 + r];
    }
    for (int k = 0; k < 12; ++k) T[k] = Tn[k];
    T[12] = 0.0; T[13] = 0.0; T[14] = 0.0; T[15] = 1.0;
    return 1;
}

// (out of line for the same reason: the lane-parallel LDL^T keeps ~40 doubles live)
__device__ __noinline__ double list_solve_p2p(const double *tot, const double *sh, int *rc_lds)
{
    int rc;
    const double Tn = wave_solve_point2plane(tot, sh, rc);
    if ((threadIdx.x & 63) == 0) *rc_lds = rc;
    return Tn;
}

// Block barrier for data exchanged through LDS only.  __syncthreads() is a fence over ALL address spaces: it waits for every
// outstanding global store and atomic of the wave (vmcnt(0)) -- a memory round trip at each of the eight barriers of an iteration.
__device__ __forceinline__ void ls_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int LS_LDS_TILES = 384;                       // target tiles whose boxes (32 B) and table entries (8 B) fit the block's LDS budget
constexpr int LS_CAND = 32;                             // candidate tiles a wave lists before it scans them

template <int EST, bool GATED, bool DBG = false>
__global__ __launch_bounds__(64 * LS_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_list_icp(
    const PairPtrs *__restrict__ pairs, Geometry g, int iters, int n_coarse, int nsets, int N,
    double *__restrict__ Tcur, double *__restrict__ trace_T, double *__restrict__ trace_S, int *__restrict__ flags,
    long long *__restrict__ acc, unsigned int *__restrict__ ticket, float4 *__restrict__ prev /* [B][N]: previous match (x, y, z, index) by sorted source position */,
    int *__restrict__ corr, float *__restrict__ cd2, int *__restrict__ corr_trace /* nullable: [iters][maxB][nslots] */, int maxB, int nslots,
    double *__restrict__ res_host, int *__restrict__ end_run,
    long long *__restrict__ dbg /* nullable (SLAM3D_LIST_DEBUG=1): [iters][G][12] per block, thread 0: 100 MHz ticks spent in bounds, listing, scans, rows, Gram, arrive, barrier wait, totals, derive, solve; tiles wave 0 scanned; iteration start */)
{
    __shared__ double Tsh[16], tot[32];
    __shared__ long long Gs[NRAW];
    __shared__ double slab[256];
    __shared__ unsigned long long skey[64];
    __shared__ int s_rc;
    __shared__ long long ph[12];       // (DBG) thread 0's phase times
    __shared__ float4 s_box[2 * LS_LDS_TILES];
    __shared__ int2 s_tile[LS_LDS_TILES];
    __shared__ float4 s_stage[LS_WAVES][64];
    __shared__ float4 s_prev[2][64];
    __shared__ int s_cand[LS_WAVES][LS_CAND];
    const int b = blockIdx.y, G = gridDim.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const PairPtrs &pp = pairs[b];
    const int ns = __builtin_amdgcn_readfirstlane(pp.ls_ns[0]), nt = __builtin_amdgcn_readfirstlane(pp.ls_nt[0]);
    const int nst = __builtin_amdgcn_readfirstlane(pp.ls_ns[1]), ntt = __builtin_amdgcn_readfirstlane(pp.ls_nt[1]);
    const ls_gptr4 spts = (ls_gptr4)pp.ls_src, tpts = (ls_gptr4)pp.ls_tgt, tboxg = (ls_gptr4)pp.ls_tbox;
    const ls_cptri2 stile = (ls_cptri2)pp.ls_stile;
    const ls_i2 __attribute__((address_space(1))) *ttileg = (const ls_i2 __attribute__((address_space(1))) *)pp.ls_ttile;
    float4 *__restrict__ gprev = prev + (size_t)b * N;
    const bool lds_boxes = ntt <= LS_LDS_TILES;
    if (lds_boxes) {
        for (int k = tid; k < 2 * ntt; k += 64 * LS_WAVES) s_box[k] = ls_ld(tboxg, k);
        for (int k = tid; k < ntt; k += 64 * LS_WAVES) { const ls_i2 v = ttileg[k]; s_tile[k] = make_int2(v.x, v.y); }
    }
    if (tid < 16) Tsh[tid] = Tcur[b * 16 + tid];
    // the block's first TWO source tiles (a 16 k-point list has ~330 tiles for 256 blocks): resident in registers
    float4 r_s4a = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1)), r_s4b = r_s4a;
    int r_starta = 0, r_cnta = 0, r_startb = 0, r_cntb = 0;
    if ((int)blockIdx.x < nst) {
        const ls_i2 sd = stile[blockIdx.x];
        r_starta = sd.x; r_cnta = sd.y;
        if (lane < r_cnta) r_s4a = ls_ld(spts, r_starta + lane);
    }
    if ((int)blockIdx.x + G < nst) {
        const ls_i2 sd = stile[blockIdx.x + G];
        r_startb = sd.x; r_cntb = sd.y;
        if (lane < r_cntb) r_s4b = ls_ld(spts, r_startb + lane);
    }
    int flag = 0;
    const float inf = __int_as_float(0x7f800000);
    const unsigned long long key_gate = ((unsigned long long)(unsigned int)__float_as_int(g.gate2) << 32) | 0xffffffffull;
    auto box_of = [&](int t, float4 &lo, float4 &hi) __attribute__((always_inline)) {
        if (lds_boxes) { lo = s_box[2 * t]; hi = s_box[2 * t + 1]; }
        else { lo = ls_ld(tboxg, 2 * t); hi = ls_ld(tboxg, 2 * t + 1); }
    };
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        const bool coarse = it < n_coarse;
        long long ph_prev = 0;
        const bool stamp = DBG && dbg && b == 0 && tid == 0;          // SLAM3D_LIST_DEBUG: thread 0 of EVERY block books its phases
        if (stamp) for (int k = 0; k < 12; ++k) ph[k] = 0;
        if (stamp) { ph_prev = (long long)wall_clock64(); ph[11] = ph_prev; }
        int n_scanned = 0;
        Rt m;
        m.r00 = uni_f((float)Tsh[0]); m.r01 = uni_f((float)Tsh[1]); m.r02 = uni_f((float)Tsh[2]);  m.t0 = uni_f((float)Tsh[3]);
        m.r10 = uni_f((float)Tsh[4]); m.r11 = uni_f((float)Tsh[5]); m.r12 = uni_f((float)Tsh[6]);  m.t1 = uni_f((float)Tsh[7]);
        m.r20 = uni_f((float)Tsh[8]); m.r21 = uni_f((float)Tsh[9]); m.r22 = uni_f((float)Tsh[10]); m.t2 = uni_f((float)Tsh[11]);
        long long *__restrict__ set = acc + (((size_t)b * nsets + it) * ACC_R + (blockIdx.x & (ACC_R - 1))) * ACC_STRIDE;
        const bool last = it == iters - 1;
        for (int tile = blockIdx.x; tile < nst; tile += G) {
            const bool res_a = tile == (int)blockIdx.x, res_b = tile == (int)blockIdx.x + G, resident = res_a || res_b;
            int start = res_a ? r_starta : r_startb, cnt = res_a ? r_cnta : r_cntb;
            float4 s4 = res_a ? r_s4a : r_s4b;
            if (!resident) {
                const ls_i2 sd = stile[tile];                 // (wave-uniform)
                start = sd.x; cnt = sd.y;
                s4 = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
                if (lane < cnt) s4 = ls_ld(spts, start + lane);
            }
            const int idx = start + lane;
            const bool valid = lane < cnt;
            const int i = __float_as_int(s4.w);
            const bool active = valid && !(coarse && ((i >> 3) & 3) != 0);       // spec S4c on a list: the points of every fourth group of eight
            float px, py, pz;
            xform(m, s4.x, s4.y, s4.z, px, py, pz);
            unsigned long long key = key_gate;
            if (active && it > 0) {
                const float4 pv = resident ? s_prev[res_a ? 0 : 1][lane] : gprev[idx];
                const int jprev = __float_as_int(pv.w);
                if (jprev >= 0) {
                    const float d2 = canon_d2(px, py, pz, pv.x, pv.y, pv.z);
                    if (d2 <= g.gate2) key = ((unsigned long long)(unsigned int)__float_as_int(d2) << 32) | (unsigned int)jprev;
                }
            }
            const unsigned long long key0 = key;
            float U = active ? __int_as_float((int)(unsigned int)(key >> 32)) : -1.0f;       // an inactive lane passes no box test
            const float lx = wave_min(active ? px : inf), ly = wave_min(active ? py : inf), lz = wave_min(active ? pz : inf);
            const float hx = wave_max(active ? px : -inf), hy = wave_max(active ? py : -inf), hz = wave_max(active ? pz : -inf);
            const float Umax = wave_max(U);
            if (w == 0) skey[lane] = key;
            ls_barrier();
            if (stamp) { const long long now_ = (long long)wall_clock64(); ph[0] += now_ - ph_prev; ph_prev = now_; }
            if (Umax >= 0.0f) {
                // ---- the candidate tiles of this wave: every LS_WAVES-th tile whose box is within sqrt(Umax) of the wave's box and
                // within sqrt(U) of one of its lanes; scanned as soon as LS_CAND are listed, and at the end
                int dealt = 0, nc = 0;                                          // wave-uniform
                // the listed tiles, FOUR loads in flight: a tile's points arrive ~2.5 us after they are asked for and are scanned in 0.7 us
                // (with one tile ahead a wave that scans nine tiles took 27 us: one load latency per tile)
                // the listed tiles, three loads in flight (registers A, B, C in turn -- no rotation: a register must not be copied while
                // its load is in flight): the points of tiles k + 1 and k + 2 arrive while tile k is scanned out of the wave's 1 KB LDS stage
                auto issue = [&](int kk, ls_f4 &r, int &n_) __attribute__((always_inline)) {
                    const int t_ = __builtin_amdgcn_readfirstlane(s_cand[w][kk < nc ? kk : 0]);     // (past the end: the first tile again, count 0)
                    const int2 td = lds_boxes ? s_tile[t_] : make_int2(ttileg[t_].x, ttileg[t_].y);
                    n_ = kk < nc ? __builtin_amdgcn_readfirstlane(td.y) : 0;
                    ls_issue(r, tpts + (td.x + lane));            // all 64 lanes: behind a short tile lie the next tile's points or the padding
                };
                auto scan_one = [&](const ls_f4 &r, int n_) __attribute__((always_inline)) {
                    s_stage[w][lane] = lane < n_ ? make_float4(r.x, r.y, r.z, r.w) : make_float4(inf, inf, inf, __int_as_float(-1));
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifndef LS_SCAN_EXP
#define LS_SCAN_EXP 0
#endif
#if LS_SCAN_EXP == 0
                    for (int c0 = 0; c0 < n_; c0 += 8) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const float4 q = s_stage[w][c0 + c];          // (broadcast read; entries behind the tile's count are at infinity)
                            const float d2 = canon_d2(px, py, pz, q.x, q.y, q.z);
                            key = key_min(key, ((unsigned long long)(unsigned int)__float_as_int(d2) << 32) | (unsigned int)__float_as_int(q.w));
                        }
                    }
#elif LS_SCAN_EXP == 1      // timing experiment: LDS reads + float minimum only (results wrong)
                    float dm = __int_as_float((int)(unsigned int)(key >> 32));
                    for (int c0 = 0; c0 < n_; c0 += 8) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const float4 q = s_stage[w][c0 + c];
                            dm = fminf(dm, canon_d2(px, py, pz, q.x, q.y, q.z));
                        }
                    }
                    key = ((unsigned long long)(unsigned int)__float_as_int(dm) << 32) | (unsigned int)key;
#elif LS_SCAN_EXP == 2      // lane broadcast from the register instead of LDS, keyed minimum
                    for (int c0 = 0; c0 < n_; c0 += 8) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.x), c0 + c)), qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.y), c0 + c));
                            const float qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.z), c0 + c));
                            const unsigned int qj = (unsigned int)__builtin_amdgcn_readlane(__float_as_int(r.w), c0 + c);
                            const float d2 = canon_d2(px, py, pz, qx, qy, qz);
                            key = key_min(key, ((unsigned long long)(unsigned int)__float_as_int(d2) << 32) | qj);
                        }
                    }
#elif LS_SCAN_EXP == 3      // timing experiment: no scan at all
#endif
                    __builtin_amdgcn_wave_barrier();                   // (the stage is rewritten by the next tile)
                };
                auto scan_listed = [&]() __attribute__((always_inline)) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    ls_f4 rA, rB, rC;
                    int nA, nB, nC;
                    issue(0, rA, nA); issue(1, rB, nB);
                    for (int k = 0; k < nc; k += 3) {
                        issue(k + 2, rC, nC); ls_wait<2>(rA); scan_one(rA, nA);
                        issue(k + 3, rA, nA); ls_wait<2>(rB); scan_one(rB, nB);
                        issue(k + 4, rB, nB); ls_wait<2>(rC); scan_one(rC, nC);
                    }
                    ls_wait<0>(rA); ls_wait<0>(rB);                    // (the two loads still in flight write registers: let them land)
                    U = active ? __int_as_float((int)(unsigned int)(key >> 32)) : -1.0f;
                    n_scanned += nc;
                    nc = 0;
                };
                for (int base = 0; base < ntt; base += 64) {
                    const int tt = base + lane;
                    float gap = inf;
                    if (tt < ntt) { float4 lo, hi; box_of(tt, lo, hi); gap = box_gap2(lo, hi, lx, ly, lz, hx, hy, hz); }
                    unsigned long long mask = __ballot(gap <= Umax);
                    while (mask != 0ull) {
                        const int bit = __builtin_ctzll(mask);
                        mask &= mask - 1ull;
                        const bool mine = (dealt & (LS_WAVES - 1)) == w;
                        dealt += 1;
                        if (!mine) continue;
                        const int t = __builtin_amdgcn_readfirstlane(base + bit);
                        float4 lo, hi;
                        box_of(t, lo, hi);
                        const float gp = box_gap2(lo, hi, px, py, pz, px, py, pz);
                        if (__ballot(gp <= U) == 0ull) continue;
                        if (lane == 0) s_cand[w][nc] = t;
                        nc += 1;
                        if (nc == LS_CAND) scan_listed();
                    }
                }
                if (stamp) { const long long now_ = (long long)wall_clock64(); ph[1] += now_ - ph_prev; ph_prev = now_; }
                if (nc > 0) scan_listed();
                if (key != key0) atomicMin(&skey[lane], key);
            }
            ls_barrier();
            if (stamp) { const long long now_ = (long long)wall_clock64(); ph[2] += now_ - ph_prev; ph_prev = now_; }
            if (w == 0) {
                key = skey[lane];
                RowBasis rb;
                float4 pq = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
                const int slot = ((max(i, 0) >> 3) << 6) + (max(i, 0) & 7);
                int *cp = corr + (size_t)b * nslots + slot;
                float *dp = cd2 + (size_t)b * nslots + slot;
                if constexpr (GATED) {
                    SlotGates sg;
                    sg.resid2 = g.resid2; sg.min_ncos = g.min_ncos; sg.snrm = pp.snrm; sg.spix = max(i, 0);
                    sg.assoc = g.pair_gate ? pp.assoc : nullptr;
                    sg.r[0] = m.r00; sg.r[1] = m.r01; sg.r[2] = m.r02; sg.r[3] = m.r10; sg.r[4] = m.r11; sg.r[5] = m.r12;
                    sg.r[6] = m.r20; sg.r[7] = m.r21; sg.r[8] = m.r22;
                    finish_slot<true>(active, key, px, py, pz, pp.tgt, pp.nrm, g.gate2, EST, g.b_scale, cp, dp, &pq, rb, last && valid, -2, &sg);
                } else {
                    finish_slot<false>(active, key, px, py, pz, pp.tgt, pp.nrm, g.gate2, EST, g.b_scale, cp, dp, &pq, rb, last && valid, -2, nullptr);
                }
                if (corr_trace && valid) corr_trace[((size_t)it * maxB + b) * nslots + slot] = rb.v[7] != 0.0 ? (int)(unsigned int)key : -1;
                // the next iteration's bound: this match (an inactive point of a coarse iteration keeps the one it has; none yet in iteration 0)
                if (active || it == 0) {
                    if (resident) s_prev[res_a ? 0 : 1][lane] = pq;
                    else if (valid) gprev[idx] = pq;
                }
                if (stamp) { const long long now_ = (long long)wall_clock64(); ph[3] += now_ - ph_prev; ph_prev = now_; }
                tile_accumulate(rb, set, slab);
                if (stamp) { const long long now_ = (long long)wall_clock64(); ph[4] += now_ - ph_prev; ph_prev = now_; }
            }
        }
        // ---- grid barrier of the pair: every block's Gram sums are in the accumulator set
        ls_barrier();
        if (w == 0) {
            // Wave 0 issued every global write of this block that another block reads: the Gram atomics (device-scope read-modify-writes,
            // performed where all XCDs see them).  They must have been PERFORMED before the ticket is taken -- vmcnt(0) --, nothing has
            // to be written back or invalidated: __threadfence() here is buffer_wbl2 + buffer_inv on gfx950, an L2 flush per block and
            // iteration that turned every later read of the (read-only) target into a miss.  The totals are read with device-scope loads.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                __hip_atomic_fetch_add(ticket + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int want = (unsigned int)(it + 1) * (unsigned int)G;
                if (stamp) { const long long now_ = (long long)wall_clock64(); ph[5] += now_ - ph_prev; ph_prev = now_; }
                while (__hip_atomic_load(ticket + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
            }
        }
        ls_barrier();
        if (stamp) { const long long now_ = (long long)wall_clock64(); ph[6] += now_ - ph_prev; ph_prev = now_; }
        const long long *__restrict__ A = acc + ((size_t)b * nsets + it) * ACC_R * ACC_STRIDE;
        if (tid < NRAW) {
            long long q = 0;
#pragma unroll
            for (int r = 0; r < ACC_R; ++r) q += __hip_atomic_load(A + r * ACC_STRIDE + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            Gs[tid] = q;
        }
        ls_barrier();
        if (stamp) { const long long now_ = (long long)wall_clock64(); ph[7] += now_ - ph_prev; ph_prev = now_; }
        if (tid < NSUMS) tot[tid] = derive_sum(EST, g.eb, tid, Gs);
        ls_barrier();
        if (stamp) { const long long now_ = (long long)wall_clock64(); ph[8] += now_ - ph_prev; ph_prev = now_; }
        if (blockIdx.x == 0 && tid < NSUMS) trace_S[((size_t)b * iters + it) * NSUMS + tid] = tot[tid];
        if (w == 0) {
            if constexpr (EST == 0) {
                const double Tn = list_solve_p2p(tot, Tsh, &s_rc);
                __builtin_amdgcn_wave_barrier();
                if (lane < 16) Tsh[lane] = Tn;
            } else {
                if (lane == 0) s_rc = list_solve_svd(tot, Tsh, slab);       // (the Gram slab is free between two tiles)
            }
        }
        ls_barrier();
        const int rc = s_rc;
        if (stamp) { const long long now_ = (long long)wall_clock64(); ph[9] += now_ - ph_prev; ph_prev = now_; }
        if (stamp) {
            ph[10] = n_scanned;
            long long *__restrict__ o = dbg + ((size_t)it * G + blockIdx.x) * 12;
            for (int k = 0; k < 12; ++k) o[k] = ph[k];
        }
        if (rc == 2) flag |= 1;
        if (rc == 0) flag |= 2;            // no update in this iteration: never a silent "ok"
        if (blockIdx.x == 0) {
            if (tid < 16) trace_T[((size_t)b * (iters + 1) + it + 1) * 16 + tid] = Tsh[tid];
            if (last) {
                if (tid < 16) Tcur[b * 16 + tid] = Tsh[tid];
                if (tid == 0) flags[b] = flag;
                if (res_host) {
                    double *__restrict__ r = res_host + (size_t)b * RES_REC;
                    if (tid < 16) r[tid] = Tsh[tid];
                    if (tid < NSUMS) r[16 + tid] = tot[tid];
                    if (tid == 0) { r[45] = (double)flag; r[46] = (double)ns; r[47] = (double)nt; }
                }
            }
        }
        ls_barrier();                   // (s_rc / Tsh / tot are rewritten by the next iteration)
    }
    if (end_run && blockIdx.x == 0 && b == 0 && tid == 0) atomicAdd(end_run, -1);      // one run less in flight on the device (k_pair_init counted it in)
}

// slot-order correspondences of a list handle -> original order (get_correspondences; k_scatter_corr needs the tile-major source slots,
// which the list path does not build): slot = (i / 8) * 64 + i % 8.  Invalid source points keep k_fill_corr's -1 / +inf.
__global__ __launch_bounds__(256) void k_scatter_corr_list(const PairPtrs *__restrict__ pairs, const int *__restrict__ corr, const float *__restrict__ cd2,
                                                           int b, int N, float zmax, int *__restrict__ idx, float *__restrict__ d2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float4 p = pairs[b].src[i];
    if (!pt_valid(p.x, p.y, p.z, zmax)) return;
    const int slot = ((i >> 3) << 6) + (i & 7);
    idx[i] = corr[slot];
    if (cd2) d2[i] = cd2[slot];
}

} // namespace s3d
