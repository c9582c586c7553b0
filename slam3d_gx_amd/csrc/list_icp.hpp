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

// (1a) grid (ceil(N / 256), ntasks), whole blocks: every kept point takes a rank in its cell.  cnt / grp are zero on entry (self-cleaning: k_list_scan
// zeroes the cells it read, k_list_boxes the groups).
__global__ __launch_bounds__(256) void k_list_bin(ListTasks a, int *__restrict__ cnt, int *__restrict__ grp, int2 *__restrict__ cr, int N, float zmax)
{
    const int t = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const ListTask &T = a.t[t];
    float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const bool ok = i < N && i >= T.i_begin && compact_keep(T.cloud, T.nrm, i, T.i_end, T.which, T.use_normals, zmax, q);
    const int c = ok ? ls_cell(q.x, q.y, q.z, zmax) : -1;
    // One returning atomic per DISTINCT cell of the wave, not per point: neighbours in a list are neighbours in space (a voxel grid's
    // output is sorted by voxel), and 16 k same-address returning atomics from all XCDs took 45 us -- two thirds of the preprocessing.
    int rank = 0;
    unsigned long long todo = __ballot(ok);
    while (todo != 0ull) {
        const int leader = __builtin_ctzll(todo);
        const int c0 = __builtin_amdgcn_readlane(c, leader);
        const unsigned long long same = __ballot(ok && c == c0);
        int base = 0;
        if (lane == leader) {
            base = atomicAdd(cnt + (size_t)t * LS_NCELL + c0, (int)__popcll(same));
            atomicAdd(grp + t * LS_NGROUP + (c0 >> 7), (int)__popcll(same));
        }
        base = __builtin_amdgcn_readlane(base, leader);
        if (ok && c == c0) rank = base + (int)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if (i < N) cr[(size_t)t * N + i] = make_int2(c, rank);
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
// The svd estimator's step (solve_step_one(sums, T, 1): the same operations in the same order per value, so the same bits) for the
// persistent kernel.  Inlined, the register form (H, g, v, dR, T: ~60 doubles in one lane) pushed the loop-carried values of the search
// out to scratch (33 spilled VGPRs, reloaded at the head of every iteration); one lane working out of LDS took 7.9 us per iteration.
// Spread over a wave: lane m < 3 owns ROW m of g and of v (lanes >= 3 shadow row 2); a column dot product is three
// products, one per row, summed in the serial code's order from v_readlane values -- (g0p g0q + g1p g1q) + g2p g2q --, so every lane
// computes the same al, be, ga, c and s and rotates its own row.  Same operations, same order per value as solve_step_one: the same bits (tests/test_unorganized.py compares every iterate with the oracle's), a third of the arithmetic and no
// LDS round trip inside the chain (the one-lane LDS form: 7.9 us per iteration).  All 64 lanes call it; T is read before it is written.
__device__ __noinline__ int list_solve_svd_wave(const double *sums, double *T)
{
    const int lane = threadIdx.x & 63, m = lane < 2 ? lane : 2;
    const double n = sums[27];
    if (n < 3.0) return 0;
    const double pm0 = sums[0] / n, pm1 = sums[1] / n, pm2 = sums[2] / n, qm0 = sums[3] / n, qm1 = sums[4] / n, qm2 = sums[5] / n;
    const double pmm = m == 0 ? pm0 : (m == 1 ? pm1 : pm2);
    double g0 = sums[6 + m * 3 + 0] - (n * pmm) * qm0, g1 = sums[6 + m * 3 + 1] - (n * pmm) * qm1, g2 = sums[6 + m * 3 + 2] - (n * pmm) * qm2;
    double v0 = m == 0 ? 1.0 : 0.0, v1 = m == 1 ? 1.0 : 0.0, v2 = m == 2 ? 1.0 : 0.0;
    auto col3 = [&](double x) __attribute__((always_inline)) { return (bcast_d(x, 0) + bcast_d(x, 1)) + bcast_d(x, 2); };
    for (int sweep = 0; sweep < 12; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double &gp = k == 2 ? g1 : g0, &gq = k == 0 ? g1 : g2, &vp = k == 2 ? v1 : v0, &vq = k == 0 ? v1 : v2;
            const double al = col3(gp * gp), be = col3(gq * gq), ga = col3(gp * gq);
            if (ga * ga <= 0x1p-100 * (al * be)) continue;
            rotated = true;
            const double zeta = (be - al) / (2.0 * ga);
            double t = 1.0 / (fabs(zeta) + sqrt(zeta * zeta + 1.0));
            if (zeta < 0.0) t = -t;
            const double c = 1.0 / sqrt(t * t + 1.0);
            const double s = c * t;
            const double a = gp, b_ = gq;
            gp = c * a - s * b_;
            gq = s * a + c * b_;
            const double e = vp, f = vq;
            vp = c * e - s * f;
            vq = s * e + c * f;
        }
        if (!rotated) break;
    }
    const double sg0 = sqrt(col3(g0 * g0)), sg1 = sqrt(col3(g1 * g1)), sg2 = sqrt(col3(g2 * g2));
    auto sgof = [&](int i) __attribute__((always_inline)) { return i == 0 ? sg0 : (i == 1 ? sg1 : sg2); };
    int i0 = 0, i1 = 1, i2 = 2, tmp;
    if (sgof(i1) > sgof(i0)) { tmp = i0; i0 = i1; i1 = tmp; }
    if (sgof(i2) > sgof(i1)) { tmp = i1; i1 = i2; i2 = tmp; }
    if (sgof(i1) > sgof(i0)) { tmp = i0; i0 = i1; i1 = tmp; }
    double R0 = m == 0 ? 1.0 : 0.0, R1 = m == 1 ? 1.0 : 0.0, R2 = m == 2 ? 1.0 : 0.0;       // row m of dR
    if ((sgof(i0) > 0.0) && (sgof(i1) > 1e-14 * sgof(i0))) {
        const double s0 = sgof(i0), s1 = sgof(i1);
        const double u0m = (i0 == 0 ? g0 : (i0 == 1 ? g1 : g2)) / s0, u1m = (i1 == 0 ? g0 : (i1 == 1 ? g1 : g2)) / s1;      // u0[m], u1[m]
        const double w0m = i0 == 0 ? v0 : (i0 == 1 ? v1 : v2), w1m = i1 == 0 ? v0 : (i1 == 1 ? v1 : v2);                // v0[m], v1[m]
        const double u00 = bcast_d(u0m, 0), u01 = bcast_d(u0m, 1), u02 = bcast_d(u0m, 2), u10 = bcast_d(u1m, 0), u11 = bcast_d(u1m, 1), u12 = bcast_d(u1m, 2);
        const double w00 = bcast_d(w0m, 0), w01 = bcast_d(w0m, 1), w02 = bcast_d(w0m, 2), w10 = bcast_d(w1m, 0), w11 = bcast_d(w1m, 1), w12 = bcast_d(w1m, 2);
        const double u20 = u01 * u12 - u02 * u11, u21 = u02 * u10 - u00 * u12, u22 = u00 * u11 - u01 * u10;
        const double w20 = w01 * w12 - w02 * w11, w21 = w02 * w10 - w00 * w12, w22 = w00 * w11 - w01 * w10;
        const double w2m = m == 0 ? w20 : (m == 1 ? w21 : w22);
        R0 = (w0m * u00 + w1m * u10) + w2m * u20;
        R1 = (w0m * u01 + w1m * u11) + w2m * u21;
        R2 = (w0m * u02 + w1m * u12) + w2m * u22;
    }
    const double qmm = m == 0 ? qm0 : (m == 1 ? qm1 : qm2);
    const double dt = qmm - ((R0 * pm0 + R1 * pm1) + R2 * pm2);
    const double t0 = (R0 * T[0] + R1 * T[4]) + R2 * T[8], t1 = (R0 * T[1] + R1 * T[5]) + R2 * T[9], t2 = (R0 * T[2] + R1 * T[6]) + R2 * T[10];
    const double t3 = ((R0 * T[3] + R1 * T[7]) + R2 * T[11]) + dt;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                               // (every lane has read T)
    if (lane < 3) { T[lane * 4 + 0] = t0; T[lane * 4 + 1] = t1; T[lane * 4 + 2] = t2; T[lane * 4 + 3] = t3; }
    if (lane == 3) { T[12] = 0.0; T[13] = 0.0; T[14] = 0.0; T[15] = 1.0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
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

constexpr unsigned int LS_BARRIER_POLLS = 1500000u;         // ticket polls (a memory round trip + s_sleep each, ~1.5 us) a block spends at a grid barrier before it gives the run up
constexpr unsigned int LS_ABORT = 0x40000000u;              // bit of the barrier ticket (arrivals stay below 2^13): the run was given up, every barrier is open
// the wait of the grid barrier, out of line (the gated instance of the kernel has no register to spare: inlined, this loop cost it two
// spills): wave-uniform -- lane 0 loads, the value is broadcast -- so the poll counter lives in an SGPR
__device__ __noinline__ void ls_wait_ticket(unsigned int *ticket, unsigned int want)
{
    const int lane = threadIdx.x & 63;
    for (unsigned int polls = 0u;; ++polls) {
        unsigned int seen = 0u;
        if (lane == 0) seen = __hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned int)__builtin_amdgcn_readfirstlane((int)seen) >= want) break;
        if (polls > LS_BARRIER_POLLS) {
            if (lane == 0) __hip_atomic_fetch_or(ticket, LS_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
constexpr int LS_LDS_TILES = 384;                       // target tiles whose boxes (32 B) and table entries (8 B) fit the block's LDS budget
constexpr int LS_CAND = 32;                             // candidate tiles a wave lists before it scans them

template <int EST, bool GATED, bool DBG = false>
__global__ __launch_bounds__(64 * LS_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_list_icp(
    const PairPtrs *__restrict__ pairs, Geometry g, int iters, int n_coarse_and_hook /* coarse iterations | (1 + iteration whose barrier block 0 of pair 0 never reaches: the watchdog's test hook, 0 = none) << 16 -- the gated instance has no register left for another argument */, int nsets, int N,
    double *__restrict__ Tcur, double *__restrict__ trace_T, double *__restrict__ trace_S, int *__restrict__ flags,
    long long *__restrict__ acc, unsigned int *__restrict__ ticket, unsigned int *__restrict__ claim /* [B], zero at launch */, float4 *__restrict__ prev /* [B][N]: previous match (x, y, z, index) by sorted source position */,
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
    __shared__ float4 s_prev[1][64];
    __shared__ int s_claim;
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
    // the block's OWN source tile (tile = block): resident in registers.  The tiles beyond the grid width (a 16 k-point list has ~330
    // tiles for 256 blocks) are CLAIMED, one at a time, by whichever block has finished: blocks whose own tile is cheap take them, and the
    // slowest block -- which every other block waits for at the barrier -- is the one with the dearest single tile, not the one that
    // was dealt two dear ones (static deal: slowest block 30-37 us, median 6-9).
    float4 r_s4a = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
    int r_starta = 0, r_cnta = 0;
    if ((int)blockIdx.x < nst) {
        const ls_i2 sd = stile[blockIdx.x];
        r_starta = sd.x; r_cnta = sd.y;
        if (lane < r_cnta) r_s4a = ls_ld(spts, r_starta + lane);
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
        const bool coarse = it < (n_coarse_and_hook & 0xffff);
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
        const int extras = nst > G ? nst - G : 0;                     // tiles beyond the grid: claimed
        // (the claim for the NEXT tile is taken when the scans of the current one are merged: its round trip flies under the rows and the
        //  Gram sums -- asked for after them it sat between the slowest block's last row and the barrier, 2.7 us; asked for a whole tile
        //  ahead the extra tiles went to whoever STARTED first, not to whoever finished: the static deal again, 0.84 instead of 0.72 ms)
        unsigned int my_claim = 0u;
        for (int rnd = 0; ; ++rnd) {
            int tile = blockIdx.x;
            if (rnd > 0) {
                // every block claims until a claim fails: extras + G claims per iteration, so iteration `it` starts at it * (extras + G)
                if (extras == 0) break;
                if (tid == 0) s_claim = (int)(my_claim - (unsigned int)it * (unsigned int)(extras + G));
                ls_barrier();
                const int c = s_claim;
                ls_barrier();
                if (c >= extras) break;
                tile = G + c;
            } else if (tile >= nst) continue;
            const bool resident = rnd == 0;
            int start = r_starta, cnt = r_cnta;
            float4 s4 = r_s4a;
            float4 pv = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
            if (!resident) {
                const ls_i2 sd = stile[tile];                 // (wave-uniform)
                start = sd.x; cnt = sd.y;
                s4 = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
                if (lane < cnt) {
                    s4 = ls_ld(spts, start + lane);
                    if (it > 0) {      // its previous matches were stored by another block, maybe on another XCD: device-scope loads
                        const unsigned long long *pg = reinterpret_cast<const unsigned long long *>(gprev + start + lane);
                        const unsigned long long lo = __hip_atomic_load(pg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi = __hip_atomic_load(pg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pv = make_float4(__int_as_float((int)(unsigned int)lo), __int_as_float((int)(unsigned int)(lo >> 32)), __int_as_float((int)(unsigned int)hi), __int_as_float((int)(unsigned int)(hi >> 32)));
                    }
                }
            } else pv = s_prev[0][lane];
            const int idx = start + lane;
            const bool valid = lane < cnt;
            const int i = __float_as_int(s4.w);
            const bool active = valid && !(coarse && ((i >> 3) & 3) != 0);       // spec S4c on a list: the points of every fourth group of eight
            float px, py, pz;
            xform(m, s4.x, s4.y, s4.z, px, py, pz);
            unsigned long long key = key_gate;
            if (active && it > 0) {
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
                    // (beyond LS_LDS_TILES the table entry comes through the SCALAR cache: a vector load here -- what hipcc made of it even on the LDS
                    //  branch's join -- was followed by s_waitcnt vmcnt(0), which also drained the two tile loads in flight.  Measured after the fix:
                    //  the same 29 us per iteration -- a tile's cost is its 64 candidates at one wave per SIMD (tools/ubench_pk.hip: 40 cycles per
                    //  candidate alone on a SIMD, 21 with four waves), not its load)
                    const ls_i2 tdc = lds_boxes ? ls_i2{ 0, 0 } : ((ls_cptri2)pp.ls_ttile)[t_];
                    const int2 td = lds_boxes ? s_tile[t_] : make_int2(tdc.x, tdc.y);
                    n_ = kk < nc ? __builtin_amdgcn_readfirstlane(td.y) : 0;
                    ls_issue(r, tpts + (td.x + lane));            // all 64 lanes: behind a short tile lie the next tile's points or the padding
                };
                auto scan_one = [&](const ls_f4 &r, int n_) __attribute__((always_inline)) {
                    s_stage[w][lane] = lane < n_ ? make_float4(r.x, r.y, r.z, r.w) : make_float4(inf, inf, inf, __int_as_float(-1));
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    // Four candidates interleaved by hand, two key accumulators: with ONE wave per SIMD nothing else fills the issue slots
                    // behind a dependent instruction (measured: ~10 cycles per VALU instruction in the straight form, 2.5 us per 64-candidate
                    // tile; variants timed on the 16 k-point pair, tools/variants: LDS read + float minimum only -- the same time; lane
                    // broadcast by v_readlane instead of LDS -- the same; next reads prefetched under the arithmetic -- the same; this form -9 %).
                    unsigned long long kb = key;
                    for (int c0 = 0; c0 < n_; c0 += 8) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float4 q0 = s_stage[w][c0 + 4 * h], q1 = s_stage[w][c0 + 4 * h + 1], q2 = s_stage[w][c0 + 4 * h + 2], q3 = s_stage[w][c0 + 4 * h + 3];
                            const float x0 = q0.x - px, x1 = q1.x - px, x2 = q2.x - px, x3 = q3.x - px;
                            const float y0 = q0.y - py, y1 = q1.y - py, y2 = q2.y - py, y3 = q3.y - py;
                            const float z0 = q0.z - pz, z1 = q1.z - pz, z2 = q2.z - pz, z3 = q3.z - pz;
                            float a0 = x0 * x0, a1 = x1 * x1, a2 = x2 * x2, a3 = x3 * x3;                // the canonical d2: fma(dz, dz, fma(dy, dy, dx dx))
                            a0 = __fmaf_rn(y0, y0, a0); a1 = __fmaf_rn(y1, y1, a1); a2 = __fmaf_rn(y2, y2, a2); a3 = __fmaf_rn(y3, y3, a3);
                            a0 = __fmaf_rn(z0, z0, a0); a1 = __fmaf_rn(z1, z1, a1); a2 = __fmaf_rn(z2, z2, a2); a3 = __fmaf_rn(z3, z3, a3);
                            key = key_min(key, ((unsigned long long)(unsigned int)__float_as_int(a0) << 32) | (unsigned int)__float_as_int(q0.w));
                            kb = key_min(kb, ((unsigned long long)(unsigned int)__float_as_int(a1) << 32) | (unsigned int)__float_as_int(q1.w));
                            key = key_min(key, ((unsigned long long)(unsigned int)__float_as_int(a2) << 32) | (unsigned int)__float_as_int(q2.w));
                            kb = key_min(kb, ((unsigned long long)(unsigned int)__float_as_int(a3) << 32) | (unsigned int)__float_as_int(q3.w));
                        }
                    }
                    key = key_min(key, kb);
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
            if (extras > 0 && tid == 0) my_claim = __hip_atomic_fetch_add(claim + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                    if (resident) s_prev[0][lane] = pq;
                    else if (valid) {      // (device-scope stores: the next iteration's claimant may sit on another XCD)
                        unsigned long long *pg = reinterpret_cast<unsigned long long *>(gprev + idx);
                        __hip_atomic_store(pg, (unsigned long long)(unsigned int)__float_as_int(pq.x) | ((unsigned long long)(unsigned int)__float_as_int(pq.y) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(pg + 1, (unsigned long long)(unsigned int)__float_as_int(pq.z) | ((unsigned long long)(unsigned int)__float_as_int(pq.w) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (stamp) { const long long now_ = (long long)wall_clock64(); ph[3] += now_ - ph_prev; ph_prev = now_; }
                tile_accumulate(rb, set, slab);
                if (stamp) { const long long now_ = (long long)wall_clock64(); ph[4] += now_ - ph_prev; ph_prev = now_; }
            }
        }
        // ---- grid barrier of the pair: every block's Gram sums are in the accumulator set
        if (tid < NRAW) Gs[tid] = 0;
        ls_barrier();
        if (w == 0) {
            // Wave 0 issued every global write of this block that another block reads: the Gram atomics (device-scope read-modify-writes,
            // performed where all XCDs see them).  They must have been PERFORMED before the ticket is taken -- vmcnt(0) --, nothing has
            // to be written back or invalidated: __threadfence() here is buffer_wbl2 + buffer_inv on gfx950, an L2 flush per block and
            // iteration that turned every later read of the (read-only) target into a miss.  The totals are read with device-scope loads.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0 && !(it + 1 == (n_coarse_and_hook >> 16) && b == 0 && blockIdx.x == 0)) __hip_atomic_fetch_add(ticket + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int want = (unsigned int)(it + 1) * (unsigned int)G;
            if (stamp) { const long long now_ = (long long)wall_clock64(); ph[5] += now_ - ph_prev; ph_prev = now_; }
            // WATCHDOG.  The barrier needs every block of the launch resident; the launch is sized so that this holds for four launches
            // of one process, but nothing on the device can promise it against OTHER processes' persistent work.
            // A block that has polled LS_BARRIER_POLLS times (about two seconds) raises LS_ABORT in the ticket word itself: every waiter (and every later
            // barrier) then passes at once, the launch runs to its end on garbage, and block 0 reports flag 4 -- the host returns
            // SLAM3D_E_HIP with an identity pose instead of hanging the device.  The loop is wave-uniform (lane 0 loads, the value is
            // broadcast), so its counter lives in an SGPR: the kernel has no VGPR to spare.
            ls_wait_ticket(ticket + b, want);
        }
        ls_barrier();
        if (stamp) { const long long now_ = (long long)wall_clock64(); ph[6] += now_ - ph_prev; ph_prev = now_; }
        const long long *__restrict__ A = acc + ((size_t)b * nsets + it) * ACC_R * ACC_STRIDE;
        {   // the 16 replicas x 36 totals: every thread fetches two or three words (device-scope loads, all in flight together) and adds
            // them into the LDS totals (zeroed before the grid barrier) -- 36 threads x 16 loads each took 4.4 us
            long long v[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int idx = tid + u * 64 * LS_WAVES;
                v[u] = 0;
                if (idx < ACC_R * ACC_STRIDE && idx % ACC_STRIDE < NRAW) v[u] = __hip_atomic_load(A + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int idx = tid + u * 64 * LS_WAVES;
                if (idx < ACC_R * ACC_STRIDE && idx % ACC_STRIDE < NRAW && v[u] != 0)
                    atomicAdd(reinterpret_cast<unsigned long long *>(&Gs[idx % ACC_STRIDE]), (unsigned long long)v[u]);
            }
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
                const int rcw = list_solve_svd_wave(tot, Tsh);
                if (lane == 0) s_rc = rcw;
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
        if (last && blockIdx.x == 0 && tid == 0 && (__hip_atomic_load(ticket + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & LS_ABORT)) flag |= 4;      // the run was given up at a barrier
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
