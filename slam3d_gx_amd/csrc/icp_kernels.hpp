// icp_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) of the plane-ICP path.
//
// Stage map (DESIGN.md section 3 / SURVEY.md section 8a):
//   S1 k_backproject      u16 depth -> organized float4 cloud      src/convert2PCD.cpp:54-72
//   S2 k_normals          7x7 window covariance -> normal          src/planarFeatures.cpp:88-136 (a7)
//   S3 k_frame_tiles      8x8-pixel tiles (one wavefront each) of a frame as source / as target + target AABBs
//      k_compact_*        raster-ordered dense lists (only the full brute-force modes need them)
//   S4 k_nn_tiles_acc     exact tile-pruned 1-NN fused with the normal-equation accumulation
//      k_nn_valu          exact full brute-force 1-NN (replaces FLANN matching, src/GraphicEnd.cpp:486-520)
//      k_accumulate       point-to-plane / Kabsch normal equations, deterministic 256-slot chunk tree
//   S5 k_solve_acc        integer accumulators -> 29 sums, 6x6 LDL^T or 3x3 SVD, SE(3) update on device
//   a6 k_fit_moments / k_fit_refine (plane_seg.hpp)  per-plane {sum p, sum pp^T, n} -> (n,d)   src/GraphicEnd.cpp:360-387
//
// Numerics contract: compiled with -ffp-contract=off; every float/double operation is an
// individually rounded IEEE op in the order written (explicit __fmaf_rn where the spec has an
// fma), so results are bit-identical to the CPU oracle's restatement of the same spec.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace s3d {

constexpr int NSUMS = 29;         // the derived sums of the trace / the solve (21 A^T A + 6 A^T b + count + sum b^2)
constexpr int NRAW = 36;          // what is ACCUMULATED: the upper triangle of the 8x8 integer Gram matrix of the quantised row vectors (spec S4)
constexpr int CHUNK = 256;
constexpr int NN_TILE = 1024;      // targets staged in LDS per tile (brute-force kernel)
constexpr int NN_QPT = 4;          // queries per thread (brute-force VALU kernel)
constexpr int NN_BLOCK = 256;

// A FRAME is the resident unit: its organized cloud and what the two roles need of it -- as an ICP *target* the
// normals, tile records and boxes, as a *source* the tile-major slots.  They are built once per frame (and role), not
// once per pair: the 32 loop-closure candidates of src/GraphicEnd.cpp:685-762 share one target frame, and the
// keyframe of GraphicEnd::run (src/GraphicEnd.cpp:168) stays the source of many consecutive pairs.
// A PAIR is two frame references plus its own iteration state (T, accumulators, slot records / prevq, ownership map).
// SLAM3D_EST_PLANE (spec S2p): the planes of a frame as its last build found them (plane_seg.hpp fills it)
struct FramePlane { float a, b, c, d, cx, cy, cz; int count; };
struct FramePlanes { FramePlane pl[8]; int n, pad[3]; };
struct PairPtrs {                  // per frame-pair device pointers: the resident products of its two frames
    const float4 *src;             // organized source cloud (only the brute-force compaction reads it)
    const float4 *tgt;             // organized target cloud
    const float4 *nrm;             // target normals
    const float4 *snrm;            // source frame's normals (only the optional normal-angle gate reads them)
    const float4 *tq;              // target frame in image order, records (pixel, x, y, z), invalid = (-1, +inf ...): the
                                   // projective window search; null unless the target cloud is a back-projected depth image
    const float4 *srcT;            // source tile slots   [ntiles * 64]
    const float4 *tgtT;            // target tile records [ntiles * TILE_REC]
    const float4 *tbox;            // target tile boxes   [ntiles * 2]
    const float4 *cbox;            // target coarse boxes [ncoarse * 2]
    const int *src_counts;         // [0] = valid source points of the source frame (inside its row shard)
    const int *tgt_counts;         // [1] = valid target points of the target frame
    const FramePlanes *spl, *tpl;  // planes of the source / target frame (SLAM3D_EST_PLANE; null otherwise)
    int *assoc;                    // [8] plane-pair gate: target plane of every source plane (-1 none), written by k_plane_assoc at the start of a run
    // point lists (height == 1 handles, list_icp.hpp): both frames' points sorted into space cells, 64-point tiles and their boxes
    const float4 *ls_src, *ls_tgt; // sorted (x, y, z, original index) of the source / target frame's role list
    const float4 *ls_tbox;         // target tile boxes [2 t], [2 t + 1]
    const int2 *ls_stile, *ls_ttile;   // tile tables (start, count <= 64) of the two lists
    const int *ls_ns, *ls_nt;      // [0] points, [1] tiles of the two lists
};
constexpr int RES_REC = 48;        // doubles per pair in the host-mapped result record
constexpr unsigned long long HEAD_EMPTY = 0x7ff8dead0badc0deull;     // a pose entry "not published yet" (head solve): a quiet NaN with a payload no arithmetic produces
// Launch stamps (opt-in, slam3d_icp_set_stamping): every block of a stamped launch folds the constant-rate 100 MHz
// real-time counter (common to all XCDs, unlike s_memtime) into the launch's row -- STAMP_R replicas of the earliest
// start (atomic min) followed by STAMP_R replicas of the latest end (atomic max), fire-and-forget atomics spread
// over the replicas, ONE of each per block (the block's last wave stamps the end).  They tell WHEN a launch really
// occupied the chip without a tracer serialising the streams (profiles/r03_overlap.md).  The rows of the last
// `ring` runs stay on the device (StampRing: k_pair_init advances `seq` and resets the new run's rows; a launch finds
// its row through seq), so nothing is copied while the measurement runs.  A null pointer (the default) costs one
// scalar branch.
// (head solve, see k_nn_tiles_acc) Poll or solve?  Measured on the MI355X (640x480, 20 iterations): with ONE alignment on the chip every block solving for
// itself is fastest (0.761 ms per alignment; polling 0.78; two launches per iteration 0.80) -- the solve is ~1.6 us, a
// published pose needs that plus a trip through memory; with FOUR alignments in flight the 1,200 redundant solves of a
// launch are VALU work the other alignments want (63 k it/s against 68 k polling).  So the blocks poll only while other
// runs are in flight on the device: k_pair_init / the run's last k_solve_acc keep the count (slam3d_icp_run's launches
// only).  A stale count (a run that died between the two) costs speed, never correctness.
// Round 5: the count is NOT a module global any more but a word of per-DEVICE state that the library owns
// (icp_capi.hip: device_state(): allocated with the first handle of a device, freed with the last, handed to every handle of that
// device): `runs` below.  Contract: handles are used by one thread each; handles of one device may run on different threads --
// the word is only ever touched by device-side atomics, it shapes scheduling (poll or solve), never a result.

constexpr int STAMP_R = 16, STAMP_ROW = 2 * STAMP_R;
struct StampRing { unsigned long long *rows; unsigned int *seq; int ring, rows_per_run; };
__device__ __forceinline__ unsigned long long *stamp_row(const StampRing &sr, int row)
{
    if (!sr.rows) return nullptr;
    const unsigned int run = __builtin_amdgcn_readfirstlane(*sr.seq) % (unsigned int)sr.ring;
    return sr.rows + ((size_t)run * sr.rows_per_run + row) * STAMP_ROW;
}
__device__ __forceinline__ void stamp_start(unsigned long long *__restrict__ row, int c)
{
    if (row) atomicMin(row + (c & (STAMP_R - 1)), (unsigned long long)wall_clock64());
}
__device__ __forceinline__ void stamp_end(unsigned long long *__restrict__ row, int c)
{
    if (row) atomicMax(row + STAMP_R + (c & (STAMP_R - 1)), (unsigned long long)wall_clock64());
}

constexpr int PAIR_ARGS = 16;       // (x 152 bytes: kernel arguments are limited to 4 KB)
struct PairArgs { PairPtrs p[PAIR_ARGS]; };
// the pair table travels as a kernel argument (copied at launch), not through pinned host memory
__global__ void k_set_pairs(PairPtrs *__restrict__ dst, PairArgs a, int n)
{
    if ((int)threadIdx.x < n) dst[threadIdx.x] = a.p[threadIdx.x];
}

// one (frame, role) to (re)build; role 0 = source (tile-major slots of rows [row0, row1)), 1 = target
struct FrameTask {
    const float4 *cloud;
    float4 *nrm;                   // target role: the frame's normals (read by the tile build when use_normals)
    float4 *tiles;                 // srcT or tgtT of the frame
    float4 *tq;                    // target role: the frame in image order for the projective window search
    float4 *tbox, *cbox;           // target role only
    int *scount;                   // per-tile valid counts (scratch of the frame), totalled by k_coarse_boxes
    int *counts;                   // the frame's totals: [0] source role, [1] target role
    int role, row0, row1, use_normals;
};
constexpr int FRAME_ARGS = 32;
struct FrameTasks { FrameTask t[FRAME_ARGS]; };

struct Geometry {
    int W, H, N;
    float zmax;
    int win_r, min_in;
    double in_dist;
    float gate2;
    float resid2, min_ncos;        // optional gates of the point-to-plane estimator (spec S4g), 0 = off
    int pair_gate;                 // spec S4p: the plane-pair gate of SLAM3D_EST_PLANE (normal.w carries 1 + plane)
    float proj_c;                  // projective window search: pixels of radius r around a query's projection cover every target
                                   // closer than (r + 0.49) * z / proj_c  (= fmax * sqrt(1 + amax^2 + bmax^2) * 1.001), DESIGN.md 5
    int estimator;
    float cert_m, cert_track;      // clearance certificates: margin added to every pruning radius / pose motion below which a launch tracks (CERT_M, CERT_TRACK_MOTION)
    int eb;                        // spec S4: the residual component of a row vector is rint(b * 2^eb), eb = 20 - min(k, 8) with gate = m 2^k, 0.5 <= m < 1
    double b_scale;                // 2^eb
    double fx, fy, cx, cy, factor, zf;
};

__device__ __forceinline__ bool pt_valid(float x, float y, float z, float zmax)
{
    return isfinite(x) && isfinite(y) && isfinite(z) && z > 0.0f && z <= zmax;
}

// ------------------------------------------------------------------------------------ S1
__global__ __launch_bounds__(256) void k_backproject(const uint16_t *__restrict__ depth, float4 *__restrict__ out,
                                                      Geometry g)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= g.N) return;
    const int v = i / g.W, u = i - v * g.W;
    const uint16_t d = depth[i];
    const double z = (double)d / g.factor;
    float4 o;
    if (d == 0 || !(z <= g.zf)) {
        const float qnan = __int_as_float(0x7fc00000);
        o = make_float4(qnan, qnan, qnan, 0.0f);
    } else {
        const double x = ((double)u - g.cx) * z / g.fx;
        const double y = ((double)v - g.cy) * z / g.fy;
        o = make_float4((float)x, (float)y, (float)z, 1.0f);
    }
    out[i] = o;
}

// records of arbitrary stride (pcl::PointXYZRGBA = 32 B) -> packed float4
__global__ __launch_bounds__(256) void k_repack(const unsigned char *__restrict__ raw, int stride,
                                                 float4 *__restrict__ out, int N)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float *p = reinterpret_cast<const float *>(raw + (size_t)i * stride);
    out[i] = make_float4(p[0], p[1], p[2], 1.0f);
}

// --------------------------------------------------------------------- 3x3 Jacobi eigen
struct Sym3 { double a00, a01, a02, a11, a12, a22; };
struct Mat3 { double m00, m01, m02, m10, m11, m12, m20, m21, m22; };

__device__ __forceinline__ void jrot(double &app, double &aqq, double &apq, double &arp, double &arq,
                                     double &v0p, double &v0q, double &v1p, double &v1q, double &v2p, double &v2q)
{
    if (apq == 0.0) return;
    const double theta = (aqq - app) / (2.0 * apq);
    double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
    if (theta < 0.0) t = -t;
    const double c = 1.0 / sqrt(t * t + 1.0);
    const double s = t * c;
    app = app - t * apq;
    aqq = aqq + t * apq;
    apq = 0.0;
    const double rp = arp, rq = arq;
    arp = c * rp - s * rq;
    arq = s * rp + c * rq;
    double a = v0p, b = v0q; v0p = c * a - s * b; v0q = s * a + c * b;
    a = v1p; b = v1q;        v1p = c * a - s * b; v1q = s * a + c * b;
    a = v2p; b = v2q;        v2p = c * a - s * b; v2q = s * a + c * b;
}

// eigenvector (unit) of the smallest eigenvalue; cyclic Jacobi, 8 sweeps, order (0,1),(0,2),(1,2)
__device__ __forceinline__ void eig3_smallest(Sym3 A, double &nx, double &ny, double &nz)
{
    Mat3 V = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
#pragma unroll 1
    for (int sweep = 0; sweep < 8; ++sweep) {
        jrot(A.a00, A.a11, A.a01, A.a02, A.a12, V.m00, V.m01, V.m10, V.m11, V.m20, V.m21); // (0,1) r=2
        jrot(A.a00, A.a22, A.a02, A.a01, A.a12, V.m00, V.m02, V.m10, V.m12, V.m20, V.m22); // (0,2) r=1
        jrot(A.a11, A.a22, A.a12, A.a01, A.a02, V.m01, V.m02, V.m11, V.m12, V.m21, V.m22); // (1,2) r=0
    }
    double e = A.a00; nx = V.m00; ny = V.m10; nz = V.m20;
    if (A.a11 < e) { e = A.a11; nx = V.m01; ny = V.m11; nz = V.m21; }
    if (A.a22 < e) { e = A.a22; nx = V.m02; ny = V.m12; nz = V.m22; }
    const double len = sqrt(nx * nx + ny * ny + nz * nz);
    nx = nx / len; ny = ny / len; nz = nz / len;
}

// Spec S2 (round 3; seven squarings + dominance test: round 4): unit eigenvector of the smallest eigenvalue of the window covariance by power iteration on the adjugate
// (oracle/icp_oracle.c::orc_smallest_evec3 is the same sequence of individually rounded operations): M = adj(C); seven times
// { scale by the exact power of two that brings the trace into [1, 2); M = M * M }; M must then be rank one to 2^-40; the column
// with the largest diagonal entry, normalised.  ~200 fp64 operations, one sqrt and three divisions, against ~1,700 for the eight Jacobi sweeps it replaces
// (k_normals sat on the fp64 floor: 50 us per 640x480 frame, 70 % of it here).  eig3_smallest stays for the plane fits.
constexpr int EVEC_SQUARINGS = 7;       // (round 3: five)
__device__ __forceinline__ bool smallest_evec3(const Sym3 &C, double &nx, double &ny, double &nz)
{
    double m00 = C.a11 * C.a22 - C.a12 * C.a12, m01 = C.a02 * C.a12 - C.a01 * C.a22, m02 = C.a01 * C.a12 - C.a02 * C.a11;
    double m11 = C.a00 * C.a22 - C.a02 * C.a02, m12 = C.a01 * C.a02 - C.a00 * C.a12, m22 = C.a00 * C.a11 - C.a01 * C.a01;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < EVEC_SQUARINGS; ++k) {
        const double tr = (m00 + m11) + m22;
        const int hi = __double2hiint(tr);
        const int be = (hi >> 20) & 0x7ff;
        ok = ok && hi >= 0 && be != 0 && be != 0x7ff;                  // a positive normal number
        const double sc = __hiloint2double((2046 - be) << 20, 0);      // 2^-(be - 1023), exact
        const double a00 = m00 * sc, a01 = m01 * sc, a02 = m02 * sc, a11 = m11 * sc, a12 = m12 * sc, a22 = m22 * sc;
        m00 = (a00 * a00 + a01 * a01) + a02 * a02;
        m01 = (a00 * a01 + a01 * a11) + a02 * a12;
        m02 = (a00 * a02 + a01 * a12) + a02 * a22;
        m11 = (a01 * a01 + a11 * a11) + a12 * a12;
        m12 = (a01 * a02 + a11 * a12) + a12 * a22;
        m22 = (a02 * a02 + a12 * a12) + a22 * a22;
    }
    {   // dominance (spec S2, round 4): M must be rank one to 2^-40, i.e. l1 / l0 above ~1.25 -- else the window has no
        // well-defined direction of least variance and gets no normal (oracle/icp_oracle.c::orc_smallest_evec3)
        const double t = (m00 + m11) + m22;
        const double F = ((m00 * m00 + m11 * m11) + m22 * m22) + 2.0 * ((m01 * m01 + m02 * m02) + m12 * m12);
        ok = ok && F >= (t * t) * (1.0 - 0x1p-40);
    }
    nx = m00; ny = m01; nz = m02;
    double best = m00;
    if (m11 > best) { best = m11; nx = m01; ny = m11; nz = m12; }
    if (m22 > best) { best = m22; nx = m02; ny = m12; nz = m22; }
    const double len = sqrt((nx * nx + ny * ny) + nz * nz);
    ok = ok && len > 0.0 && isfinite(len);
    nx = nx / len; ny = ny / len; nz = nz / len;
    return ok;
}

// ------------------------------------------------------------------------------------ S2
// Spec S2 (round 4b: INTEGER window moments).  Every valid pixel's coordinates are quantised once, Xq = rintf(x * 2^16) (15 um; an
// integer below 2^20, exact in float); a window's moments n, S1 = sum Xq, S2 = sum Xq Xq^T over its valid pixels are exact integers,
// hence order-free, hence a BOX FILTER: column sums of seven rows, then row sums of seven columns -- 14 additions per moment
// instead of 49 (rounds 1-3 accumulated fp64 moments about the centre point in raster order: 880 of the kernel's ~1,750 VALU
// operations per pixel).  C' = n S2 - S1 S1^T (= n^2 2^32 x the covariance, every entry an exact integer below 2^53) goes through
// smallest_evec3; the planar test runs on the quantised coordinates in fp32:
//     |fmaf(nf.z, Zq, fmaf(nf.y, Yq, nf.x * Xq)) - dqf| <= (float)(in_dist * 2^16)
// with nf the float normal that is stored and dqf = (float)((nx S1x/n + ny S1y/n) + nz S1z/n), n / S1 in double.
// oracle/icp_oracle.c::normal_at is the same sequence of operations; tests/test_oracle_independent.py::normals_numpy_full the
// numpy.linalg.eigh restatement of it.
// One thread per target pixel, 32 x 8 pixels per block.  The (32+2r) x (8+2r) quantised neighbourhood is staged in LDS; phase V: one
// work item per (neighbourhood column, output row) sums the 2r+1 rows below it -> LDS (64 B: n and S1 as int32, S2 as six
// doubles); phase H: every thread adds the 2r+1 column sums of its window, solves, and walks the window once for the inliers.
constexpr int NRM_BX = 32, NRM_BY = 8, NRM_RMAX = 4;
struct NrmColSum { int n, sx, sy, sz; double sxx, sxy, sxz, syy, syz, szz; };      // 64 bytes

template <int RT>
__global__ __launch_bounds__(NRM_BX * NRM_BY) void k_normals(FrameTasks a, Geometry g)
{
    __shared__ float4 tile[(NRM_BY + 2 * NRM_RMAX) * (NRM_BX + 2 * NRM_RMAX)];
    __shared__ NrmColSum colsum[NRM_BY * (NRM_BX + 2 * NRM_RMAX)];
    const float4 *__restrict__ cloud = a.t[blockIdx.z].cloud;
    float4 *__restrict__ nrm = a.t[blockIdx.z].nrm;
    const int r = RT > 0 ? RT : g.win_r;
    constexpr int UN = RT > 0 ? 2 * RT + 1 : 1;
    const int tw = NRM_BX + 2 * r, th = NRM_BY + 2 * r;
    const int u0 = blockIdx.x * NRM_BX - r, v0 = blockIdx.y * NRM_BY - r;
    const int tid = threadIdx.y * NRM_BX + threadIdx.x;
    for (int k = tid; k < tw * th; k += NRM_BX * NRM_BY) {
        const int ty = k / tw, tx = k - ty * tw;
        const int uu = u0 + tx, vv = v0 + ty;
        float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);          // an invalid pixel adds zeros everywhere (w = 0: not counted)
        if (uu >= 0 && uu < g.W && vv >= 0 && vv < g.H) {
            const float4 c = cloud[(size_t)vv * g.W + uu];
            if (pt_valid(c.x, c.y, c.z, g.zmax)) q = make_float4(rintf(c.x * 65536.0f), rintf(c.y * 65536.0f), rintf(c.z * 65536.0f), 1.0f);
        }
        tile[k] = q;
    }
    __syncthreads();
    // ---- phase V: (column tx of the neighbourhood, output row oy) -> sums over the rows oy .. oy + 2r
    for (int k = tid; k < tw * NRM_BY; k += NRM_BX * NRM_BY) {
        const int oy = k / tw, tx = k - oy * tw;
        int n = 0;
        double sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
#pragma unroll UN
        for (int dv = 0; dv <= 2 * r; ++dv) {
            const float4 q = tile[(oy + dv) * tw + tx];
            const double x = q.x, y = q.y, z = q.z;               // integers below 2^20: every product and sum below is exact
            n += (int)q.w;
            sx += x; sy += y; sz += z;
            sxx = __fma_rn(x, x, sxx); sxy = __fma_rn(x, y, sxy); sxz = __fma_rn(x, z, sxz);
            syy = __fma_rn(y, y, syy); syz = __fma_rn(y, z, syz); szz = __fma_rn(z, z, szz);
        }
        NrmColSum cs;
        cs.n = n; cs.sx = (int)sx; cs.sy = (int)sy; cs.sz = (int)sz;
        cs.sxx = sxx; cs.sxy = sxy; cs.sxz = sxz; cs.syy = syy; cs.syz = syz; cs.szz = szz;
        colsum[k] = cs;
    }
    __syncthreads();
    const int u = blockIdx.x * NRM_BX + threadIdx.x, v = blockIdx.y * NRM_BY + threadIdx.y;
    if (u >= g.W || v >= g.H) return;
    float4 out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 *__restrict__ win = tile + threadIdx.y * tw + threadIdx.x;      // top-left corner of this pixel's window
    const float4 c0 = win[r * tw + r];
    if (c0.w > 0.5f) {
        // ---- phase H: the window's moments = the 2r+1 column sums of its row
        int n = 0, isx = 0, isy = 0, isz = 0;
        double sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
        const NrmColSum *__restrict__ cs = colsum + threadIdx.y * tw + threadIdx.x;
#pragma unroll UN
        for (int du = 0; du <= 2 * r; ++du) {
            n += cs[du].n; isx += cs[du].sx; isy += cs[du].sy; isz += cs[du].sz;
            sxx += cs[du].sxx; sxy += cs[du].sxy; sxz += cs[du].sxz; syy += cs[du].syy; syz += cs[du].syz; szz += cs[du].szz;
        }
        if (n >= g.min_in) {
            const double dn = (double)n, sx = (double)isx, sy = (double)isy, sz = (double)isz;
            Sym3 C;                                               // n S2 - S1 S1^T: exact integers below 2^53
            C.a00 = dn * sxx - sx * sx; C.a01 = dn * sxy - sx * sy; C.a02 = dn * sxz - sx * sz;
            C.a11 = dn * syy - sy * sy; C.a12 = dn * syz - sy * sz; C.a22 = dn * szz - sz * sz;
            double nx, ny, nz;
            const bool have = smallest_evec3(C, nx, ny, nz);
            if ((nx * (double)c0.x + ny * (double)c0.y) + nz * (double)c0.z > 0.0) { nx = -nx; ny = -ny; nz = -nz; }      // toward the camera
            const double inv = 1.0 / dn;
            const float dqf = (float)((nx * (sx * inv) + ny * (sy * inv)) + nz * (sz * inv));      // the LS plane passes through the window mean
            const float nxf = (float)nx, nyf = (float)ny, nzf = (float)nz;
            const float thr = (float)(g.in_dist * 65536.0);
            int cnt = 0;
#pragma unroll 1
            for (int dv = 0; dv <= 2 * r; ++dv)
#pragma unroll UN
                for (int du = 0; du <= 2 * r; ++du) {
                    const float4 q = win[dv * tw + du];
                    const float e = __fmaf_rn(nzf, q.z, __fmaf_rn(nyf, q.y, nxf * q.x)) - dqf;
                    cnt += (q.w > 0.5f && fabsf(e) <= thr) ? 1 : 0;
                }
            if (have && cnt >= g.min_in) out = make_float4(nxf, nyf, nzf, 1.0f);
        }
    }
    nrm[(size_t)v * g.W + u] = out;
}

// ------------------------------------------------------------------------------------ S3
// Both organized clouds are cut into 8x8-pixel tiles: one wavefront = one tile = 64 slots.
// Source slot id = tile*64 + (v%8)*8 + (u%8).  Correspondences refer to targets by ORIGINAL pixel index j.
constexpr int ACC_R = 16;                      // accumulator replicas per pair: same-address atomics serialise
constexpr int ACC_STRIDE = 40;                 // int64 per replica (NRAW = 36 used)
constexpr int TILE_PX = 8;                 // tile edge in pixels
constexpr int TILE_SLOTS = 64;             // = one wavefront
constexpr int TILE_REC = 72;               // target tile record: 64 slots (4 quadrants x 16) + 4 x (lo, hi) quadrant boxes
constexpr int COARSE_TILES = 8;            // coarse box edge in tiles (64 px)
constexpr int TILES_PER_CHUNK = CHUNK / TILE_SLOTS;

// nchunks = launch blocks of 4 tiles, nslots = padded slot count
// mag_x = ceil(2^32 / x): q = umulhi(n, mag_x) is n / x exactly for n * x < 2^32 (scalar multiply, no VALU division)
struct TileGrid { int ntx, nty, ntiles, ncx, ncy, ncoarse, nchunks, nslots; unsigned int mag_ncx, mag_W, mag_ntx; };
// spec S4c (coarse iterations): the source tiles that take part -- every fourth 8x8-pixel tile, staggered by rows
__device__ __forceinline__ bool coarse_tile(int tx, int ty) { return ((tx + 2 * ty) & 3) == 0; }
__device__ __forceinline__ bool coarse_tile_id(int t, const TileGrid &tg)
{
    const int ty = tg.ntx == 1 ? t : (int)__umulhi((unsigned int)t, tg.mag_ntx);      // t / ntx (2^32 / 1 has no 32-bit magic)
    return coarse_tile(t - ty * tg.ntx, ty);
}

// ---- wave64 cross-lane helpers on the VALU (DPP within a 16-lane row, v_permlane16/32_swap across
// rows -- gfx950); no LDS round trips.  Inputs are never NaN here (+-inf marks "no value").
template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float rdlane(float v, int lane_uniform)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_uniform));
}

// all-lanes min / max: xor-1, xor-2 (quad_perm), row_half_mirror, row_mirror, then rows 0<->1 / 2<->3
// (v_permlane16_swap) and halves (v_permlane32_swap).  The four in-row steps are single v_min/v_max_f32 with a DPP
// source (every lane of these permutations is valid, so no `old` operand is needed); the s_nop covers the two wait
// states a DPP read needs after the VALU write of the same register.  Inputs are never NaN.
#define S3D_DPP4(OP, v)                                                                                              \
    asm("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                             \
        "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                             \
        "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                                 \
        "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"                                           \
        : "+v"(v))
__device__ __forceinline__ float wave_min(float v)
{
    S3D_DPP4("v_min_f32_dpp", v);
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
    asm("v_min_f32 %0, %1, %2" : "=v"(v) : "v"(r[0]), "v"(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    asm("v_min_f32 %0, %1, %2" : "=v"(v) : "v"(r[0]), "v"(r[1]));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
    S3D_DPP4("v_max_f32_dpp", v);
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
    asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(r[0]), "v"(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(r[0]), "v"(r[1]));
    return v;
}

// FOUR all-lanes minima for the price of one: v_permlane32_swap(a, b) lays a's lower / upper halves beside b's, so ONE
// v_min yields min(a[i], a[i+32]) in lanes 0..31 and the same for b in lanes 32..63; v_permlane16_swap does it again for
// two such registers; the four in-row DPP steps finish all four at once.  Result: every lane of row 0 holds min(a), row 1
// min(c), row 2 min(b), row 3 min(d) -- read them with v_readlane at lanes 0 / 16 / 32 / 48.  (A maximum is the negated
// minimum of the negated values.)  10 VALU instructions instead of 40.  Inputs are never NaN.
__device__ __forceinline__ float pair32_min(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(b), false, false);
    float v;
    asm("v_min_f32 %0, %1, %2" : "=v"(v) : "v"(r[0]), "v"(r[1]));
    return v;
}
__device__ __forceinline__ float pair16_min(float x, float y)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(x), __float_as_int(y), false, false);
    float v;
    asm("v_min_f32 %0, %1, %2" : "=v"(v) : "v"(r[0]), "v"(r[1]));
    return v;
}
__device__ __forceinline__ float wave_min_x4(float a, float b, float c, float d)
{
    float x = pair16_min(pair32_min(a, b), pair32_min(c, d));
    S3D_DPP4("v_min_f32_dpp", x);
    return x;      // row 0: a, row 1: c, row 2: b, row 3: d
}

// grid (ntiles, ntasks), block 64: one (frame, role) per blockIdx.y.  Source role: srcT slot w = pixel index (int
// bits) or -1; rows outside [row0,row1) hold no source (dense multi-GPU mode).  Target role: a tile record (TILE_REC
// float4) holds its four 4x4-pixel quadrants, each compacted (valid slots first; a slot is (pixel index, x, y, z))
// into 16 slots, followed by the quadrants' AABBs (lo.xyz, count | hi.xyz); the tile AABB goes to
// box[2t] = (min, count), box[2t+1] = (max, 0).  All boxes are taken from the data (no camera model).
__global__ __launch_bounds__(64) void k_frame_tiles(FrameTasks a, Geometry g, TileGrid tg)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    const FrameTask &ft = a.t[blockIdx.y];
    const int which = ft.role;
    const int tx = t % tg.ntx, ty = t / tg.ntx;
    const int u = tx * TILE_PX + (lane & 7), v = ty * TILE_PX + (lane >> 3);
    const float inf = __int_as_float(0x7f800000);
    float4 q = make_float4(inf, inf, inf, __int_as_float(-1));
    bool ok = false;
    if (u < g.W && v < g.H) {
        const int pix = v * g.W + u;
        const float4 c = ft.cloud[pix];
        ok = pt_valid(c.x, c.y, c.z, g.zmax);
        if (which == 0) ok = ok && v >= ft.row0 && v < ft.row1;
        else if (ok && ft.use_normals) ok = ft.nrm[pix].w > 0.5f;
        if (ok) q = make_float4(c.x, c.y, c.z, __int_as_float(pix));
        if (which != 0) ft.tq[pix] = make_float4(q.w, q.x, q.y, q.z);    // image order, invalid = (-1, +inf, +inf, +inf)
    }
    const unsigned long long m = __ballot(ok);
    const int cnt = __popcll(m);
    if (lane == 0) ft.scount[t] = cnt;                                // summed per coarse cell (no hot atomic)
    if (t == 0 && lane == 0) ft.counts[which] = 0;                    // the role's total: k_coarse_boxes (next launch) adds to it
    if (which == 0) {
        ft.tiles[(size_t)t * TILE_SLOTS + lane] = q;
        return;
    }
    float4 *__restrict__ tgtT = ft.tiles;
    // quadrant of this lane's pixel: (row >= 4) * 2 + (col >= 4); lanes of one quadrant differ in bits 0,1,3,4
    const int qd = ((lane >> 5) << 1) | ((lane >> 2) & 1);
    const unsigned long long qmask = (qd & 1 ? 0xF0F0F0F0ull : 0x0F0F0F0Full) << (qd & 2 ? 32 : 0);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int cntq = __popcll(m & qmask);
    const int rank = ok ? __popcll(m & qmask & below) : cntq + __popcll(~m & qmask & below);
    const size_t base = (size_t)t * TILE_REC;
    // stored as (pixel, x, y, z): the scan's 64-bit key (d2 bits << 32 | pixel) then forms in place -- the
    // distance lands in the register next to the pixel index, no move per candidate
    tgtT[base + qd * 16 + rank] = make_float4(q.w, q.x, q.y, q.z);
    float mnx = ok ? q.x : inf, mny = ok ? q.y : inf, mnz = ok ? q.z : inf;
    float mxx = ok ? q.x : -inf, mxy = ok ? q.y : -inf, mxz = ok ? q.z : -inf;
#pragma unroll
    for (int o = 1; o <= 16; o = (o == 2 ? 8 : o * 2)) {          // xor 1, 2, 8, 16: inside the quadrant
        mnx = fminf(mnx, __shfl_xor(mnx, o)); mny = fminf(mny, __shfl_xor(mny, o)); mnz = fminf(mnz, __shfl_xor(mnz, o));
        mxx = fmaxf(mxx, __shfl_xor(mxx, o)); mxy = fmaxf(mxy, __shfl_xor(mxy, o)); mxz = fmaxf(mxz, __shfl_xor(mxz, o));
    }
    if ((lane & 0x1B) == 0) {                                     // lanes 0, 4, 32, 36: first lane of each quadrant
        tgtT[base + TILE_SLOTS + 2 * qd] = make_float4(mnx, mny, mnz, __int_as_float(cntq));
        tgtT[base + TILE_SLOTS + 2 * qd + 1] = make_float4(mxx, mxy, mxz, 0.0f);
    }
#pragma unroll
    for (int o = 4; o <= 32; o *= 8) {                            // xor 4, 32: across the quadrants
        mnx = fminf(mnx, __shfl_xor(mnx, o)); mny = fminf(mny, __shfl_xor(mny, o)); mnz = fminf(mnz, __shfl_xor(mnz, o));
        mxx = fmaxf(mxx, __shfl_xor(mxx, o)); mxy = fmaxf(mxy, __shfl_xor(mxy, o)); mxz = fmaxf(mxz, __shfl_xor(mxz, o));
    }
    if (lane == 0) {
        ft.tbox[(size_t)t * 2] = make_float4(mnx, mny, mnz, __int_as_float(cnt));
        ft.tbox[(size_t)t * 2 + 1] = make_float4(mxx, mxy, mxz, 0.0f);
    }
}

// grid (ncoarse, ntasks), block 64: totals the valid points of the task's role into the frame's counts (zeroed by
// k_frame_tiles, the launch before); target role: also the AABB of the 8x8 child tiles
__global__ __launch_bounds__(64) void k_coarse_boxes(FrameTasks a, TileGrid tg)
{
    const int c = blockIdx.x, lane = threadIdx.x;
    const FrameTask &ft = a.t[blockIdx.y];
    const int tx = (c % tg.ncx) * COARSE_TILES + (lane & 7), ty = (c / tg.ncx) * COARSE_TILES + (lane >> 3);
    const float inf = __int_as_float(0x7f800000);
    float4 lo = make_float4(inf, inf, inf, 0.0f), hi = make_float4(-inf, -inf, -inf, 0.0f);
    int n = 0;
    if (tx < tg.ntx && ty < tg.nty) {
        const size_t t = (size_t)ty * tg.ntx + tx;
        n = ft.scount[t];
        if (ft.role == 1) { lo = ft.tbox[t * 2]; hi = ft.tbox[t * 2 + 1]; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o);
    if (lane == 0 && n) atomicAdd(ft.counts + ft.role, n);
    if (ft.role == 0) return;
    const float mnx = wave_min(lo.x), mny = wave_min(lo.y), mnz = wave_min(lo.z);
    const float mxx = wave_max(hi.x), mxy = wave_max(hi.y), mxz = wave_max(hi.z);
    if (lane == 0) {
        ft.cbox[(size_t)c * 2] = make_float4(mnx, mny, mnz, 0.0f);
        ft.cbox[(size_t)c * 2 + 1] = make_float4(mxx, mxy, mxz, 0.0f);
    }
}

// start of a run for the pairs [b0, b0 + n): T = T_init (kernel argument) or Identity, trace row 0, flags, clean
// accumulators.  grid (n), block 256.  (slot records / prevq / corr need no reset: the first iteration ignores them.)
constexpr int TINIT_ARGS = 16;
struct TinitArgs { double T[TINIT_ARGS][16]; };
__global__ __launch_bounds__(256) void k_pair_init(TinitArgs ti, int has_T, int b0, double *__restrict__ Tcur,
                                                  double *__restrict__ trace_T, int *__restrict__ flags,
                                                  long long *__restrict__ acc, unsigned int *__restrict__ ticket, int claim_off /* ticket[claim_off + b]: the list kernel's claim counter */, int iters, int nsets,
                                                  StampRing sr /* rows null: no stamps */, int *__restrict__ runs /* the device's run counter; non-null (slam3d_icp_run): one more run in flight */,
                                                  const PairPtrs *__restrict__ gate_pairs /* SLAM3D_PLANE_PAIR_GATE: the pair table (spec S4p association); else null */)
{
    const int k = blockIdx.x, b = b0 + k, lane = threadIdx.x;
    if (gate_pairs && lane >= 64 && lane < 72) {
        // spec S4p, association (oracle/icp_oracle.c::orc_plane_assoc; slam3d_plane_gate's arithmetic): source plane i carried by the run's
        // initial pose, nearest target plane on (a, b, c, d) -- GraphicEnd::match, src/GraphicEnd.cpp:459-484, exact.  (Round 5: here
        // instead of a launch of its own; the planes of both frames were written by launches before this one.)
        const int i = lane - 64;
        const PairPtrs &pp = gate_pairs[b];
        int best = -1;
        if (pp.spl && pp.tpl && i < pp.spl->n) {
            double T[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) T[q] = has_T ? ti.T[k % TINIT_ARGS][q] : ((q % 5 == 0) ? 1.0 : 0.0);
            const FramePlane &P = pp.spl->pl[i];
            const double pa = P.a, pb = P.b, pc = P.c, pd = P.d;
            double n0 = (T[0] * pa + T[1] * pb) + T[2] * pc, n1 = (T[4] * pa + T[5] * pb) + T[6] * pc, n2 = (T[8] * pa + T[9] * pb) + T[10] * pc;
            double dd = pd - ((n0 * T[3] + n1 * T[7]) + n2 * T[11]);
            if (dd < 0.0) { n0 = -n0; n1 = -n1; n2 = -n2; dd = -dd; }        // src/GraphicEnd.cpp:383-387
            const float m0 = (float)n0, m1 = (float)n1, m2 = (float)n2, m3 = (float)dd;
            float bd = __int_as_float(0x7f800000);
            const int n2p = pp.tpl->n;
            for (int j = 0; j < n2p; ++j) {
                const FramePlane &Q = pp.tpl->pl[j];
                float d2 = 0.0f, e;
                e = m0 - Q.a; d2 = __fmaf_rn(e, e, d2);
                e = m1 - Q.b; d2 = __fmaf_rn(e, e, d2);
                e = m2 - Q.c; d2 = __fmaf_rn(e, e, d2);
                e = m3 - Q.d; d2 = __fmaf_rn(e, e, d2);
                if (d2 < bd) { bd = d2; best = j; }
            }
        }
        if (pp.assoc) pp.assoc[i] = best;
    }
    if (runs && k == 0 && lane == 0) atomicAdd(runs, 1);
    if (sr.rows && k == 0 && lane < 64) {       // launch stamps (opt-in, first wave): this run takes the next slot of the ring; start = min -> ~0, end = max -> 0
        const unsigned int run = (*sr.seq + 1u) % (unsigned int)sr.ring;
        unsigned long long *__restrict__ rows = sr.rows + (size_t)run * sr.rows_per_run * STAMP_ROW;
        for (int j = lane; j < sr.rows_per_run * STAMP_ROW; j += 64) rows[j] = (j % STAMP_ROW) < STAMP_R ? ~0ull : 0ull;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) *sr.seq = *sr.seq + 1u;
    }
    {   // every accumulator set of the pair (one per iteration when the solve runs at the head of the next launch)
        longlong2 *__restrict__ a2 = reinterpret_cast<longlong2 *>(acc + (size_t)b * nsets * ACC_R * ACC_STRIDE);
        for (int j = lane; j < nsets * ACC_R * ACC_STRIDE / 2; j += 256) a2[j] = make_longlong2(0, 0);       // (80 KB per pair: four waves)
    }
    if (lane < 16) {
        const double v = has_T ? ti.T[k % TINIT_ARGS][lane] : ((lane % 5 == 0) ? 1.0 : 0.0);
        Tcur[b * 16 + lane] = v;
        trace_T[((size_t)b * (iters + 1)) * 16 + lane] = v;
    }
    // the pose rows the head solves will publish: every entry "not published yet"
    for (int j = 16 + lane; j < (iters + 1) * 16; j += 256)
        reinterpret_cast<unsigned long long *>(trace_T)[(size_t)b * (iters + 1) * 16 + j] = HEAD_EMPTY;
    if (lane == 0) { flags[b] = 0; ticket[b] = 0u; ticket[claim_off + b] = 0u; }
}

// a run that was counted by k_pair_init but whose launches could not all be enqueued: take it out of the count again
__global__ void k_run_uncount(int *__restrict__ runs) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(runs, -1); }

// stable raster-order stream compaction of the valid sources / the eligible targets of each pair.  Only the full brute-force
// modes use these lists.  src w = SLOT id of the pixel, tgt w = pixel index.  Two launches, one 1024-thread block per chunk of 1024
// records (round 5; one block per (pair, role) walked its 300 chunks one dependent load after the other: 327 us per 640x480 pair):
//   k_compact_count     the chunk's number of kept records -> chunk_cnt[(b * 2 + role) * nchunks + chunk]
//   k_compact_scatter   base = the counts of the chunks in front (<= a few hundred integers, summed by the block), then the chunk's
//                       records at base + rank; the last chunk of a role records the total in ccounts[b * 4 + role]
__device__ __forceinline__ bool compact_keep(const float4 *__restrict__ cloud, const float4 *__restrict__ nrm, int i, int i_end, int which,
                                             int use_normals, float zmax, float4 &q)
{
    q = make_float4(0, 0, 0, 0);
    bool ok = false;
    if (i < i_end) {
        q = cloud[i];
        ok = pt_valid(q.x, q.y, q.z, zmax);
        if (ok && which == 1 && use_normals) ok = nrm[i].w > 0.5f;
    }
    return ok;
}
// the block's kept records: rank of this thread's record among them, and their number
__device__ __forceinline__ int compact_rank(bool ok, int &total)
{
    __shared__ int wave_tot[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long m = __ballot(ok);
    if (lane == 0) wave_tot[w] = __popcll(m);
    __syncthreads();
    int woff = 0;
    total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int c = wave_tot[k]; if (k < w) woff += c; total += c; }
    return woff + __popcll(m & ((1ull << lane) - 1ull));
}
// grid (nchunks, 2, B)
__global__ __launch_bounds__(1024) void k_compact_count(const PairPtrs *__restrict__ pairs, int *__restrict__ chunk_cnt, Geometry g,
                                                        int use_normals, int row0, int row1)
{
    const int chunk = blockIdx.x, which = blockIdx.y, b = blockIdx.z;
    const int i_begin = which == 0 ? row0 * g.W : 0, i_end = which == 0 ? row1 * g.W : g.N;
    if (i_begin + chunk * 1024 >= i_end) return;
    float4 q;
    int total;
    compact_rank(compact_keep(which == 0 ? pairs[b].src : pairs[b].tgt, pairs[b].nrm, i_begin + chunk * 1024 + (int)threadIdx.x, i_end, which,
                              use_normals, g.zmax, q), total);
    if (threadIdx.x == 0) chunk_cnt[((size_t)b * 2 + which) * gridDim.x + chunk] = total;
}
__global__ __launch_bounds__(1024) void k_compact_scatter(const PairPtrs *__restrict__ pairs, const int *__restrict__ chunk_cnt,
                                                          float4 *__restrict__ src_c, float4 *__restrict__ tgt_c,
                                                          int *__restrict__ ccounts, Geometry g, TileGrid tg,
                                                          int use_normals, int row0, int row1)
{
    __shared__ int front[16];
    const int chunk = blockIdx.x, which = blockIdx.y, b = blockIdx.z;
    const int i_begin = which == 0 ? row0 * g.W : 0, i_end = which == 0 ? row1 * g.W : g.N;
    const int nch = i_end > i_begin ? (i_end - i_begin + 1023) / 1024 : 0;
    if (chunk >= nch) {
        if (nch == 0 && chunk == 0 && threadIdx.x == 0) ccounts[b * 4 + which] = 0;       // an empty row shard
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = i_begin + chunk * 1024 + tid;
    float4 q;
    const bool ok = compact_keep(which == 0 ? pairs[b].src : pairs[b].tgt, pairs[b].nrm, i, i_end, which, use_normals, g.zmax, q);
    int pre = 0;
    for (int c = tid; c < chunk; c += 1024) pre += chunk_cnt[((size_t)b * 2 + which) * gridDim.x + c];
    for (int o = 32; o >= 1; o >>= 1) pre += __shfl_xor(pre, o);
    if (lane == 0) front[w] = pre;
    int total;
    const int rank = compact_rank(ok, total);            // (its barrier also publishes front[])
    int base = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) base += front[k];
    if (ok) {
        int tag = i;
        if (which == 0) {
            const int v = i / g.W, u = i - v * g.W;
            tag = ((v / TILE_PX) * tg.ntx + u / TILE_PX) * TILE_SLOTS + (v % TILE_PX) * TILE_PX + (u % TILE_PX);
        }
        ((which == 0 ? src_c : tgt_c) + (size_t)b * g.N)[base + rank] = make_float4(q.x, q.y, q.z, __int_as_float(tag));
    }
    if (chunk == nch - 1 && tid == 0) ccounts[b * 4 + which] = base + total;
}

// ------------------------------------------------------------------------------------ S4
struct Rt { float r00, r01, r02, r10, r11, r12, r20, r21, r22, t0, t1, t2; };

__device__ __forceinline__ Rt load_rt(const double *__restrict__ T)
{
    Rt m;
    m.r00 = (float)T[0]; m.r01 = (float)T[1]; m.r02 = (float)T[2];  m.t0 = (float)T[3];
    m.r10 = (float)T[4]; m.r11 = (float)T[5]; m.r12 = (float)T[6];  m.t1 = (float)T[7];
    m.r20 = (float)T[8]; m.r21 = (float)T[9]; m.r22 = (float)T[10]; m.t2 = (float)T[11];
    return m;
}

// the same from LDS, where the block keeps it already rounded to float (wave-uniform values: moved to SGPRs, so the pose
// costs no VGPR across the kernel)
__device__ __forceinline__ float uni_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ Rt load_rt_lds(const float *T)
{
    Rt m;
    m.r00 = uni_f(T[0]); m.r01 = uni_f(T[1]); m.r02 = uni_f(T[2]);  m.t0 = uni_f(T[3]);
    m.r10 = uni_f(T[4]); m.r11 = uni_f(T[5]); m.r12 = uni_f(T[6]);  m.t1 = uni_f(T[7]);
    m.r20 = uni_f(T[8]); m.r21 = uni_f(T[9]); m.r22 = uni_f(T[10]); m.t2 = uni_f(T[11]);
    return m;
}

__device__ __forceinline__ void xform(const Rt &m, float x, float y, float z, float &ox, float &oy, float &oz)
{
    ox = __fmaf_rn(m.r02, z, __fmaf_rn(m.r01, y, m.r00 * x)) + m.t0;
    oy = __fmaf_rn(m.r12, z, __fmaf_rn(m.r11, y, m.r10 * x)) + m.t1;
    oz = __fmaf_rn(m.r22, z, __fmaf_rn(m.r21, y, m.r20 * x)) + m.t2;
}

__device__ __forceinline__ float canon_d2(float px, float py, float pz, float qx, float qy, float qz)
{
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx));
}

__device__ __forceinline__ unsigned long long key_min(unsigned long long a, unsigned long long b);

// ---- the two full-scan kernels start every query from the same upper bound ---------------------------------------
// U = d2 to a target that EXISTS: the previous iteration's match (prevq), else the target at the query's own pixel,
// else the gate.  The bound is a candidate like any other -- its key (d2 bits << 32 | pixel) is what the scan has to
// beat or tie --, so starting from it changes no result; what it buys is that the scan need not keep an argmin per
// candidate: almost nothing it meets is within the bound.
__device__ __forceinline__ unsigned long long brute_bound(bool valid, int slot, float px, float py, float pz,
                                                          const float4 *__restrict__ prevq_b, int first,
                                                          const float4 *__restrict__ tcloud, const float4 *__restrict__ tnrm,
                                                          const Geometry &g, const TileGrid &tg)
{
    unsigned long long bkey = ((unsigned long long)(unsigned int)__float_as_int(g.gate2) << 32) | 0xffffffffull;
    if (valid) {
        float4 pq = prevq_b[slot];
        if (first) pq.w = __int_as_float(-1);                  // a run's first iteration has no previous match
        const int jprev = __float_as_int(pq.w);
        float4 qg = pq;
        int jg = jprev;
        bool tv = jprev >= 0;
        if (!tv) {                                             // same-pixel target as the first guess
            const int tile = slot >> 6, ln = slot & 63;
            const int u = (tile % tg.ntx) * TILE_PX + (ln & 7), v = (tile / tg.ntx) * TILE_PX + (ln >> 3);
            jg = v * g.W + u;
            qg = tcloud[jg];
            tv = pt_valid(qg.x, qg.y, qg.z, g.zmax) && (g.estimator != 0 || tnrm[jg].w > 0.5f);
        }
        const float d2g = canon_d2(px, py, pz, qg.x, qg.y, qg.z);
        if (tv && d2g <= g.gate2) bkey = ((unsigned long long)(unsigned int)__float_as_int(d2g) << 32) | (unsigned int)jg;
    }
    return bkey;
}

// exact brute force on the VALU.  grid = (query blocks, target splits, pairs).  Each thread owns NN_QPT queries
// (registers) taken from the raster-compacted source list; target tiles of NN_TILE points are staged in LDS as
// (pixel, x, y, z) and read with broadcast ds_read_b128.  EVERY source x target distance is evaluated, canonically
// (3 sub, mul, 2 fma).  Round 4, second form: a candidate costs 6.5 VALU operations instead of 7.75 + a half-rate
// v_min_f64 -- per chunk of NV_CH candidates each query keeps only the chunk's smallest d2 (one v_min3_f32 per two
// candidates), and a chunk is rescanned with the packed keys (d2 bits << 32 | pixel, v_min_f64: smallest d2, ties to
// the smallest pixel -- the spec's tie-break, independent of scheduling) only if some query of the wave met a distance
// within its bound (brute_bound above; the bound tightens with every rescan).  With a bound that is the distance to a
// real neighbour a few chunks per query qualify out of ~14,000.  Splits merge through a 64-bit atomicMin on the key;
// best[] is indexed by source SLOT.
// max |q - centre|^2 over a pair's compacted targets (the eps of the expanded-form filters needs it)
__global__ __launch_bounds__(256) void k_qmax2(const float4 *__restrict__ tgt_c, const int *__restrict__ ccounts,
                                               unsigned int *__restrict__ qmax2_bits, int N, float cz)
{
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    float n2 = 0.0f;
    if (j < ccounts[b * 4 + 1]) {
        const float4 q = tgt_c[(size_t)b * N + j];
        const float rz = q.z - cz;
        n2 = __fmaf_rn(rz, rz, __fmaf_rn(q.y, q.y, q.x * q.x));
    }
    for (int o = 32; o >= 1; o >>= 1) n2 = fmaxf(n2, __shfl_xor(n2, o));
    if ((threadIdx.x & 63) == 0 && n2 > 0.0f) atomicMax(qmax2_bits + b, (unsigned int)__float_as_int(n2));   // non-negative floats order like uints
}

// FILT (SLAM3D_VALU_FILTER=1; not the default -- the default evaluates every canonical distance): the chunk minimum is
// taken over the EXPANDED form |q|^2 - 2 p.q (three fmas on centred coordinates, the contraction the matrix-core kernels
// run) and compared with thr = U - |p|^2 + eps -- the same conservative filter as k_nn_mfma, eps included, on the VALU:
// 3.5 operations per candidate.  Flagged chunks are rescanned canonically as before, so the result is the same.
constexpr int NV_CH = 16;
template <bool FILT, int QPT>
__global__ __launch_bounds__(NN_BLOCK) void k_nn_valu(const PairPtrs *__restrict__ pairs,
                                                      const float4 *__restrict__ src_c,
                                                      const float4 *__restrict__ tgt_c,
                                                      const int *__restrict__ ccounts,
                                                      const float4 *__restrict__ prevq,
                                                      const double *__restrict__ Tcur,
                                                      unsigned long long *__restrict__ best, Geometry g, TileGrid tg,
                                                      int nsplit, int first, const unsigned int *__restrict__ qmax2_bits, float cz)
{
    __shared__ float4 tile[NN_TILE];
    __shared__ float4 tileR[FILT ? NN_TILE : 1];               // FILT: (rx, ry, rz, |r|^2) of the same candidates
    const int b = blockIdx.z;
    const int N = g.N, nslots = tg.nslots;
    const int ns = ccounts[b * 4 + 0], nt = ccounts[b * 4 + 1];
    const int q0 = blockIdx.x * (NN_BLOCK * QPT);
    if (q0 >= ns) return;
    const int ntiles = (nt + NN_TILE - 1) / NN_TILE;
    const int tile_begin = (int)(((long long)blockIdx.y * ntiles) / nsplit);
    const int tile_end = (int)(((long long)(blockIdx.y + 1) * ntiles) / nsplit);
    if (tile_begin >= tile_end) return;
    const float4 *__restrict__ S = src_c + (size_t)b * N;
    const float4 *__restrict__ Q = tgt_c + (size_t)b * N;
    const Rt m = load_rt(Tcur + b * 16);
    float px[QPT], py[QPT], pz[QPT];
    unsigned long long bk[QPT];
    unsigned int jb0[QPT];
    int slot[QPT];
    const float inf = __int_as_float(0x7f800000);
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        const int i = q0 + k * NN_BLOCK + threadIdx.x;
        float4 s = make_float4(0, 0, 0, __int_as_float(-1));
        if (i < ns) s = S[i];
        slot[k] = __float_as_int(s.w);
        xform(m, s.x, s.y, s.z, px[k], py[k], pz[k]);
        bk[k] = brute_bound(slot[k] >= 0, slot[k], px[k], py[k], pz[k], prevq + (size_t)b * nslots, first, pairs[b].tgt, pairs[b].nrm, g, tg);
        jb0[k] = (unsigned int)bk[k];                // the bound's own target: a slice that did not improve on it has nothing to merge
    }
    // FILT: a = -2 (p - centre), and what the threshold needs: |p - centre|^2 and the U-independent part of eps (k_nn_mfma's)
    float ax[QPT], ay[QPT], az[QPT], n2p[QPT], epsb[QPT], thr[QPT];
    auto set_thr = [&](int k) __attribute__((always_inline)) {
        const float U = __int_as_float((int)(unsigned int)(bk[k] >> 32));
        thr[k] = slot[k] >= 0 ? (U - n2p[k]) + (epsb[k] + 4.0e-6f * sqrtf(U)) : -1e30f;
    };
    if constexpr (FILT) {
        const float qmax2 = __int_as_float((int)qmax2_bits[b]);
#pragma unroll
        for (int k = 0; k < QPT; ++k) {
            const float rx = px[k], ry = py[k], rz = pz[k] - cz;
            n2p[k] = __fmaf_rn(rz, rz, __fmaf_rn(ry, ry, rx * rx));
            epsb[k] = 2.0e-6f * (n2p[k] + qmax2) + 1.0e-6f;          // (twice k_nn_mfma's factor: here the threshold is not part of the fma chain)
            ax[k] = -2.0f * rx; ay[k] = -2.0f * ry; az[k] = -2.0f * rz;
            set_thr(k);
        }
    }
    for (int t = tile_begin; t < tile_end; ++t) {
        const int j0 = t * NN_TILE;
        __syncthreads();
        for (int k = threadIdx.x; k < NN_TILE; k += NN_BLOCK) {
            const int j = j0 + k;
            float4 q = make_float4(__int_as_float(-1), inf, inf, inf);        // padding: d2 = +inf is within no bound
            float4 qr = make_float4(1e4f, 1e4f, 1e4f, 3e8f);                  // FILT padding: far away, never flagged
            if (j < nt) {
                const float4 c = Q[j];
                q = make_float4(c.w, c.x, c.y, c.z);                          // (pixel, x, y, z)
                if constexpr (FILT) {
                    const float rz = c.z - cz;
                    qr = make_float4(c.x, c.y, rz, __fmaf_rn(rz, rz, __fmaf_rn(c.y, c.y, c.x * c.x)));
                }
            }
            tile[k] = q;
            if constexpr (FILT) tileR[k] = qr;
        }
        __syncthreads();
#pragma unroll 1
        for (int c0 = 0; c0 < NN_TILE; c0 += NV_CH) {
            float mn[QPT];
            bool hit = false;
            if constexpr (FILT) {
#pragma unroll
                for (int jj = 0; jj < NV_CH; jj += 2) {
                    const float4 ca = tileR[c0 + jj], cb = tileR[c0 + jj + 1];
#pragma unroll
                    for (int k = 0; k < QPT; ++k) {
                        const float ea = __fmaf_rn(az[k], ca.z, __fmaf_rn(ay[k], ca.y, __fmaf_rn(ax[k], ca.x, ca.w)));
                        const float eb = __fmaf_rn(az[k], cb.z, __fmaf_rn(ay[k], cb.y, __fmaf_rn(ax[k], cb.x, cb.w)));
                        if (jj == 0) asm("v_min_f32 %0, %1, %2" : "=v"(mn[k]) : "v"(ea), "v"(eb));
                        else asm("v_min3_f32 %0, %0, %1, %2" : "+v"(mn[k]) : "v"(ea), "v"(eb));
                    }
                }
#pragma unroll
                for (int k = 0; k < QPT; ++k) hit = hit || mn[k] <= thr[k];
            } else {
#pragma unroll
                for (int jj = 0; jj < NV_CH; jj += 2) {
                    const float4 ca = tile[c0 + jj], cb = tile[c0 + jj + 1];
#pragma unroll
                    for (int k = 0; k < QPT; ++k) {
                        const float da = canon_d2(px[k], py[k], pz[k], ca.y, ca.z, ca.w);
                        const float db = canon_d2(px[k], py[k], pz[k], cb.y, cb.z, cb.w);
                        if (jj == 0) asm("v_min_f32 %0, %1, %2" : "=v"(mn[k]) : "v"(da), "v"(db));
                        else asm("v_min3_f32 %0, %0, %1, %2" : "+v"(mn[k]) : "v"(da), "v"(db));       // (never NaN: no canonicalisation wanted)
                    }
                }
#pragma unroll
                for (int k = 0; k < QPT; ++k) hit = hit || mn[k] <= __int_as_float((int)(unsigned int)(bk[k] >> 32));
            }
            if (__ballot(hit) != 0ull) {
#pragma unroll 2
                for (int jj = 0; jj < NV_CH; ++jj) {
                    const float4 c = tile[c0 + jj];
#pragma unroll
                    for (int k = 0; k < QPT; ++k) {
                        const float d2 = canon_d2(px[k], py[k], pz[k], c.y, c.z, c.w);
                        bk[k] = key_min(bk[k], ((unsigned long long)(unsigned int)__float_as_int(d2) << 32) | (unsigned int)__float_as_int(c.x));
                    }
                }
                if constexpr (FILT) {
#pragma unroll
                    for (int k = 0; k < QPT; ++k) set_thr(k);          // the bound may have tightened
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < QPT; ++k)
        // (round 5: the bound is a candidate every slice starts from -- only the first non-empty slice merges it, the others only what beat it: an
        //  improvement is a DIFFERENT target.  A settled iteration made nsplit same-address atomics per query of which one mattered:
        //  524 k of them were ~50 us of the 16 k x 15 k scan)
        if (slot[k] >= 0 && (unsigned int)bk[k] != 0xffffffffu && ((unsigned int)bk[k] != jb0[k] || tile_begin == 0)) atomicMin(best + (size_t)b * nslots + slot[k], bk[k]);
}

// ------------------------------------------------------------- full brute force on the matrix cores
// SLAM3D_NN_BRUTE_MFMA: the distance step as a dense contraction on v_mfma_f32_16x16x4_f32.  With
// coordinates taken relative to a fixed centre c, for query i (row) and candidate j (column)
//     D[i][j] = sum_k A[i][k] B[k][j] + C[i],   A[i] = (-2px,-2py,-2pz, 1),  B[.][j] = (qx,qy,qz,|q|^2),
//     C[i] = -(U_i - |p|^2 + eps_i)                     =>   D[i][j] = |p-q|^2 - U_i - eps_i  (+ rounding)
// so D <= 0 flags every pair whose distance can be <= the query's upper bound U_i (previous match /
// same-pixel target / gate).  The MFMA result is only a FILTER: flagged pairs (a few per query) are
// re-evaluated with the canonical fp32 distance and merged with ds_min_u64 on (d2 bits << 32 | pixel),
// so the output is bit-identical to the VALU scan.  eps_i bounds every rounding in the chain
// (DESIGN.md section 5): the k-ordered fma chain of the MFMA, |q|^2, |p|^2, the centring.
// 128 queries per wave (8 row blocks) share each B fragment (4 bytes per lane and group of 16 targets, streamed
// straight from L2 eight groups ahead); the VALU only folds the 32 results per lane with v_min3 and tests the sign.
// This is the f32 form (SLAM3D_MFMA_BF16=0); the default is k_nn_mfma16 further down.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MF_TCH = 256;

// B-layout target arrays for the MFMA scan: rel coords + squared norm, SoA rows of length npad
__global__ __launch_bounds__(256) void k_make_bfrag(const float4 *__restrict__ tgt_c, const int *__restrict__ ccounts,
                                                    float *__restrict__ tgtB, unsigned int *__restrict__ qmax2_bits,
                                                    int N, int npad, float cz)
{
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= npad) return;
    const int nt = ccounts[b * 4 + 1];
    float rx = 1e4f, ry = 1e4f, rz = 1e4f, n2 = 3e8f;                 // padding: far away, never flagged
    if (j < nt) {
        const float4 q = tgt_c[(size_t)b * N + j];
        rx = q.x; ry = q.y; rz = q.z - cz;
        n2 = __fmaf_rn(rz, rz, __fmaf_rn(ry, ry, rx * rx));
        atomicMax(qmax2_bits + b, (unsigned int)__float_as_int(n2));   // non-negative floats order like uints
    }
    float *__restrict__ Bb = tgtB + (size_t)b * 4 * npad;
    Bb[j] = rx; Bb[(size_t)npad + j] = ry; Bb[2 * (size_t)npad + j] = rz; Bb[3 * (size_t)npad + j] = n2;
}

// grid (ceil(N/MF_Q), splits, B), block 64 = ONE independent wave: MF_Q = 128 consecutive compacted sources (8 row
// blocks of 16: eight MFMAs share each B fragment, halving the L2->register stream per flop) against one
// contiguous slice of the targets.  No LDS staging and no barriers: the B fragments (4 bytes per lane and
// group) stream straight from L2 into registers, eight groups ahead; exact candidates are fetched only
// for flagged pairs.  Slices merge with a 64-bit atomicMin on best[] (same key as the VALU kernel).
#ifndef MF_AHEAD_N
#define MF_AHEAD_N 8
#endif
constexpr int MF_AHEAD = MF_AHEAD_N;
constexpr int MF_RB = 8;                 // row blocks (of 16 queries) per wave
constexpr int MF_Q = 16 * MF_RB;         // queries per wave

#ifndef MF_WPE
#define MF_WPE 3
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MF_WPE, MF_WPE))) void k_nn_mfma(const PairPtrs *__restrict__ pairs,
                                                const float4 *__restrict__ src_c, const float4 *__restrict__ tgt_c,
                                                const float *__restrict__ tgtB, const unsigned int *__restrict__ qmax2_bits,
                                                const int *__restrict__ ccounts, const float4 *__restrict__ prevq,
                                                const double *__restrict__ Tcur, unsigned long long *__restrict__ best,
                                                Geometry g, TileGrid tg, int npad, float cz, int nsplit, int first)
{
    __shared__ float4 qpos[MF_Q];
    __shared__ float4 qrel[MF_Q];
    __shared__ float qthr[MF_Q];
    __shared__ unsigned long long qkey[MF_Q];
    const int b = blockIdx.z, lane = threadIdx.x;
    const int N = g.N;
    const int ns = ccounts[b * 4 + 0], nt = ccounts[b * 4 + 1];
    const int i0 = blockIdx.x * MF_Q;
    if (i0 >= ns) return;
    const float4 *__restrict__ Q = tgt_c + (size_t)b * N;
    const float *__restrict__ Bb = tgtB + (size_t)b * 4 * npad;
    const float4 *__restrict__ tcloud = pairs[b].tgt;
    const float4 *__restrict__ tnrm = pairs[b].nrm;
    // ---- each lane prepares MF_Q/64 queries: transformed point, upper bound, filter threshold
    const Rt m = load_rt(Tcur + b * 16);
    const float qmax2 = __int_as_float((int)qmax2_bits[b]);
    int my_slot[MF_Q / 64];
    unsigned int my_j0[MF_Q / 64];             // the target of each query's starting bound (see k_nn_valu: only improvements are merged)
#pragma unroll
    for (int h = 0; h < MF_Q / 64; ++h) {
        const int ql = h * 64 + lane, i = i0 + ql;
        float4 s4 = make_float4(0, 0, 0, __int_as_float(-1));
        if (i < ns) s4 = src_c[(size_t)b * N + i];
        const int slot = __float_as_int(s4.w);
        const bool valid = slot >= 0;
        my_slot[h] = slot;
        float px, py, pz;
        xform(m, s4.x, s4.y, s4.z, px, py, pz);
        const unsigned long long bkey = brute_bound(valid, slot, px, py, pz, prevq + (size_t)b * tg.nslots, first, tcloud, tnrm, g, tg);
        const float U = __int_as_float((int)(unsigned int)(bkey >> 32));
        const float rx = px, ry = py, rz = pz - cz;
        const float n2p = __fmaf_rn(rz, rz, __fmaf_rn(ry, ry, rx * rx));
        const float eps = 1.0e-6f * (n2p + qmax2) + 4.0e-6f * sqrtf(U) + 1.0e-6f;
        qpos[ql] = make_float4(px, py, pz, 0.0f);
        qrel[ql] = make_float4(-2.0f * rx, -2.0f * ry, -2.0f * rz, 1.0f);
        qthr[ql] = valid ? (U - n2p) + eps : -1e30f;           // invalid rows can never be flagged
        qkey[ql] = bkey;
        my_j0[h] = (unsigned int)bkey;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- MFMA operands: A[rb] = component (lane>>4) of query rb*16 + (lane&15); C[rb][r] = -thr of row 4*(lane>>4)+r
    const int kq = lane >> 4, jq = lane & 15;
    float A[MF_RB];
    f32x4 Cc[MF_RB];
#pragma unroll
    for (int rb = 0; rb < MF_RB; ++rb) {
        const float4 a4 = qrel[rb * 16 + jq];
        A[rb] = kq == 0 ? a4.x : (kq == 1 ? a4.y : (kq == 2 ? a4.z : a4.w));
#pragma unroll
        for (int r = 0; r < 4; ++r) Cc[rb][r] = -qthr[rb * 16 + 4 * kq + r];
    }
    // ---- this block's slice of the target groups (16 targets per group; the padded tail is never flagged)
    const int ngroups = (nt + 15) / 16;
    const int g_begin = (int)(((long long)blockIdx.y * ngroups) / nsplit);
    const int g_end = (int)(((long long)(blockIdx.y + 1) * ngroups) / nsplit);
    const float *__restrict__ Bl = Bb + (size_t)kq * npad + jq;             // this lane's stream: Bl[16 * group]
    // exact re-evaluation of a group in which some pair was flagged
    auto flagged = [&](const f32x4 *D, int grp) __attribute__((always_inline)) {
        {
            // exact re-evaluation of the flagged pairs (rare): canonical distance, 64-bit key, LDS atomic min
            const int j = grp * 16 + jq;
            float4 c4 = make_float4(0, 0, 0, __int_as_float(-1));
            if (j < nt) c4 = Q[j];
            // (round 5) every lane first collects WHICH of its 32 accumulators flagged a pair, then the wave loops until no lane has
            // one left: as many trips as the busiest lane has flags (one or two) instead of 32 ballots and branches per flagged group
            // -- on voxel clouds, whose neighbour spacing is of the order of the filter's margin, nearly every group near a query is
            // flagged and this path was the scan's critical one
            unsigned int fm = 0u;
#pragma unroll
            for (int rb = 0; rb < MF_RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) fm |= (D[rb][r] <= 0.0f ? 1u : 0u) << (rb * 4 + r);
            if (!(j < nt)) fm = 0u;
            while (__ballot(fm != 0u) != 0ull) {
                if (fm != 0u) {
                    const int bit = __builtin_ctz(fm);
                    fm &= fm - 1u;
                    const int qi = (bit >> 2) * 16 + 4 * kq + (bit & 3);
                    const float4 p4 = qpos[qi];
                    const float d2 = canon_d2(p4.x, p4.y, p4.z, c4.x, c4.y, c4.z);
                    const unsigned long long key =
                        ((unsigned long long)(unsigned int)__float_as_int(d2) << 32) | (unsigned int)__float_as_int(c4.w);
                    atomicMin(&qkey[qi], key);
                }
            }
        }
    };
    auto fold = [&](const f32x4 *D, int grp) __attribute__((always_inline)) {
        int mn = 0x7fffffff;                                               // "some value <= 0" <=> "min of the bits as int <= 0"
#pragma unroll
        for (int rb = 0; rb < MF_RB; ++rb)
            mn = min(mn, min(min(__float_as_int(D[rb][0]), __float_as_int(D[rb][1])),
                             min(__float_as_int(D[rb][2]), __float_as_int(D[rb][3]))));
        if (__ballot(mn <= 0) != 0ull) flagged(D, grp);
    };
    (void)fold;
#ifndef MFMA_SCHED
#define MFMA_SCHED 1      // 0: the round-3 loop; 1: rotating pipeline (default); 2, 3: the same with that many fold operations PLACED behind every MFMA
#endif
#if MFMA_SCHED == 0
    float bf[MF_AHEAD], bn[MF_AHEAD];
#pragma unroll
    for (int u = 0; u < MF_AHEAD; ++u) bf[u] = Bl[(size_t)16 * min(g_begin + u, ngroups)];     // npad leaves room past the end
    for (int g0 = g_begin; g0 < g_end; g0 += MF_AHEAD) {
#pragma unroll
        for (int u = 0; u < MF_AHEAD; ++u) bn[u] = Bl[(size_t)16 * min(g0 + MF_AHEAD + u, ngroups)];   // next 8 groups in flight
        f32x4 D[2][MF_RB];
#pragma unroll
        for (int rb = 0; rb < MF_RB; ++rb) D[0][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rb], bf[0], Cc[rb], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < MF_AHEAD; ++u) {
            if (u + 1 < MF_AHEAD) {
#pragma unroll
                for (int rb = 0; rb < MF_RB; ++rb)
                    D[(u + 1) & 1][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rb], bf[u + 1], Cc[rb], 0, 0, 0);
            }
            if (g0 + u < g_end) fold(D[u & 1], g0 + u);
        }
#pragma unroll
        for (int u = 0; u < MF_AHEAD; ++u) bf[u] = bn[u];
    }
#else
    // One rotating pipeline: while the VALU folds group G, the matrix core runs group G + 1 and the fragment register
    // that G + 1 just consumed is reloaded with group G + 9 -- one pointer and immediate offsets (the 2 x 256 floats
    // of padding behind every row of tgtB hold far-away points, so nothing is clamped), no second set of fragment
    // registers, sixteen v_min3_i32 per group: 97.4 -> 101.2 TFLOP/s.  Measured and NOT used: placing the fold's VALU
    // operations between the MFMAs with sched_group_barrier (MFMA_SCHED 2 / 3: 92.5 / 95.8 TFLOP/s) -- an issue slot
    // between two back-to-back MFMAs costs the matrix pipe more than the three resident waves' overlap gives back.
    float bf[MF_AHEAD];
    const float *__restrict__ pB = Bl + (size_t)16 * g_begin;
#pragma unroll
    for (int u = 0; u < MF_AHEAD; ++u) bf[u] = pB[16 * u];
    f32x4 D[2][MF_RB];
#pragma unroll
    for (int rb = 0; rb < MF_RB; ++rb) D[0][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rb], bf[0], Cc[rb], 0, 0, 0);
    bf[0] = pB[16 * MF_AHEAD];
    for (int g0 = g_begin; g0 < g_end; g0 += MF_AHEAD, pB += 16 * MF_AHEAD) {
#pragma unroll
        for (int u = 0; u < MF_AHEAD; ++u) {
            const int un = (u + 1) % MF_AHEAD;
#pragma unroll
            for (int rb = 0; rb < MF_RB; ++rb)
                D[(u + 1) & 1][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rb], bf[un], Cc[rb], 0, 0, 0);
            bf[un] = pB[16 * (u + 1 + MF_AHEAD)];                       // the group eight after the one just consumed
            int mn = 0x7fffffff;                                               // "some value <= 0" <=> "min of the bits as int <= 0"
#pragma unroll
            for (int rb = 0; rb < MF_RB; ++rb) {                                // sixteen v_min3_i32
                mn = min(min(mn, __float_as_int(D[u & 1][rb][0])), __float_as_int(D[u & 1][rb][1]));
                mn = min(min(mn, __float_as_int(D[u & 1][rb][2])), __float_as_int(D[u & 1][rb][3]));
            }
#if MFMA_SCHED >= 2
#pragma unroll
            for (int rb = 0; rb < MF_RB; ++rb) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, MFMA_SCHED, 0);    // MFMA_SCHED VALU operations of the fold
            }
#endif
            // (the groups a last trip runs past g_end belong to the next slice or to the padding: scanning them again
            //  changes nothing -- the merge is a minimum -- and the padding is never flagged)
            if (__ballot(mn <= 0) != 0ull) flagged(D[u & 1], g0 + u);
        }
    }
#endif
#pragma unroll
    for (int h = 0; h < MF_Q / 64; ++h) {
        if (my_slot[h] >= 0) {
            const unsigned long long key = qkey[h * 64 + lane];
            if ((unsigned int)(key & 0xffffffffull) != 0xffffffffu && ((unsigned int)key != my_j0[h] || g_begin == 0))
                atomicMin(best + (size_t)b * tg.nslots + my_slot[h], key);
        }
    }
}

// ------------------------------------------------- full brute force on the bf16 matrix cores (round 4)
// gfx950 runs bf16 MFMAs at sixteen times its f32 rate, and a float is EXACTLY the sum of three bf16 numbers
// (h = bf16(v), m = bf16(v - h), l = bf16(v - h - m): 8 + 8 + 8 significand bits).  So the same contraction
//     D[i][j] = |q_j|^2 - 2 p_i . q_j - thr_i           (thr_i = U_i - |p_i|^2 + eps_i;  D <= 0 flags the pair)
// fits ONE v_mfma_f32_16x16x32_bf16 per 16 x 16 tile -- 16 cycles where the f32 form takes 32.  Its K = 32 slots,
// eight per 16-lane group kb:
//     kb = 0, 1, 2 (coordinate c; a = -2 p_c, b = q_c):   A = (a_h, a_h | a_h, a_l | a_m, a_m | 0, 0)
//                                                         B = (b_h, b_m | b_l, b_h | b_h, b_m | 0, 0)
//         -> a_h b_h + a_h b_m + a_h b_l + a_l b_h + a_m b_h + a_m b_m (|m| <= 2^-8 |v|, |l| <= 2^-16 |v|: what is left out,
//            a_m b_l + a_l b_m + a_l b_l, is below 2^-23 |a b|)
//     kb = 3:   A = (1, 1 | 1, t_h | t_m, t_l | 0, 0),  t = -thr;   B = (n_h, n_m | n_l, 1 | 1, 1 | 0, 0),  n = |q|^2
// bf16 x bf16 products are exact in f32; C is the inline constant 0 (no threshold registers).  In memory a target keeps
// 8 bytes per group kb -- (h | m << 16, l | h << 16), resp. (n_h | n_m << 16, n_l | 1.0 << 16) -- and the third register
// of the operand is a copy of the first (kb = 3: the constant (1, 1)): one v_cndmask per fragment, 512 B per group of
// 16 targets.  As in the f32 form the matrix result is only a FILTER: flagged pairs are re-evaluated canonically, so the
// output is bit-identical to the VALU scan as long as eps covers every error of the chain.  Beyond the f32 form's eps
// (roundings of |q|^2, |p|^2, the centring): the dropped cross terms, < 2^-23 |a||b| summed over the coordinates (the
// three-way split itself is exact: tests/test_oracle_math.py), and the matrix core's accumulation -- at most 23
// additions, each bounded by a TRUNCATION (2^-23) of the running magnitude S <= 2 |p||q| + |q|^2 + |thr|, whatever order
// or width the hardware uses: together 24 x 2^-23 S = 2.9e-6 S.
// eps16 = 4e-6 (2 sqrt(|p|^2 max|q|^2) + max|q|^2 + |p|^2 + U) on top.  (Measured, tools/ubench_bf16acc.hip: the matrix core's
// own error on this slot pattern is 2.5 x 2^-24 S at worst -- the model bound has a factor 19 in hand; a GPU test holds it.)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned int bf16_rne(float v)              // the bf16 nearest to a finite v, as its 16 bits
{
    const unsigned int u = (unsigned int)__float_as_int(v);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void bf16_split3(float v, unsigned int &h, unsigned int &m, unsigned int &l)
{
    h = bf16_rne(v);
    const float r1 = v - __int_as_float((int)(h << 16));               // exact: the low bits of v
    m = bf16_rne(r1);
    const float r2 = r1 - __int_as_float((int)(m << 16));              // exact
    l = bf16_rne(r2);
}
constexpr unsigned int BF16_ONE = 0x3f80u;

// B-layout targets of the bf16 scan: tgtB16[b][group][lane = kb * 16 + (j & 15)] = the 8 bytes of target j for group kb
__global__ __launch_bounds__(256) void k_make_bfrag16(const float4 *__restrict__ tgt_c, const int *__restrict__ ccounts,
                                                      uint2 *__restrict__ tgtB16, unsigned int *__restrict__ qmax2_bits,
                                                      int N, int npad, float cz)
{
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= npad) return;
    const int nt = ccounts[b * 4 + 1];
    float r[3] = { 1e4f, 1e4f, 1e4f };                                 // padding: far away, never flagged
    float n2 = 3e8f;
    if (j < nt) {
        const float4 q = tgt_c[(size_t)b * N + j];
        r[0] = q.x; r[1] = q.y; r[2] = q.z - cz;
        n2 = __fmaf_rn(r[2], r[2], __fmaf_rn(r[1], r[1], r[0] * r[0]));
        atomicMax(qmax2_bits + b, (unsigned int)__float_as_int(n2));   // non-negative floats order like uints
    }
    uint2 *__restrict__ out = tgtB16 + ((size_t)b * (npad >> 4) + (j >> 4)) * 64 + (j & 15);
    unsigned int h, m, l;
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
        bf16_split3(r[kb], h, m, l);
        out[kb * 16] = make_uint2(h | (m << 16), l | (h << 16));
    }
    bf16_split3(n2, h, m, l);
    out[3 * 16] = make_uint2(h | (m << 16), l | (BF16_ONE << 16));
}

#ifndef MF16_WPE
#define MF16_WPE 3
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MF16_WPE, MF16_WPE))) void k_nn_mfma16(const PairPtrs *__restrict__ pairs,
                                                const float4 *__restrict__ src_c, const float4 *__restrict__ tgt_c,
                                                const uint2 *__restrict__ tgtB16, const unsigned int *__restrict__ qmax2_bits,
                                                const int *__restrict__ ccounts, const float4 *__restrict__ prevq,
                                                const double *__restrict__ Tcur, unsigned long long *__restrict__ best,
                                                Geometry g, TileGrid tg, int npad, float cz, int nsplit, int first)
{
    __shared__ float4 qpos[MF_Q];
    __shared__ float4 qrel[MF_Q];                   // (-2 rx, -2 ry, -2 rz, -thr)
    __shared__ unsigned long long qkey[MF_Q];
    __shared__ int qslot[MF_Q];                     // each query's slot and the target of its starting bound: only needed again after the scan --
    __shared__ unsigned int qj0[MF_Q];              // in LDS, not in registers across the loop (the kernel sits at its 168-VGPR budget)
    const int b = blockIdx.z, lane = threadIdx.x;
    const int N = g.N;
    const int ns = ccounts[b * 4 + 0], nt = ccounts[b * 4 + 1];
    const int i0 = blockIdx.x * MF_Q;
    if (i0 >= ns) return;
    const float4 *__restrict__ Q = tgt_c + (size_t)b * N;
    const float4 *__restrict__ tcloud = pairs[b].tgt;
    const float4 *__restrict__ tnrm = pairs[b].nrm;
    const Rt m = load_rt(Tcur + b * 16);
    const float qmax2 = __int_as_float((int)qmax2_bits[b]);
#pragma unroll
    for (int h = 0; h < MF_Q / 64; ++h) {
        const int ql = h * 64 + lane, i = i0 + ql;
        float4 s4 = make_float4(0, 0, 0, __int_as_float(-1));
        if (i < ns) s4 = src_c[(size_t)b * N + i];
        const int slot = __float_as_int(s4.w);
        const bool valid = slot >= 0;
        qslot[ql] = slot;
        float px, py, pz;
        xform(m, s4.x, s4.y, s4.z, px, py, pz);
        const unsigned long long bkey = brute_bound(valid, slot, px, py, pz, prevq + (size_t)b * tg.nslots, first, tcloud, tnrm, g, tg);
        const float U = __int_as_float((int)(unsigned int)(bkey >> 32));
        const float rx = px, ry = py, rz = pz - cz;
        const float n2p = __fmaf_rn(rz, rz, __fmaf_rn(ry, ry, rx * rx));
        const float eps = 1.0e-6f * (n2p + qmax2) + 4.0e-6f * sqrtf(U) + 1.0e-6f
                        + 4.0e-6f * (2.0f * sqrtf(n2p * qmax2) + qmax2 + n2p + U);
        const float thr = valid ? (U - n2p) + eps : -1e30f;           // invalid rows can never be flagged
        qpos[ql] = make_float4(px, py, pz, 0.0f);
        qrel[ql] = make_float4(-2.0f * rx, -2.0f * ry, -2.0f * rz, -thr);
        qkey[ql] = bkey;
        qj0[ql] = (unsigned int)bkey;              // (see k_nn_valu: only improvements on the bound are merged)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- A operands: lane (kb = lane >> 4, row = lane & 15) of row block rb holds the eight slots of group kb for query rb * 16 + row
    const int kq = lane >> 4, jq = lane & 15;
    const bool is3 = kq == 3;
    uint4 A[MF_RB];
#pragma unroll
    for (int rb = 0; rb < MF_RB; ++rb) {
        const float4 a4 = qrel[rb * 16 + jq];
        const float v = kq == 0 ? a4.x : (kq == 1 ? a4.y : (kq == 2 ? a4.z : a4.w));
        unsigned int h, mm, l;
        bf16_split3(v, h, mm, l);
        A[rb] = is3 ? make_uint4(BF16_ONE | (BF16_ONE << 16), BF16_ONE | (h << 16), mm | (l << 16), 0u)
                    : make_uint4(h | (h << 16), h | (l << 16), mm | (mm << 16), 0u);
    }
    const int ngroups = (nt + 15) / 16;
    const int g_begin = (int)(((long long)blockIdx.y * ngroups) / nsplit);
    const int g_end = (int)(((long long)(blockIdx.y + 1) * ngroups) / nsplit);
    // ---- flagged groups are QUEUED, not re-evaluated on the spot (round 5).  The exact re-evaluation needs the group's canonical
    // targets Q[j] from memory; inside the scan loop that load is the YOUNGEST in flight, so waiting for it (s_waitcnt vmcnt(0))
    // drained all eight fragment prefetches -- one L2 round trip per flagged group, ~9 % of the groups of a 640x480 scan.  A flagged
    // group now only leaves its per-lane flag mask (one v_alignbit per accumulator: fm = fm << 1 | sign(D); accumulator
    // k = rb * 4 + r ends at bit 31 - k) and its number in an LDS queue; the queue is drained when it is full and after the loop, the
    // Q[j] of all its entries loaded together.  The thresholds sit in the A operands, fixed for the launch, so WHEN a flagged pair is
    // re-evaluated changes nothing.  (The sign bit alone decides: a pair within the bound has D_exact <= -eps, and eps exceeds the
    // proven error of the chain by 1.1e-6 S + 1e-6 -- 4e-6 S against 24 x 2^-23 S, see above --, so its computed D is strictly
    // negative; a D of exactly +0 that sent the wave here belongs to no such pair.)
    constexpr int FQ_CAP = 16;
    __shared__ unsigned int fq_mask[FQ_CAP][64];
    __shared__ int fq_grp[FQ_CAP];
    int fq_n = 0;                                   // wave-uniform
    auto drain = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
        for (int e0 = 0; e0 < fq_n; e0 += 4) {
            float4 c4[4];
            unsigned int fm[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {           // four entries' targets in flight together
                const int e = e0 + u;
                const int j = e < fq_n ? fq_grp[e] * 16 + jq : nt;
                c4[u] = make_float4(0, 0, 0, __int_as_float(-1));
                fm[u] = 0u;
                if (j < nt) { c4[u] = Q[j]; fm[u] = fq_mask[e & (FQ_CAP - 1)][lane]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned int m = fm[u];
                while (__ballot(m != 0u) != 0ull) {
                    if (m != 0u) {
                        const int bit = 31 - __builtin_clz(m);
                        m &= ~(1u << bit);
                        const int k = 31 - bit;
                        const int qi = (k >> 2) * 16 + 4 * kq + (k & 3);
                        const float4 p4 = qpos[qi];
                        const float d2 = canon_d2(p4.x, p4.y, p4.z, c4[u].x, c4[u].y, c4[u].z);
                        const unsigned long long key =
                            ((unsigned long long)(unsigned int)__float_as_int(d2) << 32) | (unsigned int)__float_as_int(c4[u].w);
                        atomicMin(&qkey[qi], key);
                    }
                }
            }
        }
        fq_n = 0;
        __builtin_amdgcn_wave_barrier();
    };
    auto flagged = [&](const f32x4 *D, int grp) __attribute__((always_inline)) {
        unsigned int fm = 0u;
#pragma unroll
        for (int rb = 0; rb < MF_RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) fm = __builtin_amdgcn_alignbit(fm, (unsigned int)__float_as_int(D[rb][r]), 31);
        fq_mask[fq_n][lane] = fm;
        if (lane == 0) fq_grp[fq_n] = grp;
        fq_n += 1;
        if (fq_n == FQ_CAP) drain();
    };
    union Frag { uint4 u; bf16x8 v; };
    auto bfrag = [&](const uint2 ld) __attribute__((always_inline)) {
        Frag f;
        f.u = make_uint4(ld.x, ld.y, is3 ? (BF16_ONE | (BF16_ONE << 16)) : ld.x, 0u);
        return f.v;
    };
    const f32x4 zero = { 0.0f, 0.0f, 0.0f, 0.0f };
    // the rotating pipeline of k_nn_mfma: fold group G while the matrix core runs G + 1; the fragment registers G + 1 just
    // consumed are reloaded with group G + 1 + MF_AHEAD (the padding behind the targets covers the overrun)
    const uint2 *__restrict__ pB = tgtB16 + ((size_t)b * (npad >> 4) + g_begin) * 64 + lane;
    uint2 bf[MF_AHEAD];
    // (the loads are pinned in program order: the loop's s_waitcnt is static, and with the first fragments fetched in any
    //  other order hipcc settles for vmcnt(0) at the loop head -- every trip would drain all eight prefetches)
#pragma unroll
    for (int u = 0; u < MF_AHEAD; ++u) { bf[u] = pB[64 * u]; asm volatile("" ::: "memory"); }
    f32x4 D[2][MF_RB];
    {
        const bf16x8 b0 = bfrag(bf[0]);
#pragma unroll
        for (int rb = 0; rb < MF_RB; ++rb) { Frag a; a.u = A[rb]; D[0][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b0, zero, 0, 0, 0); }
    }
    asm volatile("" ::: "memory");
    bf[0] = pB[64 * MF_AHEAD];
    asm volatile("" ::: "memory");
    for (int g0 = g_begin; g0 < g_end; g0 += MF_AHEAD, pB += 64 * MF_AHEAD) {
#pragma unroll
        for (int u = 0; u < MF_AHEAD; ++u) {
            const int un = (u + 1) % MF_AHEAD;
            const bf16x8 bn = bfrag(bf[un]);
#pragma unroll
            for (int rb = 0; rb < MF_RB; ++rb) { Frag a; a.u = A[rb]; D[(u + 1) & 1][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, bn, zero, 0, 0, 0); }
            bf[un] = pB[64 * (u + 1 + MF_AHEAD)];
            int mn = 0x7fffffff;                                               // "some value <= 0" <=> "min of the bits as int <= 0"
#pragma unroll
            for (int rb = 0; rb < MF_RB; ++rb) {
                mn = min(min(mn, __float_as_int(D[u & 1][rb][0])), __float_as_int(D[u & 1][rb][1]));
                mn = min(min(mn, __float_as_int(D[u & 1][rb][2])), __float_as_int(D[u & 1][rb][3]));
            }
#if defined(MF16_SCHED) && MF16_SCHED > 0
#pragma unroll
            for (int rb = 0; rb < MF_RB; ++rb) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, MF16_SCHED, 0);
            }
#endif
            if (__ballot(mn <= 0) != 0ull) flagged(D[u & 1], g0 + u);
        }
    }
    drain();
#pragma unroll
    for (int h = 0; h < MF_Q / 64; ++h) {
        int le = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));      // the lane id again (one wave per block), and
        asm volatile("" : "+v"(le));               // opaque: the LDS addresses are formed HERE -- hoisted above the loop they (and the lane id) were spilled to scratch
        const int slot = qslot[h * 64 + le];
        if (slot >= 0) {
            const unsigned long long key = qkey[h * 64 + le];
            if ((unsigned int)(key & 0xffffffffull) != 0xffffffffu && ((unsigned int)key != qj0[h * 64 + le] || g_begin == 0))
                atomicMin(best + (size_t)b * tg.nslots + slot, key);
        }
    }
}

// ---- spec S4 (round 4): every correspondence contributes an 8-component INTEGER row vector V, and an iteration's totals are the
// 36 upper-triangle entries of the Gram matrix G = sum V V^T -- exact int64 sums, so neither the order inside the wave, nor which
// block owns which tile, nor the number of GPUs changes a bit:
//   point-to-plane   V = ( rint(a 2^16) [3], rint(n 2^20) [3], rint(b 2^eb), 1 ),  a = p' x n, b = n . (q - p')
//   svd              V = ( rint(p' 2^16) [3], rint(q 2^16) [3], 0, 1 )
// (oracle/icp_oracle.c::row_vector; the 29 doubles of the trace and of the solve are derived from G: derive_sum below.)
// A wave's 64 row vectors meet on the fp64 MATRIX CORES: G is a matrix product, and with integer-valued operands below 2^22
// every product and every partial sum of v_mfma_f64_16x16x4_f64 is exact (64 products of at most 2^44: below 2^51).  Rounds 1-3
// rounded the 29 PRODUCTS instead -- 29 fp64 multiplications, 29 v_rndne_f64 and eight packed wave reductions of 21
// instructions: ~300 of the 805 VALU instructions a wave spends in a settled launch, in a kernel whose stream regime is bound by
// VALU issue (DESIGN.md section 9(0) of round 3; prototype tools/ubench_gram.hip).
struct RowBasis { double v[8]; };
typedef double d4 __attribute__((ext_vector_type(4)));

// b_raw: the unrounded residual n . (q - p') (the optional residual gate tests it)
__device__ __forceinline__ void row_basis(int estimator, double b_scale, float pxf, float pyf, float pzf, const float4 q4, const float4 n4, RowBasis &B, double &b_raw)
{
    const double px = pxf, py = pyf, pz = pzf;
    const double qx = q4.x, qy = q4.y, qz = q4.z;
    b_raw = 0.0;
    if (estimator == 0) {
        const double dx = qx - px, dy = qy - py, dz = qz - pz;
        const double nx = n4.x, ny = n4.y, nz = n4.z;
        B.v[0] = rint((py * nz - pz * ny) * 65536.0); B.v[1] = rint((pz * nx - px * nz) * 65536.0); B.v[2] = rint((px * ny - py * nx) * 65536.0);
        // (n is a float: n 2^20 is exact in float, rintf of it is the same integer as rint of the double -- at the fp32 rate)
        B.v[3] = (double)rintf(n4.x * 1048576.0f); B.v[4] = (double)rintf(n4.y * 1048576.0f); B.v[5] = (double)rintf(n4.z * 1048576.0f);
        b_raw = (nx * dx + ny * dy) + nz * dz;
        B.v[6] = rint(b_raw * b_scale);
    } else {
        B.v[0] = rint(px * 65536.0); B.v[1] = rint(py * 65536.0); B.v[2] = rint(pz * 65536.0);
        B.v[3] = rint(qx * 65536.0); B.v[4] = rint(qy * 65536.0); B.v[5] = rint(qz * 65536.0);
        B.v[6] = 0.0;
    }
    B.v[7] = 1.0;
}

__host__ __device__ __forceinline__ int tri36(int i, int j) { return i * 8 - ((i * (i - 1)) >> 1) + (j - i); }       // (i <= j) of the 8x8 upper triangle, row-major

// The 29 doubles of the trace / the solve from the integer Gram totals (oracle/icp_oracle.c::orc_derive_sums): lane k < 29 gets sum k.
//   point-to-plane   A^T A (r, c) = G[r][c] 2^-(e_r + e_c), e = (16,16,16,20,20,20);  A^T b (r) = G[r][6] 2^-(e_r + eb);  count = G[7][7];
//                    sum b^2 = G[6][6] 2^-2eb
//   svd              sum p' = G[i][7] 2^-16, sum q = G[3+i][7] 2^-16, sum p' q^T (r, c) = G[r][3+c] 2^-32, count = G[7][7],
//                    sum |q - p'|^2 = (G00 + .. + G55 - 2 (G03 + G14 + G25)) 2^-32 in integer arithmetic
__host__ __device__ __forceinline__ double derive_sum(int estimator, int eb, int k, const long long *G)
{
    if (k == 27) return (double)G[35];
    if (estimator == 0) {
        if (k < 21) {
            const int r = k < 6 ? 0 : (k < 11 ? 1 : (k < 15 ? 2 : (k < 18 ? 3 : (k < 20 ? 4 : 5))));
            const int first = r == 0 ? 0 : (r == 1 ? 6 : (r == 2 ? 11 : (r == 3 ? 15 : (r == 4 ? 18 : 20))));
            const int c = r + (k - first);
            return ldexp((double)G[tri36(r, c)], -((r < 3 ? 16 : 20) + (c < 3 ? 16 : 20)));
        }
        if (k < 27) { const int r = k - 21; return ldexp((double)G[tri36(r, 6)], -((r < 3 ? 16 : 20) + eb)); }
        return ldexp((double)G[33], -2 * eb);                          // tri36(6, 6)
    }
    if (k < 6) return ldexp((double)G[tri36(k, 7)], -16);
    if (k < 15) { const int r = (k - 6) / 3, c = (k - 6) - 3 * r; return ldexp((double)G[tri36(r, 3 + c)], -32); }
    if (k < 27) return 0.0;
    const long long d2 = (G[tri36(0, 0)] + G[tri36(1, 1)] + G[tri36(2, 2)]) + (G[tri36(3, 3)] + G[tri36(4, 4)] + G[tri36(5, 5)])
                       - 2 * (G[tri36(0, 3)] + G[tri36(1, 4)] + G[tri36(2, 5)]);
    return ldexp((double)d2, -32);
}

template <int CTRL> __device__ __forceinline__ double dpp_d(double v)      // lanes the permutation does not reach read 0
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// The wave's 64 row vectors -> the pair's accumulators (replica chosen by the caller).  `slab`: 2 KB of LDS of this wave.
//   v_mfma_f64_16x16x4_f64: A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k (one f64 each); D[(lane >> 4) + 4 r][lane & 15] in
//   element r of the lane's four.  G = sum V V^T needs A = B^T = the row vectors, so ONE register serves both operands; and because
//   V has 8 components, not 16, TWO sets of four points share an instruction: rows / columns 0..7 carry component i of point
//   8 s + k, rows / columns 8..15 component i - 8 of point 8 s + 4 + k; the diagonal 8x8 blocks of D are the two sets' Gram sums
//   (the off-diagonal blocks mix the sets and are ignored).  Eight instructions cover the 64 points.  The transpose (lane = point
//   -> lane = component) goes through LDS: 64 bytes per point, half the wave at a time (the slab is what is left of the stage
//   buffers: 2 KB), read back as 512 contiguous bytes per instruction.
__device__ __forceinline__ void tile_accumulate(const RowBasis &B, long long *__restrict__ acc /* [ACC_STRIDE] */, double *slab)
{
    if (__ballot(B.v[7] != 0.0) == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int rd = (4 * ((lane >> 3) & 1) + (lane >> 4)) * 8 + (lane & 7);       // (point within the group of 8) * 8 + component
    d4 D = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                     // (whoever read the slab before is done)
        if ((lane >> 5) == half) {
            double2 *__restrict__ w = reinterpret_cast<double2 *>(slab + (lane & 31) * 8);
            w[0] = make_double2(B.v[0], B.v[1]); w[1] = make_double2(B.v[2], B.v[3]);
            w[2] = make_double2(B.v[4], B.v[5]); w[3] = make_double2(B.v[6], B.v[7]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const double x = slab[s4 * 64 + rd];
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, D, 0, 0, 0);
        }
    }
    // G[i][j] = D[i][j] + D[8 + i][8 + j]: lane (q, j < 8) holds D[q][j], D[q + 4][j]; lane (q, j + 8) holds D[q + 8][j + 8], D[q + 12][j + 8]
    const double g0 = D[0] + dpp_d<0x108>(D[2]);             // row_shl:8 -- lane l reads lane l + 8 of its 16-lane row
    const double g1 = D[1] + dpp_d<0x108>(D[3]);
    const int j = lane & 15, q = lane >> 4;
    if (j < 8) {
        // integers below 2^51: adding 1.5 * 2^52 leaves the two's complement value in the low mantissa bits
        const long long i0 = __double_as_longlong(g0 + 6755399441055744.0) - 0x4338000000000000ll;
        const long long i1 = __double_as_longlong(g1 + 6755399441055744.0) - 0x4338000000000000ll;
        if (q <= j && i0 != 0) atomicAdd(reinterpret_cast<unsigned long long *>(acc + tri36(q, j)), (unsigned long long)i0);
        if (q + 4 <= j && i1 != 0) atomicAdd(reinterpret_cast<unsigned long long *>(acc + tri36(q + 4, j)), (unsigned long long)i1);
    }
}

// decode a packed NN key, apply the gate(s), record the correspondence and form the row products.
// GATED instances also apply the optional gates of spec S4g (point-to-plane only): the squared point-to-plane residual
// e^2 <= resid2 (src/GraphicEnd.cpp~:484-489) and the angle between the rotated source normal and the target normal
// (R n_s).n_t >= min_ncos (role of the RANSAC inlier subset, src/GraphicEnd.cpp:542).  A rejected slot has no
// correspondence, but its nearest neighbour still serves as the next iteration's upper bound (prevq of the brute-force modes, slot_rec of the tile search).
struct SlotGates {
    float resid2, min_ncos;
    const int *assoc;              // plane-pair gate (spec S4p): target plane of every source plane; null = off
    const float4 *snrm;            // source normals, indexed by source pixel
    int spix;                      // this slot's source pixel
    float r[9];                    // the float rotation of the current pose (xform's)
};
template <bool GATED>
__device__ __forceinline__ void finish_slot(bool valid, unsigned long long key, float px, float py, float pz,
                                            const float4 *__restrict__ tcloud, const float4 *__restrict__ tnrm,
                                            float gate2, int estimator, double b_scale, int *__restrict__ corr_out,
                                            float *__restrict__ cd2_out, float4 *__restrict__ prevq_out,
                                            RowBasis &B, bool write_out, int jprev /* match the slot already holds (-2: unknown) */,
                                            const SlotGates *sg = nullptr,
                                            const float4 *__restrict__ tq = nullptr /* image-order (pixel, x, y, z) copy of the target, if the frame has one */,
                                            int *jnn_out = nullptr /* the nearest neighbour the next iteration starts from (-1: none); prevq_out may then be null */)
{
#pragma unroll
    for (int k = 0; k < 8; ++k) B.v[k] = 0.0;
    const int j = (int)(unsigned int)(key & 0xffffffffull);
    const float d2 = __int_as_float((int)(unsigned int)(key >> 32));
    bool ok = valid && (j >= 0) && (d2 <= gate2);
    if constexpr (!GATED) {
        if (write_out) {    // the caller-visible correspondence arrays: only the last iteration's are ever read
            *corr_out = ok ? j : -1;
            *cd2_out = ok ? d2 : __int_as_float(0x7f800000);
        }
    }
    float4 pq = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
    if (ok) {
        // the matched point: from the image-order records the window search staged a moment ago where the frame has them (the same
        // bits as tcloud[j], and the same cache lines as the window: the cloud itself is then never read by the iterations)
        float4 q4;
        if (tq) { const float4 r4 = tq[j]; q4 = make_float4(r4.y, r4.z, r4.w, 0.0f); }
        else q4 = tcloud[j];
        float4 n4 = make_float4(0, 0, 0, 0);
        if (estimator == 0) n4 = tnrm[j];
        double b_raw;
        row_basis(estimator, b_scale, px, py, pz, q4, n4, B, b_raw);
        pq = make_float4(q4.x, q4.y, q4.z, __int_as_float(j));
        if constexpr (GATED) {
            if (estimator == 0) {
                bool keep = true;
                if (sg->resid2 > 0.0f) {
                    const double e = b_raw;                             // b of the row, before its quantisation
                    keep = e * e <= (double)sg->resid2;
                }
                if (keep && sg->min_ncos > 0.0f) {
                    const float4 ns = sg->snrm[sg->spix];
                    const float rx = __fmaf_rn(sg->r[2], ns.z, __fmaf_rn(sg->r[1], ns.y, sg->r[0] * ns.x));
                    const float ry = __fmaf_rn(sg->r[5], ns.z, __fmaf_rn(sg->r[4], ns.y, sg->r[3] * ns.x));
                    const float rz = __fmaf_rn(sg->r[8], ns.z, __fmaf_rn(sg->r[7], ns.y, sg->r[6] * ns.x));
                    const float c = __fmaf_rn(rz, n4.z, __fmaf_rn(ry, n4.y, rx * n4.x));
                    keep = ns.w > 0.5f && c >= sg->min_ncos;
                }
                if (keep && sg->assoc) {
                    // spec S4p: normal.w = 1 + plane (0.75: a window normal on no plane, 0: none => label -1); a plane matches its
                    // associated plane only, clutter only clutter
                    const int ls = (int)sg->snrm[sg->spix].w - 1, lt = (int)n4.w - 1;
                    int want = -1;
                    if (ls >= 0) want = sg->assoc[ls & 7];
                    keep = want == lt;
                }
                if (!keep) {
                    ok = false;
#pragma unroll
                    for (int k = 0; k < 8; ++k) B.v[k] = 0.0;
                }
            }
        }
    }
    if constexpr (GATED) {
        if (write_out) {
            *corr_out = ok ? j : -1;
            *cd2_out = ok ? d2 : __int_as_float(0x7f800000);
        }
    }
    // next iteration's upper bound comes from this point (no dependent gather).  Once the pose has settled most matches
    // repeat: a slot that already holds this very match is not written again (same bits, 16 B of traffic less)
    if (jnn_out) *jnn_out = __float_as_int(pq.w);
    if (prevq_out && __float_as_int(pq.w) != jprev) *prevq_out = pq;
}

// accumulation for the brute-force modes: grid (nchunks, B), one thread per source slot, one wave per tile
__global__ __launch_bounds__(CHUNK) void k_accumulate(const PairPtrs *__restrict__ pairs,
                                                      const double *__restrict__ Tcur,
                                                      unsigned long long *__restrict__ best,
                                                      int *__restrict__ corr, float *__restrict__ cd2,
                                                      float4 *__restrict__ prevq,
                                                      long long *__restrict__ acc, Geometry g, TileGrid tg, int nsets,
                                                      int cmode /* 1: a coarse iteration (spec S4c) -- the slots of the other tiles take no part */)
{
    const int b = blockIdx.y, c = blockIdx.x;
    const int t = c * TILES_PER_CHUNK + (threadIdx.x >> 6);
    if (t >= tg.ntiles) return;
    const int slot = c * CHUNK + threadIdx.x;
    const size_t gs = (size_t)b * tg.nslots + slot;
    const float4 sp = pairs[b].srcT[slot];
    const bool valid = __float_as_int(sp.w) >= 0 && !(cmode == 1 && !coarse_tile_id(t, tg));
    const Rt m = load_rt(Tcur + b * 16);
    float px, py, pz;
    xform(m, sp.x, sp.y, sp.z, px, py, pz);
    const unsigned long long key = best[gs];
    best[gs] = ~0ull;
    RowBasis rb;
    SlotGates sg;
    sg.resid2 = g.resid2; sg.min_ncos = g.min_ncos; sg.snrm = pairs[b].snrm; sg.spix = max(__float_as_int(sp.w), 0);
    sg.assoc = g.pair_gate ? pairs[b].assoc : nullptr;
    sg.r[0] = m.r00; sg.r[1] = m.r01; sg.r[2] = m.r02; sg.r[3] = m.r10; sg.r[4] = m.r11; sg.r[5] = m.r12;
    sg.r[6] = m.r20; sg.r[7] = m.r21; sg.r[8] = m.r22;
    finish_slot<true>(valid, key, px, py, pz, pairs[b].tgt, pairs[b].nrm, g.gate2, g.estimator, g.b_scale, corr + gs, cd2 + gs,
                      prevq + gs, rb, true, -2, &sg);
    __shared__ double gslab[TILES_PER_CHUNK][256];                     // the Gram transpose of tile_accumulate: 2 KB per wave
    tile_accumulate(rb, acc + ((size_t)b * nsets * ACC_R + (c % ACC_R)) * ACC_STRIDE, gslab[threadIdx.x >> 6]);
}

// ------------------------------------------------------------------ S4, tile-pruned exact NN
// A tile of an organized depth image is a small 3-D patch, so its axis-aligned bounding box (taken
// from the DATA, no camera model assumed) is tight.  Every query of a source tile knows an upper
// bound U_i >= d2(i, NN(i)) (distance to the previous iteration's match, else to the target at the
// same pixel, else the max_corr_dist gate), hence every candidate that can win or tie lies in a
// target tile whose box is within sqrt(U_i) of query i.  Only those tiles are scanned -- exhaustively,
// with the canonical distance and the (d2, j) lexicographic minimum -- so the result is bit-identical
// to the full brute-force scan.  Order: the 3x3 tiles at the same image location first (bounds
// shrink), then a two-level sweep (64x64-px coarse boxes -> child tiles) by ballots against the wave's
// AABB; before a surviving tile is scanned each lane re-tests its own point against the tile box.
// Candidates are wave-uniform: the compiler turns their loads into scalar s_load_dwordx16.
__device__ __forceinline__ float box_gap2(const float4 lo, const float4 hi, float qminx, float qminy, float qminz,
                                          float qmaxx, float qmaxy, float qmaxz)
{
    const float gx = fmaxf(0.0f, fmaxf(lo.x - qmaxx, qminx - hi.x));
    const float gy = fmaxf(0.0f, fmaxf(lo.y - qmaxy, qminy - hi.y));
    const float gz = fmaxf(0.0f, fmaxf(lo.z - qmaxz, qminz - hi.z));
    return gx * gx + gy * gy + gz * gz;      // empty boxes (lo=+inf, hi=-inf) give +inf
}

// min of two packed keys (d2 bits << 32 | pixel).  d2 >= 0 is a float, so the high word is below 0x7f800001 and the
// 64 bits are a finite non-negative DOUBLE (a denormal when d2 is tiny); non-negative doubles order like their bit
// patterns, so one v_min_f64 replaces the 64-bit compare and two selects.  (FP64 denormals are always on; inline
// asm keeps the compiler from wrapping the operands in NaN canonicalisation.)
__device__ __forceinline__ unsigned long long key_min(unsigned long long a, unsigned long long b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
    return (unsigned long long)__double_as_longlong(r);
}

// Wave-uniform fetch-and-add on an LDS word: lane 0 performs the atomic, every lane gets the old value.  One opaque
// block on purpose.  The source form `if (lane == 0) old = atomicAdd(p, v); old = readfirstlane(old);` inside a loop
// is fragile: when hipcc (ROCm 7.2) threads the lane-0 branch through the loop, the other lanes spin in an inner
// loop with lane 0 masked off, readfirstlane then returns THEIR stale value and the wave never leaves -- the
// cooperative build hung that way as soon as unrelated code (the debug counters) was compiled out.  Must be called
// with all 64 lanes active and a wave-uniform v.
__device__ __forceinline__ int lds_fetch_add_uniform(int *p, int v)
{
    const unsigned int addr = (unsigned int)(__SIZE_TYPE__)(__attribute__((address_space(3))) int *)p;
    int old;
    unsigned long long save;
    asm volatile("s_mov_b64 %[save], exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "ds_add_rtn_u32 %[old], %[a], %[val]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b64 exec, %[save]"
                 : [old] "=&v"(old), [save] "=&s"(save)
                 : [a] "v"(addr), [val] "v"(v)
                 : "memory");
    return __builtin_amdgcn_readfirstlane(old);
}

// Two builds of the kernel, three staged tile records per wave in both.  <3, 7 waves per SIMD, cooperative>: the four
// waves of a block share their work items (22 KB of LDS per block -- seven blocks fit the 160 KB of a CU --, 72 VGPRs);
// the faster one while a launch holds few pairs (latency bound: the slowest block ends the launch).  <3, 8, not
// cooperative>: every wave sweeps its own cells -- no shared lists, no block barriers, 14 KB of LDS, 49 VGPRs -- at the
// full 8 waves per SIMD; it wins when many pairs fill the chip (throughput bound; from 8 pairs per launch on).
__device__ __forceinline__ double wave_solve_point2plane(const double *tot, const double *sh, int &rc_out);     // section S5 below

// ---- the solve at the HEAD of the next launch (cooperative build, point-to-plane; `head` argument) -------------------
// An iteration used to be two launches: the NN kernel and k_solve_acc (5.7 us of a 29 us iteration, plus two launch
// boundaries).  With `head` set, launch k accumulates into ITS OWN accumulator set (acc[b][k], zeroed once per run by
// k_pair_init, never cleared in between) and launch k+1 begins by turning set k into T_{k+1}:
//   * block (0, b) -- the first workgroup of pair b the dispatcher places -- sums the 16 x 29 replicas, runs the
//     lane-parallel LDL^T (wave_solve_point2plane, the code k_solve_acc runs), writes trace_S[k], the flags and the new
//     pose into trace_T[k+1]: sixteen 8-byte agent-scope stores into entries that k_pair_init reset to HEAD_EMPTY, so
//     every entry validates itself and no store ordering (no L2-wide release / acquire) is needed;
//   * the other blocks' wave 0 polls the row (lane l watches entry l; agent-scope loads, s_sleep between);
//   * a poller that has not seen the flag after HEAD_POLLS tries (block 0 not resident yet -- the dispatch order is an
//     observation, not a contract) simply solves for itself: same inputs, same code, same bits.  Nothing ever depends
//     on another block for CORRECTNESS or termination, only for speed;
//   * every wave of the block then takes T from LDS.
// One k_solve_acc launch remains, after the last NN launch (T_iters, the result record).  Iterations x 2 launches become
// iterations + 1.
constexpr int HEAD_POLLS = 200;      // x (a memory round trip + s_sleep): ~100 us until a poller gives up on block 0 and solves by itself

// ---- clearance certificates: an iteration does not search again what the last one already proved ---------------------
// Once the pose has settled, a query's nearest neighbour does not change from one iteration to the next -- and the kernel can
// PROVE it without searching.  Every slot keeps a CLEARANCE c (metres; the second word of its record, slot_rec):
//   * a slot with a match j:   every other valid target is at least c farther from the query than j is;
//   * a slot without a match:  every valid target is at least c beyond the gate.
// A full search yields it for free: the second-smallest distance among the candidates it scanned (tracked with one v_med3 and
// one v_min per candidate) bounds the scanned ones, and every pruning test of the search is run with its radius inflated by
// CERT_M, so everything it did NOT scan is at least sqrt(best) + CERT_M away.  Between two iterations the query moves by
// delta = |p'_k - p'_(k-1)| (both from the float poses, the second recomputed from the previous pose), so by the triangle
// inequality every distance changes by at most delta and the clearance by at most 2 delta: if c - 2 delta still exceeds the
// rounding slack CERT_TAU of the canonical d2, the argmin -- ties included: the inequality is strict -- is the same target,
// whose key is formed directly from the stored match point.  Such a lane takes no part in the search; a wave whose lanes are all
// certified skips window, tiles and items altogether.  The clearance then shrinks by 2 delta per skipped iteration until a
// search renews it -- without being rewritten: every source tile keeps ONE running total of the largest step its queries
// took per launch (tile_cum), a slot records clearance + 2 x the total at the time of its search, and what is left of it
// later is record - 2 x the current total (4 bytes written per tile and launch instead of 4 per certified slot).  Only
// tracking launches certify and add to the total; a launch that does not track voids every clearance it meets (no
// search of such a launch could produce one anyway: clearances never exceed CERT_M, its poses moved by more) and zeroes the totals.  Exact, and checked the way the pruning is: results bit-identical to the oracle with and without
// (SLAM3D_CERT=0), soak, the tie-heavy duplicate-target cases (a tie has clearance 0: never certified).
constexpr float CERT_M = 5.0e-4f;        // metres added to every pruning radius: the clearance of what a search does not scan (round 5: 0.5 mm, not 0.1 -- under
                                         // BASELINE.md's noise a clearance lasts the rest of the run instead of ~6 launches: +2.5 % stream rate, -2 % latency; 0.2-1.6 mm measured alike)
constexpr float CERT_TRACK_MOTION = 1.0e-3f;   // a launch tracks second-best distances and inflates its radii only once the pose moved less
                                         // than this (metres, at the far end of the depth range) since the previous iteration: while it still
                                         // moves by millimetres nothing a search could certify would survive the next update, tracking costs two
                                         // VALU operations per candidate where the candidates are most numerous, and a pair that does not converge
                                         // at all (the reference's wide-baseline Kinect pair) never pays for it
__device__ __forceinline__ float infl_thr(float U, float cm)       // (sqrt(U) + cm)^2 with room for the roundings of the gap tests and of sqrt
{
    if (cm == 0.0f) return U * 1.00001f + 1e-30f;            // (wave-uniform: launches that do not track pay no square root)
    return (U + (2.0002f * cm) * __builtin_amdgcn_sqrtf(U) + 1.0002f * cm * cm) * 1.00001f + 1e-30f;
}
// Rounding budget: the canonical d2 carries at most 5 roundings (three differences squared, two fmas): relative error < 3.2e-7,
// i.e. 1.6e-7 on the distance, + 1 ulp of v_sqrt_f32: every sqrt(fl d2) below is within 2.5e-7 (relative) of the true distance.
// 1e-6 per quantity leaves a factor 4.  fl(d2(c)) > fl(d2(b)) follows from D(c) - D(b) > 3.5e-7 D(b): cert_tau.
__device__ __forceinline__ float cert_tau(float sq) { return 1.0e-6f * sq + 2.0e-8f; }   // gap (m) that guarantees fl(d2) strictly ordered
// (best, second) of the scanned candidates' d2: second = the median of (best, second, d2) -- inputs never NaN
__device__ __forceinline__ void track2(float &b, float &s2, float d2)
{
    asm("v_med3_f32 %0, %1, %0, %2" : "+v"(s2) : "v"(b), "v"(d2));
    asm("v_min_f32 %0, %0, %1" : "+v"(b) : "v"(d2));
}

constexpr int PROJ_RMAX = 3;         // largest window radius the projective search takes on (7x7 pixels)
constexpr int QSTRIDE = 17;                        // float4 per staged quadrant: 16 candidates + 1 pad: lane-specific reads of
                                                   // different quadrants then fall into different banks
constexpr int STAGE_REC = 4 * QSTRIDE + 8;         // LDS image of a tile record (TILE_REC in global memory)
constexpr int NN_MAX_ITEMS = 32;     // (owner wave, coarse cell) work items shared by the waves of a block (max seen 15; the rest is swept by its owner)
constexpr int NN_MAX_TITEMS = 128;   // (owner wave, target tile) work items: the tiles the cell sweeps found worth scanning (max seen 31)
constexpr int NN_WAVES = 4;          // waves (= owned source tiles) per block (8 measured slower: 2 blocks per CU)

// grid (G, B), block 256 = 4 waves; G = a multiple of 8 >= ntiles/4.  Wave w of block c OWNS one source tile: in the
// cooperative build entry (c >> 3) + w * G/8 of the tile rows of XCD c & 7 (rows r = x mod 8), in the throughput build
// tile c + w * G, then perm[b][c][w] once k_balance has run (cost-balanced).
//   1. every wave: one round of loads, upper bounds; projective window search where every lane of the wave is settled by
//      it (depth-image targets), else scan of the tiles its patch projects onto;
//   2. every wave publishes its queries (point, running key, tight/loose class) in LDS and appends one work
//      item per coarse cell its two query boxes can reach;                                   -- barrier --
//   3. all four waves drain the item list together: an item = sweep one coarse cell for one owner (child
//      boxes -> ballot -> per-lane re-test -> stage tiles in this wave's LDS slab -> scan the needed
//      quadrants), merged into the owner's keys with ds_min_u64;                             -- barrier --
//   4. every wave finishes its own tile: gate, row products, level-1 reduction.
// The result is independent of which wave processes which item (keys are merged by an exact minimum).
// What does not change from launch to launch of a handle lives in CONSTANT memory, one entry per handle (slot given at
// slam3d_icp_create): as kernel arguments these ~60 dwords sat in SGPRs from the first instruction on, and with the 102 a wave
// has the compiler parked them in VGPR lanes and fetched them back with v_readlane wherever they were used -- 160 VALU
// instructions per wave for nothing.  Loads from constant memory are scalar, invariant and re-issued where needed instead.
struct NnStatic {
    const double *Tcur; int *corr; float *cd2; int *cost; long long *acc; long long *dbg;
    double *trace_T, *trace_S; int *flags; float2 *slot_rec; float *tile_cum;
    Geometry g; TileGrid tg; int iters, nsets;
};
constexpr int NN_STATIC_SLOTS = 256;
__constant__ NnStatic c_nn_static[NN_STATIC_SLOTS];

template <int NN_STAGE, int WPE, bool COOP, bool DBG, bool GATED = false>
__global__ __launch_bounds__(64 * NN_WAVES) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_nn_tiles_acc(const PairPtrs *__restrict__ pairs,
                                                        int nn_slot /* the handle's entry of c_nn_static */,
                                                        const int *__restrict__ perm,
                                                        int write_out /* corr / cd2 wanted (last iteration) */,
                                                        int it /* iteration of the run; 0: no previous match to start from */,
                                                        StampRing sring /* launch stamps; rows null (the default): none */, int stamp_idx,
                                                        int head /* solve iteration it-1 at the head of this launch (see above) */,
                                                        int cert /* certify from it >= 1 on (needs `it` = the run's iteration and trace_T[it - 1]) */,
                                                        int cmode /* spec S4c: 1 = a coarse iteration (only the tiles of coarse_tile() take part), 2 = the first full iteration
                                                                     after coarse ones (the other tiles hold no record yet), 0 = neither */,
                                                        const int *__restrict__ dev_runs /* the device's count of runs in flight (a kernel argument: it arrives with the others, no
                                                                     dependent load in front of the head's poll-or-solve decision) */)
{
    // g / tg are used all over the kernel and stay with the compiler; the pointers and counts that only the head, the prologue
    // and the epilogue need are fetched right there through SS(): an opaque copy of the entry's address, so that the loads
    // cannot be merged with the ones of another phase, hoisted to the top and parked in VGPR lanes in between
    const Geometry &g = c_nn_static[nn_slot].g;
    const TileGrid &tg = c_nn_static[nn_slot].tg;
    typedef const __attribute__((address_space(4))) NnStatic *nn_static_ptr;
    auto SS = [&]() __attribute__((always_inline)) {
        unsigned long long a = (unsigned long long)&c_nn_static[nn_slot];
        asm volatile("" : "+s"(a));
        return (nn_static_ptr)a;
    };
    const bool certify = cert && it > 0;                    // slots may carry a clearance from the previous launch (<= 0: none)
    bool trk = false;                                       // this launch tracks (best, second) and inflates its pruning radii: decided below,
    float cm = 0.0f;                                        // once this launch's pose is known (how far it moved since the last one)
    __shared__ float4 stage_all[NN_WAVES][NN_STAGE * STAGE_REC];
    __shared__ int wcost[NN_WAVES];                                    // cycles spent for each owner (all helpers)
    __shared__ float qpos[NN_WAVES][3][TILE_SLOTS];                   // p'.x / .y / .z of each owner's queries (SoA: 3 KB, not 4)
    __shared__ unsigned long long qcls[NN_WAVES][2];                  // lane masks: valid, tight (loose = valid & ~tight)
    __shared__ unsigned long long qkey[NN_WAVES][TILE_SLOTS];
    __shared__ int wcentre[NN_WAVES][8];                               // the step-1 tiles of each owner (NN_STAGE used)
    __shared__ float wbox[NN_WAVES][16];                               // each owner's tight / loose query boxes + flag
    __shared__ int items[NN_MAX_ITEMS];                                // phase A work: (owner wave << 16 | coarse cell)
    __shared__ int titems[NN_MAX_TITEMS];                              // phase B work: (owner wave << 24 | target tile)
    __shared__ int n_items, next_item, n_titems, next_titem;
    __shared__ float head_T[12], prev_T[12];                          // this launch's pose (R | t rows, rounded once to float) and the previous one
    __shared__ unsigned int qbs[NN_WAVES][2][TILE_SLOTS];               // each owner's (best, second) d2 among the candidates scanned for it
    const long long clk0 = DBG ? clock64() : 0;
    const long long rt0 = DBG ? (long long)wall_clock64() : 0;
    long long clk1 = 0, clk2 = 0, clk3 = 0, clkP = 0, clkB1 = 0, clkD = 0, clkM = 0, clkE = 0;
    int n_my_items = 0;
    // per-tile work counters and clocks of the instrumented (DBG) instances; compiled out of the production ones
    int n_scanned = 0, n_cand = 0, n_batches = 0, n_chit = 0, n_fhit = 0, n_refined = 0;
    int dbg_cert = 0;
    long long n_b_hist = 0;     // phase B items this wave processed: count | <=2 | <=4 | <=8 | <=16 active lanes (8 bits each) | sum of active lanes
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.y, c = blockIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ int waves_done;
    unsigned long long *const stamp = stamp_row(sring, stamp_idx);
    if (threadIdx.x == 0) { stamp_start(stamp, c); waves_done = 0; }
    // ownership: the measured-cost balanced assignment once k_balance has run (perm >= 0 tile, -2 none),
    // before that (-1) tiles interleaved over the image bands
    // (the cooperative build always uses the interleaved default: on a stream of distinct pairs the measured-cost deal
    // bought nothing and its kernel cost 12 us per run; no map to load either -- one dependent round trip less per wave)
    const int pt = COOP ? -1 : __builtin_amdgcn_readfirstlane(perm[((size_t)b * gridDim.x + c) * NN_WAVES + w]);
    // default (-1, a handle's first two iterations): interleaved, tile c + w * G -- no locality, but even over the XCDs
    // whatever part of the frame holds the work (row shards of the dense mode)
    int t = pt == -1 ? c + w * (int)gridDim.x : (pt < 0 ? tg.ntiles : pt);
    if constexpr (COOP) {
        // XCD-local AND balanced without a map: workgroup c runs on XCD c % 8 (observed; only speed depends on it), and XCD x
        // owns the tile ROWS r = x (mod 8) -- a uniform sample of the whole frame (depth edges, the row shard of the dense
        // mode), while its L2 sees 3/8 of the target frame (rows r-1 .. r+1) instead of all of it.  Inside the XCD's row list
        // (raster order) wave w of the XCD's i-th block takes entry i + w * G/8.
        // The nty % 8 rows left over after the whole rounds are dealt in eight equal raster chunks, so every XCD owns the same
        // number of tiles (60 rows: 7 rows + half a row each, not 8 rows for four XCDs and 7 for the others).
        const int x = c & 7, k = (c >> 3) + w * ((int)gridDim.x >> 3);
        const int full_rows = tg.nty & ~7, n_full = (full_rows >> 3) * tg.ntx;
        if (k < n_full) {
            const int rl = k / tg.ntx, col = k - rl * tg.ntx;
            t = (x + 8 * rl) * tg.ntx + col;
        } else {
            const int n_rem = (tg.nty - full_rows) * tg.ntx, chunk = (n_rem + 7) >> 3, j = k - n_full;
            const int idx = x * chunk + j;
            t = (j < chunk && idx < n_rem) ? full_rows * tg.ntx + idx : tg.ntiles;
        }
    }
    // spec S4c: in a coarse iteration only every fourth tile takes part (the others' waves own nothing: they still help to drain the
    // block's shared work); in the first full iteration after coarse ones those other tiles start like a run's first iteration
    bool first = it == 0;
    if (cmode == 1) {
        if (t < tg.ntiles && !coarse_tile_id(t, tg)) {
            if (write_out) {            // (only a traced run asks for a coarse iteration's correspondences: none for this tile)
                const nn_static_ptr S0 = SS();
                const size_t g0 = (size_t)b * tg.nslots + (size_t)t * TILE_SLOTS + (threadIdx.x & 63);
                S0->corr[g0] = -1; S0->cd2[g0] = __int_as_float(0x7f800000);
            }
            t = tg.ntiles;
        }
        if constexpr (COOP) {
            // A coarse launch is a quarter of the tiles with the widest searches of a run: its length is the serial instruction
            // stream of its heaviest owners (a 10 cm ball at 2 m meets ~50 target tiles), and what shortens that is more waves per
            // owner.  So the participating tiles are dealt ONE per block -- active tile number c + w G of the raster enumeration
            // below, i.e. wave 0 of block c for every frame whose grid has at least as many blocks as active tiles -- and the block's
            // other three waves own nothing and drain its items.  (Any assignment gives the same bits.)
            const int n_even = (tg.ntx + 3) >> 2, n_odd = (tg.ntx + 1) >> 2;        // tiles per even / odd tile row: tx = 0, 4, .. / 2, 6, ..
            const int A = c + w * (int)gridDim.x;
            const int pr = A / (n_even + n_odd), rem = A - pr * (n_even + n_odd);
            const int ty = rem < n_even ? 2 * pr : 2 * pr + 1, tx = rem < n_even ? 4 * rem : 2 + 4 * (rem - n_even);
            t = ty < tg.nty ? ty * tg.ntx + tx : tg.ntiles;
        }
    } else if (cmode == 2 && t < tg.ntiles) first = first || !coarse_tile_id(t, tg);
    const bool has_tile = t < tg.ntiles;
    const long long cw0 = COOP ? 0 : clock64();         // per-tile cost: input of k_balance (throughput build only)
    float4 *__restrict__ st = stage_all[w];
    const size_t gs = (size_t)b * tg.nslots + (size_t)(has_tile ? t : 0) * TILE_SLOTS + lane;
    const float inf = __int_as_float(0x7f800000);
    const PairPtrs &pp = pairs[b];                    // wave-uniform: scalar loads
    const float4 *__restrict__ tcloud = pp.tgt;
    const float4 *__restrict__ tnrm = pp.nrm;
    const float4 *__restrict__ TB = pp.tbox;
    const float4 *__restrict__ CB = pp.cbox;
    const float4 *__restrict__ TT = pp.tgtT;
    if (threadIdx.x == 0) { n_items = 0; next_item = 0; n_titems = 0; next_titem = 0; }
    if (threadIdx.x < NN_WAVES) wcost[threadIdx.x] = 0;
    // the launch's pose goes through LDS in both cases: read from Tcur (k_solve_acc wrote it), or solved right here
    if (!(COOP && head && it > 0)) { if (threadIdx.x < 12) head_T[threadIdx.x] = (float)SS()->Tcur[b * 16 + threadIdx.x]; }
    if (certify && threadIdx.x >= 64 && threadIdx.x < 76) {
        const nn_static_ptr S0 = SS();
        prev_T[threadIdx.x - 64] = (float)S0->trace_T[((size_t)b * (S0->iters + 1) + (it - 1)) * 16 + (threadIdx.x - 64)];
    }
    if constexpr (COOP) {
        if (head && it > 0) {
            if (w == 0) {
                const nn_static_ptr SH = SS();
                double *__restrict__ const trace_T = SH->trace_T;
                const int iters = SH->iters;
                double *__restrict__ Tnew = trace_T + ((size_t)b * (iters + 1) + it) * 16;
                double *tot = reinterpret_cast<double *>(stage_all[0]);          // 32 + 16 doubles of this wave's (still unused) stage slab
                double *tsh = tot + 32;
                bool have = false;
                // head: 1 poll while other runs are in flight on the device, else solve locally; developer knobs: 2 never poll, 3 always
                if (c != 0 && (head == 3 || (head == 1 && __hip_atomic_load(dev_runs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 1))) {
                    // lane l < 16 watches entry l of the row: every entry is ONE 8-byte agent-scope store of the publisher and was
                    // reset to HEAD_EMPTY (a NaN pattern no arithmetic produces) by k_pair_init, so each entry validates itself --
                    // no ordering between the stores is needed, hence no release / acquire fence (on gfx950 those write back and
                    // invalidate the L2: measured +10 us per launch)
                    unsigned long long v = HEAD_EMPTY;
                    for (int poll = 0; poll < HEAD_POLLS; ++poll) {
                        if (lane < 16) v = __hip_atomic_load(reinterpret_cast<unsigned long long *>(Tnew + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__ballot(lane < 16 && v == HEAD_EMPTY) == 0ull) { have = true; break; }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    if (have && lane < 12) head_T[lane] = (float)__longlong_as_double((long long)v);
                }
                if (!have) {
                    const long long *__restrict__ A = SH->acc + ((size_t)b * SH->nsets + (it - 1)) * ACC_R * ACC_STRIDE;
                    long long *Gs = reinterpret_cast<long long *>(tsh + 16);            // the 36 integer Gram totals (LDS, behind tot and tsh)
                    if (lane < NRAW) {
                        long long q = 0;
#pragma unroll
                        for (int r = 0; r < ACC_R; ++r) q += __hip_atomic_load(A + r * ACC_STRIDE + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        Gs[lane] = q;
                    }
                    if (lane < 16) tsh[lane] = trace_T[((size_t)b * (iters + 1) + (it - 1)) * 16 + lane];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (lane < NSUMS) tot[lane] = derive_sum(0, g.eb, lane, Gs);           // (the head solves point-to-plane only)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    int rc;
                    const double Tn = wave_solve_point2plane(tot, tsh, rc);
                    if (lane < 12) head_T[lane] = (float)Tn;
                    if (c == 0) {                                   // the one block that publishes
                        if (lane < NSUMS) SH->trace_S[((size_t)b * iters + (it - 1)) * NSUMS + lane] = tot[lane];
                        if (lane == 0 && rc != 1) SH->flags[b] = SH->flags[b] | (rc == 2 ? 1 : 2);
                        if (lane < 16) __hip_atomic_store(reinterpret_cast<unsigned long long *>(Tnew + lane), (unsigned long long)__double_as_longlong(Tn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    __syncthreads();
    if (certify) {      // how far can a point have moved since the previous iteration?  |R - R'| (max row sum) x depth range + |t - t'|
        float mv = 0.0f, mt = 0.0f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            mv = fmaxf(mv, (fabsf(uni_f(head_T[4 * r]) - uni_f(prev_T[4 * r])) + fabsf(uni_f(head_T[4 * r + 1]) - uni_f(prev_T[4 * r + 1]))) +
                               fabsf(uni_f(head_T[4 * r + 2]) - uni_f(prev_T[4 * r + 2])));
            mt = fmaxf(mt, fabsf(uni_f(head_T[4 * r + 3]) - uni_f(prev_T[4 * r + 3])));
        }
        trk = mv * g.zmax + mt < g.cert_track;
        cm = trk ? g.cert_m : 0.0f;
    }

    // ---- "current query" context: the wave's own tile in step 1, an item's owner in step 3
    float px = 0.0f, py = 0.0f, pz = 0.0f;
    bool valid = false, tight = false, loose = false;
    unsigned long long bkey = ((unsigned long long)(unsigned int)__float_as_int(g.gate2) << 32) | 0xffffffffull;
    float bsc = __int_as_float(0x7f800000), sec = __int_as_float(0x7f800000);     // (best, second) d2 of the candidates scanned in this context
    int ta[NN_STAGE];                                 // the tiles of the current owner's step 1
#pragma unroll
    for (int k = 0; k < NN_STAGE; ++k) ta[k] = -1;
    int tt[NN_STAGE];
    float4 r[NN_STAGE];

    // Can this lane's ball still reach into the box?  gap_i <= |q_i - p_i| for every q in the box (rounding is
    // monotone), the sum has the canonical association, and the threshold is the lane's current best d2 with a 1e-5
    // margin over the few-ulp differences that remain; a threshold taken earlier is larger, hence still conservative.
    auto lane_thr = [&]() __attribute__((always_inline)) {
        return infl_thr(__int_as_float((int)(unsigned int)(bkey >> 32)), cm);      // (radius + CERT_M: what is NOT scanned has that clearance)
    };
    auto lane_gap_le = [&](const float4 lo, const float4 hi, float thr) __attribute__((always_inline)) {
        const float gx = fmaxf(0.0f, fmaxf(lo.x - px, px - hi.x));
        const float gy = fmaxf(0.0f, fmaxf(lo.y - py, py - hi.y));
        const float gz = fmaxf(0.0f, fmaxf(lo.z - pz, pz - hi.z));
        return valid && __fmaf_rn(gz, gz, __fmaf_rn(gy, gy, gx * gx)) <= thr;
    };
    bool hinted = false;        // step 1 scans tiles that the hint picked: their tile-level test nearly always passes, skip it
    // which quadrants of staged tile k (bits 4k .. 4k+3) this lane has to scan: tile box first (wave level), then its own
    // ball against each quadrant box
    auto quad_mask = [&](int k, int tile) __attribute__((always_inline)) {
        const float thr = lane_thr();          // once per tile: the quadrant tests below may use this (larger) value
        unsigned int m = 0u;
        if (!hinted && __ballot(lane_gap_le(TB[2 * tile], TB[2 * tile + 1], thr)) == 0ull) return m;     // uniform -> scalar loads
        if constexpr (DBG) n_scanned += 1;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float4 lo = st[k * STAGE_REC + 4 * QSTRIDE + 2 * qd], hi = st[k * STAGE_REC + 4 * QSTRIDE + 2 * qd + 1];
            const int cnt = __builtin_amdgcn_readfirstlane(__float_as_int(lo.w));
            if (cnt != 0 && lane_gap_le(lo, hi, thr)) m |= 1u << (4 * k + qd);
        }
        return m;
    };
    // Exactness does not need more: a lane whose ball misses a quadrant's box cannot find a winner or a tie there.  A
    // wave's lanes reach 7.6 quadrants between them but 3.0 at most each (1.4 on average; bench pair, mid-run), so the
    // wave iterates max-over-lanes times instead of union-over-lanes times: 16 candidates per trip, each lane reading
    // ITS quadrant (same slot index, quadrant stride 17 float4: distinct quadrants sit in distinct banks).
    // two copies of the loop, chosen per call (wave-uniform): with and without the (best, second) tracking of the certificates
    auto scan_lanes = [&](unsigned int m) __attribute__((always_inline)) {
        if (trk) {
            while (__ballot(m != 0u) != 0ull) {
                if constexpr (DBG) n_cand += 16;
                if (m != 0u) {
                    const int q = __builtin_ctz(m);
                    m &= m - 1u;
                    const float4 *__restrict__ cand = st + (q >> 2) * STAGE_REC + (q & 3) * QSTRIDE;
                    const int nq = __float_as_int(st[(q >> 2) * STAGE_REC + 4 * QSTRIDE + 2 * (q & 3)].w);     // the quadrant's live slots come first (k_frame_tiles)
#pragma unroll 1
                    for (int i = 0; i < nq; i += 4) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float4 c4 = cand[i + u];
                            const float d2 = canon_d2(px, py, pz, c4.y, c4.z, c4.w);     // record = (pixel, x, y, z)
                            const unsigned long long key =
                                ((unsigned long long)(unsigned int)__float_as_int(d2) << 32) | (unsigned int)__float_as_int(c4.x);
                            bkey = key_min(bkey, key);
                            track2(bsc, sec, d2);
                        }
                    }
                }
            }
        } else {
            while (__ballot(m != 0u) != 0ull) {
                if constexpr (DBG) n_cand += 16;
                if (m != 0u) {
                    const int q = __builtin_ctz(m);
                    m &= m - 1u;
                    const float4 *__restrict__ cand = st + (q >> 2) * STAGE_REC + (q & 3) * QSTRIDE;
                    const int nq = __float_as_int(st[(q >> 2) * STAGE_REC + 4 * QSTRIDE + 2 * (q & 3)].w);     // the quadrant's live slots come first (k_frame_tiles)
                    // the padding slots of a quadrant hold (+inf, +inf, +inf): d2 = +inf never wins (the last group of four may reach into them)
#pragma unroll 1
                    for (int i = 0; i < nq; i += 4) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float4 c4 = cand[i + u];
                            const float d2 = canon_d2(px, py, pz, c4.y, c4.z, c4.w);     // record = (pixel, x, y, z)
                            const unsigned long long key =
                                ((unsigned long long)(unsigned int)__float_as_int(d2) << 32) | (unsigned int)__float_as_int(c4.x);
                            bkey = key_min(bkey, key);
                        }
                    }
                }
            }
        }
    };
    // scan staged tile k (tile id tile, both wave-uniform): tile box first, then every lane the quadrants it reaches
    auto scan_staged = [&](int k, int tile) __attribute__((always_inline)) { scan_lanes(quad_mask(k, tile)); };
    auto fetch_batch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NN_STAGE; ++k)
            r[k] = tt[k] >= 0 ? TT[(size_t)tt[k] * TILE_REC + lane] : make_float4(__int_as_float(-1), inf, inf, inf);
    };
    auto park = [&]() __attribute__((always_inline)) {
        if constexpr (DBG) n_batches += 1;
        // lanes 0..39 fetch the quadrant boxes of the staged tiles (8 float4 per tile) with one load
        int my_tile = -1;
#pragma unroll
        for (int k = 0; k < NN_STAGE; ++k) if ((lane >> 3) == k) my_tile = tt[k];
        float4 qb = make_float4(inf, inf, inf, 0.0f);                    // empty box, count 0
        if (my_tile >= 0) qb = TT[(size_t)my_tile * TILE_REC + TILE_SLOTS + (lane & 7)];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < NN_STAGE; ++k) st[k * STAGE_REC + (lane >> 4) * QSTRIDE + (lane & 15)] = r[k];
        if (lane < NN_STAGE * 8) st[(lane >> 3) * STAGE_REC + 4 * QSTRIDE + (lane & 7)] = qb;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto scan_parked = [&]() __attribute__((always_inline)) {
        unsigned int m = 0u;                    // one owner for the whole batch: one lane loop over the quadrants of all its tiles
#pragma unroll
        for (int k = 0; k < NN_STAGE; ++k)
            if (tt[k] >= 0) m |= quad_mask(k, tt[k]);
        scan_lanes(m);
        __builtin_amdgcn_wave_barrier();
    };
    // (best, second) of what this wave scanned for `owner` joins the owner's: the new best is the minimum, the new second the
    // smallest of both seconds and of the LOSER of the two bests (non-negative floats order like their bit patterns)
    auto merge_bs = [&](int owner) __attribute__((always_inline)) {
        if (bsc < inf) {
            const unsigned int old = atomicMin(&qbs[owner][0][lane], (unsigned int)__float_as_int(bsc));
            const float loser = fmaxf(__int_as_float((int)old), bsc);
            atomicMin(&qbs[owner][1][lane], (unsigned int)__float_as_int(fminf(sec, loser)));
        }
    };
    auto park_and_scan = [&]() __attribute__((always_inline)) { park(); scan_parked(); };
    // The wave-level tests use TWO query boxes: lanes whose bound is already small ("tight", radius below a
    // quarter of the gate) and the rest ("loose": no match yet / far match), so that a few loose lanes do not
    // inflate the search region of the whole wave.
    float qminx, qminy, qminz, qmaxx, qmaxy, qmaxz, lminx, lminy, lminz, lmaxx, lmaxy, lmaxz;
    bool any_loose = false;
    // both classes' boxes AND their thresholds (largest current bound of the class) from packed reductions: seven values
    // per class in two wave_min_x4 (maxima as negated minima)
    auto class_boxes_thr = [&](float &thr_t, float &thr_l) __attribute__((always_inline)) {
        const float cur = __int_as_float((int)(unsigned int)(bkey >> 32));
        const float a = wave_min_x4(tight ? px : inf, tight ? py : inf, tight ? pz : inf, tight ? -px : inf);
        const float c = wave_min_x4(tight ? -py : inf, tight ? -pz : inf, tight ? -cur : 0.0f, 0.0f);
        qminx = rdlane(a, 0); qminy = rdlane(a, 32); qminz = rdlane(a, 16); qmaxx = -rdlane(a, 48);
        qmaxy = -rdlane(c, 0); qmaxz = -rdlane(c, 32);
        thr_t = infl_thr(-rdlane(c, 16), cm);                          // covers the rounding of box_gap2 and of canon_d2, + CERT_M
        any_loose = __ballot(loose) != 0ull;
        lminx = lminy = lminz = inf; lmaxx = lmaxy = lmaxz = -inf;
        thr_l = 0.0f;
        if (any_loose) {
            const float e = wave_min_x4(loose ? px : inf, loose ? py : inf, loose ? pz : inf, loose ? -px : inf);
            const float f = wave_min_x4(loose ? -py : inf, loose ? -pz : inf, loose ? -cur : 0.0f, 0.0f);
            lminx = rdlane(e, 0); lminy = rdlane(e, 32); lminz = rdlane(e, 16); lmaxx = -rdlane(e, 48);
            lmaxy = -rdlane(f, 0); lmaxz = -rdlane(f, 32);
            thr_l = infl_thr(-rdlane(f, 16), cm);
        }
    };
    // gap test of a box against both query boxes with the CURRENT class bounds (empty class -> never hits)
    auto reach = [&](const float4 lo, const float4 hi, float thr_t, float thr_l) __attribute__((always_inline)) {
        bool h = box_gap2(lo, hi, qminx, qminy, qminz, qmaxx, qmaxy, qmaxz) <= thr_t;
        if (any_loose) h = h || box_gap2(lo, hi, lminx, lminy, lminz, lmaxx, lmaxy, lmaxz) <= thr_l;
        return h;
    };
    auto class_thr = [&](float &thr_t, float &thr_l) __attribute__((always_inline)) {
        const float cur = __int_as_float((int)(unsigned int)(bkey >> 32));
        const float m = wave_min_x4(tight ? -cur : 0.0f, loose ? -cur : 0.0f, 0.0f, 0.0f);     // both maxima in one pass
        thr_t = infl_thr(-rdlane(m, 0), cm);                             // covers the rounding of box_gap2 and of canon_d2, + CERT_M
        thr_l = any_loose ? infl_thr(-rdlane(m, 32), cm) : 0.0f;
    };
    // fine level of coarse cell cc, in three pieces so that the cooperative build can keep several loads in flight:
    // (1) the child boxes (lane k holds child tile k of the 8x8 cell) ...
    auto cell_boxes = [&](int cc, float4 &lo, float4 &hi, int &ctx, int &cty) __attribute__((always_inline)) {
        const int ccy = tg.ncx == 1 ? cc : (int)__umulhi((unsigned int)cc, tg.mag_ncx);     // (2^32 / 1 has no 32-bit magic)
        ctx = (cc - ccy * tg.ncx) * COARSE_TILES; cty = ccy * COARSE_TILES;
        const int tx = ctx + (lane & 7), ty = cty + (lane >> 3);
        lo = make_float4(inf, inf, inf, 0); hi = make_float4(-inf, -inf, -inf, 0);
        if (tx < tg.ntx && ty < tg.nty) { lo = TB[2 * (ty * tg.ntx + tx)]; hi = TB[2 * (ty * tg.ntx + tx) + 1]; }
    };
    // (2) ballot against the wave's query boxes, then the per-lane re-test BEFORE anything is fetched: a tile survives
    // only if some lane's own ball reaches its box (lane k2 holds the box of child k2 -> broadcast with v_readlane)
    auto cell_refine = [&](const float4 lo, const float4 hi, int ctx, int cty) __attribute__((always_inline)) {
        const int tx = ctx + (lane & 7), ty = cty + (lane >> 3);
        float thr_t, thr_l;
        class_thr(thr_t, thr_l);
        bool hit2 = false;
        if (tx < tg.ntx && ty < tg.nty) {
            const int tid = ty * tg.ntx + tx;                    // already scanned in step 1?
            bool in_a = false;
#pragma unroll
            for (int k = 0; k < NN_STAGE; ++k) in_a = in_a || tid == ta[k];
            hit2 = !in_a && reach(lo, hi, thr_t, thr_l);
        }
        unsigned long long tm0 = __ballot(hit2), tm = 0ull;
        if constexpr (DBG) { n_chit += 1; n_fhit += __popcll(tm0); }
        const float lthr = lane_thr();                          // (the lane's bound does not change while the cell is refined)
        while (tm0) {
            const int k2 = __builtin_ctzll(tm0);
            tm0 &= tm0 - 1;
            const float4 blo = make_float4(rdlane(lo.x, k2), rdlane(lo.y, k2), rdlane(lo.z, k2), 0.0f);
            const float4 bhi = make_float4(rdlane(hi.x, k2), rdlane(hi.y, k2), rdlane(hi.z, k2), 0.0f);
            if (__ballot(lane_gap_le(blo, bhi, lthr)) != 0ull) tm |= 1ull << k2;
        }
        if constexpr (DBG) n_refined += __popcll(tm);
        return tm;
    };
    // (3) stage + scan the surviving tiles right here (throughput build; list overflow of the cooperative build)
    auto scan_cell_tiles = [&](unsigned long long tm, int ctx, int cty) __attribute__((always_inline)) {
        while (tm) {
#pragma unroll
            for (int k = 0; k < NN_STAGE; ++k) {
                tt[k] = -1;
                if (tm) {
                    const int k2 = __builtin_ctzll(tm);
                    tm &= tm - 1;
                    tt[k] = (cty + (k2 >> 3)) * tg.ntx + ctx + (k2 & 7);
                }
            }
            fetch_batch();
            park_and_scan();
        }
    };
    auto sweep_cell = [&](int cc) __attribute__((always_inline)) {
        float4 lo, hi;
        int ctx, cty;
        cell_boxes(cc, lo, hi, ctx, cty);
        scan_cell_tiles(cell_refine(lo, hi, ctx, cty), ctx, cty);
    };
    // cooperative build: the surviving tiles of a cell become tile items (owner wave << 24 | tile) of the block's
    // shared list; what does not fit is scanned right here (the caller publishes the improved keys).  Returns true
    // when it scanned.  The current query context must be `owner`'s.
    auto append_tiles = [&](int owner, unsigned long long tm, int ctx, int cty) __attribute__((always_inline)) {
        const int cnt = __popcll(tm);
        if (!cnt) return false;
        const int base = lds_fetch_add_uniform(&n_titems, cnt);                // cnt is wave-uniform
        if ((tm >> lane) & 1ull) {                                             // lane k2 writes the entry of child k2
            const int pos = base + __popcll(tm & ((1ull << lane) - 1ull));
            if (pos < NN_MAX_TITEMS) titems[pos] = (owner << 24) | ((cty + (lane >> 3)) * tg.ntx + ctx + (lane & 7));
        }
        if (base + cnt <= NN_MAX_TITEMS) return false;
        int keep = NN_MAX_TITEMS - base; if (keep < 0) keep = 0;
        unsigned long long over = tm;
        for (int k = 0; k < keep; ++k) over &= over - 1;                        // the `keep` lowest set bits were listed
        scan_cell_tiles(over, ctx, cty);
        return true;
    };

    // ================= step 1: own tile =================
    bool certd = false;                               // this lane's result is certified unchanged (no search)
    unsigned long long clear_mask = 0ull;             // lanes whose slot held a clearance when this launch began (an untracked search must void it)
    int cum_bits = 0;                                 // tracking launches: the tile's motion total including this launch (float bits, held in an SGPR across the drain)
    unsigned long long cert_mask = 0ull;
    int own_jprev = -2;                               // the match this lane's slot record holds (-2: nothing known, always write)
    const float4 s4 = has_tile ? pp.srcT[(size_t)t * TILE_SLOTS + lane] : make_float4(0, 0, 0, __int_as_float(-1));
    const int pix = __float_as_int(s4.w);
    const bool own_valid = pix >= 0;
    float opx = 0.0f, opy = 0.0f, opz = 0.0f;
    if (__ballot(own_valid) != 0ull) {
        // first tiles (a hint only -- exactness never depends on it): th = top-left tile | extends in x << 24 | extends in y << 25
        // of the (up to 2x2) block of target tiles the 8x8 source patch lands on; without one: the same image location and its
        // left / right neighbours
        // Both clouds are organized depth images of one camera, so the source patch under the current pose lands where its
        // points project: the pinhole of the frame geometry gives the target pixel of a lane near the patch centre, and the
        // (up to 2x2) block of tiles the 8x8 patch covers around it.  Unlike last iteration's matches this follows the pose:
        // the second and third iteration of a run, where the pose still moves by centimetres, start in the right tiles.
        const Rt m = load_rt_lds(head_T);
        xform(m, s4.x, s4.y, s4.z, px, py, pz);
        int th = -1;
        int rox = 0, roy = 0;                                     // origin of the window region of S3D_PROJ_SEARCH
        {
            const unsigned long long mm = __ballot(own_valid);
            const unsigned long long ctr = mm & 0x0000001818000000ull;      // lanes 27,28,35,36
            const int src_lane = __builtin_ctzll(ctr ? ctr : mm);
            const float hx = rdlane(px, src_lane), hy = rdlane(py, src_lane), hz = rdlane(pz, src_lane);
            if (hz > 0.0f) {
                const float iz = __builtin_amdgcn_rcpf(hz);
                const float uf = (float)g.fx * hx * iz + (float)g.cx, vf = (float)g.fy * hy * iz + (float)g.cy;
                if (uf > -1.0e6f && uf < 1.0e6f && vf > -1.0e6f && vf < 1.0e6f) {
                    const int pu0 = (int)rintf(uf) - (src_lane & 7), pv0 = (int)rintf(vf) - (src_lane >> 3);   // the patch's top-left pixel
                    rox = pu0 - 4; roy = pv0 - 4;
                    if (pu0 + 7 >= 0 && pu0 < g.W && pv0 + 7 >= 0 && pv0 < g.H) {
                        const int u0 = max(0, pu0), u1 = min(g.W - 1, pu0 + 7), v0 = max(0, pv0), v1 = min(g.H - 1, pv0 + 7);
                        const int ax0 = u0 / TILE_PX, ay0 = v0 / TILE_PX;
                        th = (ay0 * tg.ntx + ax0) | ((u1 / TILE_PX > ax0) << 24) | ((v1 / TILE_PX > ay0) << 25);
                    }
                }
            }
        }
        if (th >= 0 && (th & 0xffffff) < tg.ntiles) {
            const int base = th & 0xffffff, fx = (th >> 24) & 1, fy = (th >> 25) & 1;
            tt[0] = base;
            tt[1] = fx ? base + 1 : -1;
            tt[2] = fy ? base + tg.ntx : -1;
            if constexpr (NN_STAGE >= 4) tt[3] = (fx && fy) ? base + tg.ntx + 1 : -1;     // (else: through the items)
        } else {
            const int tx0 = t % tg.ntx, ty0 = t / tg.ntx;
            const int ox[4] = { 0, -1, 1, 0 }, oy[4] = { 0, 0, 0, -1 };                     // (the rest of the ring comes through the items)
#pragma unroll
            for (int k = 0; k < NN_STAGE; ++k) {
                const int tx = tx0 + ox[k], ty = ty0 + oy[k];
                tt[k] = (tx >= 0 && tx < tg.ntx && ty >= 0 && ty < tg.nty) ? ty * tg.ntx + tx : -1;
            }
        }
#pragma unroll
        for (int k = 0; k < NN_STAGE; ++k) ta[k] = tt[k];
        // one round of independent loads: the first tile records are fetched on speculation -- except in a tracking launch (the pose
        // has settled: nearly every wave ends certified or in the window), where the few waves that do scan tiles fetch them then
        const bool spec_fetch = !trk && (first || pp.tq == nullptr || th < 0);      // (where the window search can run it settles most waves from the second launch on)
        if (spec_fetch) fetch_batch();
        float4 pq = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
        float yprev = -1.0f;                                       // the slot's clearance record (<= 0: none; see tile_cum)
        float cum_prev = 0.0f;
        const nn_static_ptr SP = SS();
        float2 *__restrict__ const slot_rec = SP->slot_rec;
        float *__restrict__ const tile_cum = SP->tile_cum;
        if (trk && !first) cum_prev = tile_cum[(size_t)b * tg.ntiles + t];   // (wave-uniform: a scalar load, in flight with the record; a tile without records starts its total over)
        if (!first) {    // (a run's first iteration: whatever an earlier run left there is ignored)
            // the slot's record is 8 bytes (match, clearance); the matched POINT is gathered again -- from the records the window
            // search and the epilogue read anyway -- instead of being kept per slot (16 B read and, where it changed, written
            // per slot and iteration)
            const float2 rec = slot_rec[gs];
            const int jp = __float_as_int(rec.x);
            if (certify) yprev = rec.y;
            if (jp >= 0 && jp < g.W * g.H) {
                if (pp.tq) { const float4 r4 = pp.tq[jp]; pq = make_float4(r4.y, r4.z, r4.w, rec.x); }
                else { const float4 c4 = tcloud[jp]; pq = make_float4(c4.x, c4.y, c4.z, rec.x); }
#ifdef S3D_PREFETCH_NRM
                // the epilogue will gather the NORMAL of the match, and once the pose has settled the match is this one: touch its cache
                // line now, in the same round trip as the match point (one dword, consumed by an empty asm: no register kept, no result used)
                if (g.estimator == 0) { const float tn = tnrm[jp].x; asm volatile("" :: "v"(tn)); }
#endif
            }
        }
        float4 qs = make_float4(0, 0, 0, 0);
        float ws = 1.0f;
        int fb_pix = pix;                 // the pixel of the fallback target below
        if (__float_as_int(pq.w) < 0) {   // no previous match (a run's first iteration; the first full one after coarse ones): fall back to the
            // target at the pixel the query PROJECTS to under the current pose (the frame geometry's pinhole; a bound only, any valid target
            // serves) -- with the identity that is the query's own pixel; after coarse iterations, or from a caller's initial guess, the
            // own pixel is tens of pixels off while the projected one holds a target millimetres away
            int ug = (t % tg.ntx) * TILE_PX + (lane & 7), vg = (t / tg.ntx) * TILE_PX + (lane >> 3);
            if (pz > 0.05f) {
                const float izq = __builtin_amdgcn_rcpf(pz);
                const float uq = (float)g.fx * px * izq + (float)g.cx, vq = (float)g.fy * py * izq + (float)g.cy;
                if (uq > -1.0e6f && uq < 1.0e6f && vq > -1.0e6f && vq < 1.0e6f) { ug = (int)rintf(uq); vg = (int)rintf(vq); }
            }
            fb_pix = vg * g.W + ug;
            if (ug >= 0 && ug < g.W && vg >= 0 && vg < g.H) {
                if (pp.tq) { const float4 r4 = pp.tq[vg * g.W + ug]; qs = make_float4(r4.y, r4.z, r4.w, 0.0f); }     // (+inf where the pixel is no target: eligibility is folded in)
                else { qs = tcloud[vg * g.W + ug]; if (g.estimator == 0) ws = tnrm[vg * g.W + ug].w; }
            }
        }
        valid = own_valid;
        if (!first) own_jprev = __float_as_int(pq.w);
        // ---- upper bound: previous match, else the target at the same pixel, else the gate
        {
            const int jprev = __float_as_int(pq.w);
            const bool have_prev = valid && jprev >= 0;
            const bool tv = have_prev || (valid && pt_valid(qs.x, qs.y, qs.z, g.zmax) && ws > 0.5f);
            const int jg = have_prev ? jprev : fb_pix;
            const float4 qg = have_prev ? pq : qs;
            const float d2g = canon_d2(px, py, pz, qg.x, qg.y, qg.z);
            if (tv && d2g <= g.gate2) bkey = ((unsigned long long)(unsigned int)__float_as_int(d2g) << 32) | (unsigned int)jg;
            // ---- clearance certificate (see CERT_M above): the slot's clearance minus twice the distance the query moved since the
            // last iteration still exceeds the rounding slack => the argmin (or "nothing within the gate") is what it was
            const bool had_clear = yprev > 0.0f;
            clear_mask = __ballot(had_clear);
            if (trk) {
                const Rt mp = load_rt_lds(prev_T);
                float ox_, oy_, oz_;
                xform(mp, s4.x, s4.y, s4.z, ox_, oy_, oz_);
                const float dl = __builtin_amdgcn_sqrtf(canon_d2(px, py, pz, ox_, oy_, oz_)) * 1.000001f + 1.0e-9f;
                // the tile's motion total: no query of this tile has moved farther than cum_now since the stretch of tracking
                // launches began (each launch adds the largest step of its lanes; the factor covers the rounding of the sum)
                const float cum_now = uni_f((cum_prev + wave_max(own_valid ? dl : 0.0f)) * 1.0000003f);
                cum_bits = __builtin_amdgcn_readfirstlane(__float_as_int(cum_now));
                asm volatile("" : "+s"(cum_bits));                           // (else hipcc keeps 2 * cum_now in a VGPR: a scratch spill in the gated instance)
                if (lane == 0) tile_cum[(size_t)b * tg.ntiles + t] = cum_now;
                const float cn = (yprev - 2.0f * cum_now) * (1.0f - 1.0e-6f) - 1.0e-9f;      // what is left of the clearance (rounded down)
                const bool t_match = have_prev && d2g <= g.gate2;          // the previous match, still inside the gate
                const bool t_none = valid && !have_prev;                     // no match last time: clearance is to the gate
                const float base = t_match ? __builtin_amdgcn_sqrtf(d2g) : __builtin_amdgcn_sqrtf(g.gate2);
                certd = had_clear && (t_match || t_none) && cn > cert_tau(base * 1.000001f);
                if constexpr (DBG) {     // why lanes are not certified: no clearance at all | match left the gate | clearance used up
                    const bool nc0 = valid && !certd && !had_clear, nc1 = valid && !certd && had_clear && have_prev && !(d2g <= g.gate2),
                               nc2 = valid && !certd && had_clear && !(have_prev && !(d2g <= g.gate2));
                    dbg_cert = __popcll(__ballot(nc0)) | (__popcll(__ballot(nc1)) << 8) | (__popcll(__ballot(nc2)) << 16) | (__popcll(__ballot(valid)) << 24);
                }
                if (certd && t_none) bkey = ((unsigned long long)(unsigned int)__float_as_int(g.gate2) << 32) | 0xffffffffull;     // (not even the same-pixel target)
            } else if (cert && lane == 0) tile_cum[(size_t)b * tg.ntiles + t] = 0.0f;       // no tracking: every clearance of the tile is voided below, the total starts over
        }
        cert_mask = __ballot(certd);
        valid = own_valid && !certd;                               // certified lanes take no part in the search; their key stands
        if constexpr (DBG) clk1 = clock64();
        hinted = th >= 0;
        // ---- projective window search.  The target cloud is OUR back-projection of a depth image: the target of pixel
        // (u, v) is z * (a, b, 1) with a = (u - cx) / fx, b = (v - cy) / fy, up to float rounding.  For a query p' (z' > 0)
        // that projects to the real pixel (u*, v*), the distance to ANY point of the ray of pixel (u, v) is
        //     |p' x r| / |r|  >=  z' * sqrt(da^2 + db^2) / sqrt(1 + a^2 + b^2),   da = (u - u*) / fx, db = (v - v*) / fy,
        // so every target outside the (2r+1)^2 window around the rounded projection is farther than
        // z' * (r + 0.49) / (fmax * sqrt(K)), K = max of 1 + a^2 + b^2 over the image.  A lane whose current bound U fits a
        // window of radius <= PROJ_RMAX therefore finds its exact neighbour (ties included: everything outside is strictly
        // farther, with a 0.1 % + 10 um margin over every rounding) by looking at that window alone: ring by ring, the
        // radius shrinking with the bound.  Mid-run 3/4 of the lanes and 5/8 of the waves are settled this way with ~22
        // candidates; settled lanes take no part in the tile search below, and a wave without other lanes skips it.
        // The windows of a wave's lanes lie in the 15x15-pixel region around the projected patch (8 pixels + 3 on one side, 4
        // on the other); the region is staged once in the wave's LDS slab (four coalesced loads per lane, one round trip) and
        // probed from there with lane-specific addresses; a lane whose window leaves the region is simply not settled here.
        bool done = false;
        if (pp.tq != nullptr && th >= 0 && __ballot(valid) != 0ull) {
            const float4 *__restrict__ TQ = pp.tq;
            constexpr int RW = 15;                                       // region edge; RW * RW float4 fit the stage slab
            static_assert(RW * RW <= NN_STAGE * STAGE_REC, "the window region must fit the wave's stage slab");
            bool sane = valid && pz > 0.05f;
            const float izp = __builtin_amdgcn_rcpf(sane ? pz : 1.0f);      // (1 ulp: the 0.1 % margin of proj_c covers it, and the 0.49 the pixel)
            const float uf = (float)g.fx * px * izp + (float)g.cx, vf = (float)g.fy * py * izp + (float)g.cy;
            sane = sane && uf > -1.0e6f && uf < 1.0e6f && vf > -1.0e6f && vf < 1.0e6f;
            const int ru = (sane ? (int)rintf(uf) : 0) - rox, rv = (sane ? (int)rintf(vf) : 0) - roy;     // region coordinates of the window centre
            const float kz = g.proj_c * izp;
            auto radius = [&]() __attribute__((always_inline)) {         // window radius the lane's CURRENT bound needs
                const float U = __int_as_float((int)(unsigned int)(bkey >> 32));
                return kz * (__builtin_amdgcn_sqrtf(U) + (1.0e-5f + 1.0002f * cm)) - 0.49f;       // (+ CERT_M: the clearance of everything outside the window)
            };
            float rn = sane ? radius() : 1.0e9f;
            int r_l = rn > 0.0f ? (int)ceilf(fminf(rn, 1.0e6f)) : 0;
            const bool settled = sane && r_l <= PROJ_RMAX && ru - r_l >= 0 && ru + r_l < RW && rv - r_l >= 0 && rv + r_l < RW;
            // all or nothing: a wave with even one lane left for the tile search pays that search in full anyway (its cost is the
            // maximum over the lanes), so the window phase only runs where it replaces it
            if (__ballot(valid && !settled) == 0ull) {
                if (!settled) r_l = -1;
#pragma unroll
                for (int i = 0; i < (RW * RW + 63) / 64; ++i) {
                    const int idx = lane + 64 * i, ry = idx / RW, rx = idx - ry * RW;
                    const int u = rox + rx, v = roy + ry;
                    float4 c4 = make_float4(__int_as_float(-1), inf, inf, inf);
                    if (idx < RW * RW && u >= 0 && u < g.W && v >= 0 && v < g.H) c4 = TQ[v * g.W + u];
                    if (idx < RW * RW) st[idx] = c4;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const float4 *__restrict__ wc = st + rv * RW + ru;        // the lane's window centre in the slab
                auto probe = [&](int dx, int dy, bool on) __attribute__((always_inline)) {
                    if (on) {
                        const float4 c4 = wc[dy * RW + dx];
                        const float d2 = canon_d2(px, py, pz, c4.y, c4.z, c4.w);          // record = (pixel, x, y, z)
                        const unsigned long long key =
                            ((unsigned long long)(unsigned int)__float_as_int(d2) << 32) | (unsigned int)__float_as_int(c4.x);
                        bkey = key_min(bkey, key);
                        track2(bsc, sec, d2);
                    }
                };
                probe(0, 0, r_l >= 0);
                for (int rho = 1; rho <= PROJ_RMAX; ++rho) {
                    rn = radius();
                    if (settled) r_l = min(r_l, rn > 0.0f ? (int)ceilf(rn) : 0);
                    const bool on = r_l >= rho;
                    if (__ballot(on) == 0ull) break;
                    if constexpr (DBG) n_cand += 8 * rho;
#pragma unroll 1
                    for (int k = 0; k < 2 * rho; ++k) {                  // the four sides of the ring, one pixel of each per trip
                        probe(-rho + k, -rho, on);
                        probe(rho, -rho + k, on);
                        probe(rho - k, rho, on);
                        probe(-rho, rho - k, on);
                    }
                }
                done = settled;
                __builtin_amdgcn_wave_barrier();
            }
        }
        valid = own_valid && !done && !certd;
        if (__ballot(valid) != 0ull) {
            if (!spec_fetch) fetch_batch();
            park();                  // the staged tile records take the slab over
            scan_parked();
        }
        hinted = false;
        if constexpr (DBG) clk2 = clock64();
        if constexpr (!COOP) { opx = px; opy = py; opz = pz; }     // (cooperative build: re-read from qpos in step 4 -- three registers less across the drain)
        // ---- step 2: publish the queries and one work item per reachable coarse cell
        const float bnd0 = __int_as_float((int)(unsigned int)(bkey >> 32));
        tight = valid && bnd0 <= 0.0625f * g.gate2; loose = valid && !tight;
        {   // A tile that straddles a depth edge holds foreground AND background points: one box around them spans the whole
            // depth range and meets every target tile in between, none of which any lane's ball reaches (bench pair: the
            // heaviest 3 % of the tiles hit 52 target tiles with tight/loose boxes, 15 with near/far boxes, 12 are needed).
            // Such a wave classes its lanes by depth instead -- the two classes are only boxes and thresholds, any split is exact.
            const float zz = wave_min_x4(valid ? pz : inf, valid ? -pz : inf, valid ? -bnd0 : 0.0f, 0.0f);
            const float zmin = rdlane(zz, 0), zmax = -rdlane(zz, 32), bmax = -rdlane(zz, 16);
            if (zmax - zmin > 4.0f * sqrtf(bmax) + 0.05f) {
                const float zmid = 0.5f * (zmin + zmax);
                tight = valid && pz <= zmid; loose = valid && !tight;
            }
        }
        if constexpr (COOP) {
            qpos[w][0][lane] = px; qpos[w][1][lane] = py; qpos[w][2][lane] = pz;
            {
                const unsigned long long vm = __ballot(valid), tm = __ballot(tight);
                if (lane == 0) { qcls[w][0] = vm; qcls[w][1] = tm; }
            }
            qkey[w][lane] = bkey;
            if (lane < NN_STAGE) {
                int v = ta[0];
#pragma unroll
                for (int k = 1; k < NN_STAGE; ++k) if (lane == k) v = ta[k];
                wcentre[w][lane] = v;
            }
        }
        float thr_t, thr_l;
        if (__ballot(valid) != 0ull) {       // (a wave whose lanes were all settled by the window search has nothing to publish)
        class_boxes_thr(thr_t, thr_l);
        if (COOP && lane < 13) {   // the boxes do not change while the items are drained: helpers read them instead of redoing 12 wave reductions
            const float bx[13] = { qminx, qminy, qminz, qmaxx, qmaxy, qmaxz, lminx, lminy, lminz, lmaxx, lmaxy, lmaxz, any_loose ? 1.0f : 0.0f };
            float v = bx[0];
#pragma unroll
            for (int k = 1; k < 13; ++k) if (lane == k) v = bx[k];
            wbox[w][lane] = v;
        }
        for (int c0 = 0; c0 < tg.ncoarse; c0 += 64) {
            const int cidx = c0 + lane;
            bool hit = false;
            if (cidx < tg.ncoarse) hit = reach(CB[2 * cidx], CB[2 * cidx + 1], thr_t, thr_l);
            const unsigned long long cm = __ballot(hit);
            if constexpr (!COOP) {          // throughput build: every wave sweeps its own cells, nothing is shared
                unsigned long long rest = cm;
                while (rest) {
                    const int k = __builtin_ctzll(rest);
                    rest &= rest - 1;
                    sweep_cell(c0 + k);
                }
                continue;
            }
            const unsigned long long cml = cm;                                     // the reachable cells go to the block's shared cell list
            const int cnt = __popcll(cml);
            const int base = cnt ? lds_fetch_add_uniform(&n_items, cnt) : 0;       // cnt is wave-uniform
            if ((cml >> lane) & 1ull) {
                const int pos = base + __popcll(cml & ((1ull << lane) - 1ull));
                if (pos < NN_MAX_ITEMS) items[pos] = (w << 16) | cidx;
            }
            // cells that do not fit in the shared list are swept right here by their owner
            if (base + cnt > NN_MAX_ITEMS) {
                unsigned long long rest = cml;
                int skip = NN_MAX_ITEMS - base; if (skip < 0) skip = 0;
                while (rest) {
                    const int k = __builtin_ctzll(rest);
                    rest &= rest - 1;
                    if (skip > 0) { --skip; continue; }
                    sweep_cell(c0 + k);
                }
                qkey[w][lane] = bkey;
            }
        }
        }
        if constexpr (COOP) { qbs[w][0][lane] = (unsigned int)__float_as_int(bsc); qbs[w][1][lane] = (unsigned int)__float_as_int(sec); }
    }
    if constexpr (!COOP) { if (lane == 0) atomicAdd(&wcost[w], (int)(clock64() - cw0)); }
    if constexpr (DBG) clkP = clock64();
    if constexpr (COOP) __syncthreads();
    if constexpr (DBG) clkB1 = clock64();
    // ================= step 3: drain the shared work, two decoupled phases =================
    // Measured on the round-1 form (one wave took an item = (owner, cell) and ran its whole dependent chain -- child
    // boxes -> refine -> fetch a batch of tiles -> scan -> next batch ...): 7 k cycles per item, nearly all of it the
    // latency of serial global round trips, and the blocks with a dozen items ended the launch (wave lifetime mean 42 k
    // cycles, launch 72 k).  Now phase A turns cell items into tile items (two cells per trip: both child-box loads in
    // flight together), phase B consumes tile items of ANY owner three per trip (three records in flight), so a heavy
    // owner's tiles spread over all four waves and every wave waits for about one load round per phase.
    if constexpr (COOP) {
        auto load_owner = [&](int owner) __attribute__((always_inline)) {
            px = qpos[owner][0][lane]; py = qpos[owner][1][lane]; pz = qpos[owner][2][lane];
            valid = (qcls[owner][0] >> lane) & 1ull; tight = (qcls[owner][1] >> lane) & 1ull; loose = valid && !tight;
            bkey = qkey[owner][lane];
            bsc = inf; sec = inf;                                   // what THIS wave scans for the owner, merged afterwards
        };
        auto load_owner_boxes = [&](int owner) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < NN_STAGE; ++k) ta[k] = wcentre[owner][k];
            qminx = wbox[owner][0]; qminy = wbox[owner][1]; qminz = wbox[owner][2];
            qmaxx = wbox[owner][3]; qmaxy = wbox[owner][4]; qmaxz = wbox[owner][5];
            lminx = wbox[owner][6]; lminy = wbox[owner][7]; lminz = wbox[owner][8];
            lmaxx = wbox[owner][9]; lmaxy = wbox[owner][10]; lmaxz = wbox[owner][11];
            any_loose = wbox[owner][12] != 0.0f;
        };
        // ---- phase A: listed cell items -> tile items (blocks without listed cells skip it and its barrier)
        const int ncell = __builtin_amdgcn_readfirstlane(min(n_items, NN_MAX_ITEMS));     // same in every wave of the block
        const int achunk = ncell > NN_WAVES ? 2 : 1;
        while (ncell > 0) {
            const int it = lds_fetch_add_uniform(&next_item, achunk);
            if (it >= ncell) break;
            const int item0 = __builtin_amdgcn_readfirstlane(items[it]);          // one address for the wave: owner and cell are scalars
            const bool two = achunk == 2 && it + 1 < ncell;
            const int item1 = two ? __builtin_amdgcn_readfirstlane(items[it + 1]) : item0;
            float4 lo0, hi0, lo1 = make_float4(inf, inf, inf, 0), hi1 = make_float4(-inf, -inf, -inf, 0);
            int ctx0, cty0, ctx1 = 0, cty1 = 0;
            cell_boxes(item0 & 0xffff, lo0, hi0, ctx0, cty0);
            if (two) cell_boxes(item1 & 0xffff, lo1, hi1, ctx1, cty1);            // both loads in flight before the first use
            auto refine_item = [&](int item, const float4 lo, const float4 hi, int ctx, int cty) __attribute__((always_inline)) {
                const int owner = item >> 16;
                if constexpr (DBG) n_my_items += 1;
                load_owner(owner);
                load_owner_boxes(owner);
                if (append_tiles(owner, cell_refine(lo, hi, ctx, cty), ctx, cty) && valid) { atomicMin(&qkey[owner][lane], bkey); merge_bs(owner); }
            };
            refine_item(item0, lo0, hi0, ctx0, cty0);
            if (two) refine_item(item1, lo1, hi1, ctx1, cty1);
        }
        if constexpr (DBG) clkD = clock64();
        if (ncell > 0) __syncthreads();               // ncell is uniform over the block (read after barrier 1)
    }
    if constexpr (DBG) clkM = clock64();
    if constexpr (COOP) {
        // ---- phase B: tile items of any owner, NN_STAGE records in flight per trip
        const int ntile = __builtin_amdgcn_readfirstlane(min(n_titems, NN_MAX_TITEMS));
        // tiles per trip: enough for every wave to get a share, at most NN_STAGE records in flight
        const int bchunk = ntile <= NN_WAVES ? 1 : (ntile <= 2 * NN_WAVES ? 2 : NN_STAGE);
        while (true) {
            const int it = lds_fetch_add_uniform(&next_titem, bchunk);
            if (it >= ntile) break;
            int own[NN_STAGE];
#pragma unroll
            for (int k = 0; k < NN_STAGE; ++k) {
                const int e = (k < bchunk && it + k < ntile) ? __builtin_amdgcn_readfirstlane(titems[it + k]) : -1;
                tt[k] = e >= 0 ? (e & 0xffffff) : -1;
                own[k] = e >= 0 ? (e >> 24) : 0;
            }
            fetch_batch();
            park();
            hinted = true;              // the cell sweep already tested these tiles' boxes: go straight to the quadrant tests
#pragma unroll
            for (int k = 0; k < NN_STAGE; ++k) {
                if (tt[k] < 0) continue;
                px = qpos[own[k]][0][lane]; py = qpos[own[k]][1][lane]; pz = qpos[own[k]][2][lane];
                valid = (qcls[own[k]][0] >> lane) & 1ull;
                bkey = qkey[own[k]][lane];                     // the owner's best so far, every earlier merge included
                bsc = inf; sec = inf;
                if constexpr (DBG) {     // how many of the owner's lanes can reach this tile at all (the lanes the scan works for)
                    const int na = __popcll(__ballot(lane_gap_le(TB[2 * tt[k]], TB[2 * tt[k] + 1], lane_thr())));
                    n_b_hist += 1ll | ((long long)(na <= 2) << 8) | ((long long)(na <= 4) << 16) | ((long long)(na <= 8) << 24) | ((long long)(na <= 16) << 32) | ((long long)na << 40);
                }
                scan_staged(k, tt[k]);
                if (valid) { atomicMin(&qkey[own[k]][lane], bkey); merge_bs(own[k]); }
            }
            hinted = false;
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (DBG) clkE = clock64();
    if constexpr (COOP) __syncthreads();
    if constexpr (DBG) clk3 = clock64();
    // (end stamp: the last of the block's four waves to get here, found with an LDS counter -- one atomic per block)
    auto stamp_wave_end = [&]() __attribute__((always_inline)) {
        if (stamp && lane == 0 && atomicAdd(&waves_done, 1) == NN_WAVES - 1) stamp_end(stamp, c);
    };
    if (!has_tile) { stamp_wave_end(); return; }
    if constexpr (!COOP) { if (lane == 0) SS()->cost[(size_t)b * tg.ntiles + t] = wcost[w]; }          // input of k_balance
    // ================= step 4: this wave's own tile: fused S4 accumulation =================
    if constexpr (COOP) {
        bkey = qkey[w][lane];
        if (__ballot(own_valid) != 0ull) { opx = qpos[w][0][lane]; opy = qpos[w][1][lane]; opz = qpos[w][2][lane]; }     // published in step 2
    }
    if (__ballot(own_valid) == 0ull) bkey = ((unsigned long long)(unsigned int)__float_as_int(g.gate2) << 32) | 0xffffffffull;
    size_t gs_ep;                      // the slot index again, recomputed behind an opaque copy of t: kept live from the top of
    {                                  // the kernel it cost a register pair across the drain (one scratch spill per wave)
        int t_ep = t, b_ep = b;        // (b too: hipcc kept the common part b * nslots + lane of the two indices in a VGPR pair and spilled it)
        asm volatile("" : "+s"(t_ep), "+s"(b_ep));
        gs_ep = (size_t)b_ep * tg.nslots + (size_t)t_ep * TILE_SLOTS + lane;
    }
    RowBasis rb;
    int jnn = -1;
    const nn_static_ptr SE = SS();
    int *__restrict__ const corr = SE->corr;
    float *__restrict__ const cd2 = SE->cd2;
    float2 *__restrict__ const slot_rec = SE->slot_rec;
    if constexpr (GATED) {             // the optional S4g gates: their own instances, the production ones carry none of this
        SlotGates sg;
        const Rt m = load_rt_lds(head_T);
        // (the slot's source pixel is read again here: kept live from step 1 across the drain it cost the gated cooperative instance two
        // VGPR spills -- 12 B of scratch per lane; one 4-byte load per lane of an instance that gathers normals anyway)
        int t_g = t, lane_g = lane;
        asm volatile("" : "+s"(t_g), "+v"(lane_g));               // (opaque copies: the address must be formed HERE, not hoisted above the drain and spilled)
        const int spix = has_tile ? __float_as_int(pp.srcT[(size_t)t_g * TILE_SLOTS + lane_g].w) : -1;
        sg.resid2 = g.resid2; sg.min_ncos = g.min_ncos; sg.snrm = pp.snrm; sg.spix = max(spix, 0);
        sg.assoc = g.pair_gate ? pp.assoc : nullptr;
        sg.r[0] = m.r00; sg.r[1] = m.r01; sg.r[2] = m.r02; sg.r[3] = m.r10; sg.r[4] = m.r11; sg.r[5] = m.r12;
        sg.r[6] = m.r20; sg.r[7] = m.r21; sg.r[8] = m.r22;
        finish_slot<true>(own_valid, bkey, opx, opy, opz, tcloud, tnrm, g.gate2, g.estimator, g.b_scale, corr + gs_ep, cd2 + gs_ep, nullptr, rb,
                          write_out != 0, own_jprev, &sg, pp.tq, &jnn);
    } else
        finish_slot<false>(own_valid, bkey, opx, opy, opz, tcloud, tnrm, g.gate2, g.estimator, g.b_scale, corr + gs_ep, cd2 + gs_ep, nullptr, rb,
                           write_out != 0, own_jprev, nullptr, pp.tq, &jnn);
    {   // the slot's record for the next iteration: the match where it changed, the clearance where this lane searched
        const bool searched = own_valid && !((cert_mask >> lane) & 1ull);
        const bool wr_j = jnn != own_jprev;
        bool wr_c = cert && !trk && searched && (first || ((clear_mask >> lane) & 1ull));          // searched without tracking: no clearance
        float cnew = -1.0f;
        if (trk && searched) {
            // this lane searched: its clearance for the next iteration.  Scanned candidates other than the winner are at least
            // sqrt(other) away (other = the second-smallest scanned d2, or the smallest when the winner itself was not among the
            // scanned ones); everything the search did not scan is at least sqrt(best) + CERT_M away (every pruning radius was
            // inflated by CERT_M and only shrank afterwards); 1e-6 covers the roundings of d2 and of sqrt (see cert_tau).
            if constexpr (COOP) { bsc = __int_as_float((int)qbs[w][0][lane]); sec = __int_as_float((int)qbs[w][1][lane]); }
            const float u1 = __int_as_float((int)(unsigned int)(bkey >> 32));
            const bool real = (unsigned int)bkey != 0xffffffffu;                    // a match (else: nothing within the gate)
            const float other = real ? (u1 == bsc ? sec : bsc) : bsc;
            const float base = __builtin_amdgcn_sqrtf(real ? u1 : g.gate2);
            cnew = fminf(__builtin_amdgcn_sqrtf(other), base + cm) * (1.0f - 1.0e-6f) - base * (1.0f + 1.0e-6f) - 1.0e-9f;
            int cb = cum_bits;
            asm volatile("" : "+s"(cb));                                  // (keeps the doubling down here: hoisted, it costs a VGPR across the drain)
            cnew = cnew > 0.0f ? (cnew + 2.0f * __int_as_float(cb)) * (1.0f - 1.0e-6f) : -1.0f;      // recorded relative to the tile's motion total (rounded down)
            wr_c = true;
        }
        if (wr_j && wr_c) slot_rec[gs_ep] = make_float2(__int_as_float(jnn), cnew);
        else if (wr_j) slot_rec[gs_ep].x = __int_as_float(jnn);
        else if (wr_c) slot_rec[gs_ep].y = cnew;
    }
    // (the wave's stage slab is free by now: every helper left it before the block's last barrier)
    tile_accumulate(rb, SE->acc + (((size_t)b * SE->nsets + (head ? it : 0)) * ACC_R + (c % ACC_R)) * ACC_STRIDE, reinterpret_cast<double *>(st));
    stamp_wave_end();
    if (DBG && SE->dbg && b == 0 && lane == 0) {
        long long *d = SE->dbg + (size_t)t * 20;
        d[0] = clk0; d[1] = clk1; d[2] = clk2; d[3] = clk3; d[4] = clock64();
        d[5] = n_scanned | ((long long)n_chit << 32); d[6] = n_cand | ((long long)n_fhit << 32);
        d[7] = n_batches | ((long long)n_refined << 32);
        // where the wave ran: HW_ID (wave / simd / cu / sh / se fields) and the XCC id; when: the constant-rate
        // (100 MHz) real-time counter, which unlike s_memtime is common to all dies
        unsigned int hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        d[8] = (long long)hw_id | ((long long)xcc_id << 32);
        d[9] = ((long long)c * NN_WAVES + w) | ((long long)dbg_cert << 32);
        d[10] = rt0; d[11] = (long long)wall_clock64();
        d[12] = clkP; d[13] = clkB1; d[14] = clkD; d[15] = (long long)n_my_items | ((long long)(COOP ? n_items : 0) << 32);
        d[16] = clkM; d[17] = clkE; d[18] = (long long)(COOP ? n_titems : 0) | ((long long)__popcll(cert_mask) << 32); d[19] = n_b_hist;
    }
}

// Cost-balanced tile -> (block, wave) assignment of the THROUGHPUT build for the following iterations.  grid (B), block
// 1024.  The tiles are ranked by measured cost (cycles the previous launch spent on them; 256 buckets, counting sort,
// heaviest first) and dealt to the blocks in serpentine order, so every block receives one tile of each cost quartile
// and the block sums even out (+6 % at 64 pairs per launch).  Any assignment gives identical results (partials are
// order-free integer sums); this only shapes time.  (Round 1 also cut the frame into eight equal-cost bands, one per XCD
// and its L2, for launches of few pairs: -42 % fetch traffic on ONE pair repeated -- but on a stream of distinct pairs
// the map is always the previous pair's, the deal bought nothing and its launch cost 12 us per run: the cooperative
// build now keeps the interleaved default ownership, and the banded mode is gone.)
__global__ __launch_bounds__(1024) void k_balance(const int *__restrict__ cost, int *__restrict__ perm, TileGrid tg, int G)
{
    __shared__ int hist[256], start[256], cmax;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int *__restrict__ C = cost + (size_t)b * tg.ntiles;
    int *__restrict__ P = perm + (size_t)b * G * NN_WAVES;
    if (tid < 256) hist[tid] = 0;
    if (tid == 0) cmax = 1;
    __syncthreads();
    int m = 1;
    for (int t = tid; t < tg.ntiles; t += 1024) m = max(m, C[t]);
    atomicMax(&cmax, m);
    for (int s = tid; s < G * NN_WAVES; s += 1024) P[s] = -2;
    __syncthreads();
    const float bscale = 255.0f / (float)cmax;
    auto bucket_of = [&](int t) { return 255 - min(255, (int)((float)max(C[t], 0) * bscale)); };        // bucket 0 = heaviest
    for (int t = tid; t < tg.ntiles; t += 1024) atomicAdd(&hist[bucket_of(t)], 1);
    __syncthreads();
    if (tid < 64) {      // exclusive scan of the 256 buckets: four per lane
        int h[4], a = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { h[k] = hist[4 * lane + k]; a += h[k]; }
        int incl = a;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        int e = incl - a;
#pragma unroll
        for (int k = 0; k < 4; ++k) { start[4 * lane + k] = e; e += h[k]; }
    }
    __syncthreads();
    for (int t = tid; t < tg.ntiles; t += 1024) {
        const int r = atomicAdd(&start[bucket_of(t)], 1);             // rank, heaviest first
        const int q = r / G, i = r - q * G;
        const int k = (q & 1) ? G - 1 - i : i;
        P[k * NN_WAVES + q] = t;
    }
}

// ------------------------------------------------------------------------------------ S5
// deterministic sin/cos (spec): Cody-Waite reduction by pi/2 + Taylor/Horner, basic ops only
__device__ inline void spec_sincos(double x, double &s, double &c)
{
    const double kf = floor(x * 0.63661977236758134308 + 0.5);
    const double r = (x - kf * 1.57079632673412561417e+00) - kf * 6.07710050650619224932e-11;
    const double z = r * r;
    double ps = 1.0 / 355687428096000.0;
    ps = ps * z - 1.0 / 1307674368000.0;
    ps = ps * z + 1.0 / 6227020800.0;
    ps = ps * z - 1.0 / 39916800.0;
    ps = ps * z + 1.0 / 362880.0;
    ps = ps * z - 1.0 / 5040.0;
    ps = ps * z + 1.0 / 120.0;
    ps = ps * z - 1.0 / 6.0;
    const double sr = r + r * (z * ps);
    double pc = -1.0 / 6402373705728000.0;
    pc = pc * z + 1.0 / 20922789888000.0;
    pc = pc * z - 1.0 / 87178291200.0;
    pc = pc * z + 1.0 / 479001600.0;
    pc = pc * z - 1.0 / 3628800.0;
    pc = pc * z + 1.0 / 40320.0;
    pc = pc * z - 1.0 / 720.0;
    pc = pc * z + 1.0 / 24.0;
    pc = pc * z - 0.5;
    const double cr = 1.0 + z * pc;
    const long long k = (long long)kf;
    const int quad = (int)(((k % 4) + 4) % 4);
    if (quad == 0) { s = sr; c = cr; }
    else if (quad == 1) { s = cr; c = -sr; }
    else if (quad == 2) { s = -sr; c = -cr; }
    else { s = -cr; c = sr; }
}

__device__ inline bool ldl6(const double (*A)[6], const double *b, double tr, double *x)
{
    double L[6][6], D[6], y[6];
    const double floor_piv = 1e-12 * tr / 6.0;
    for (int j = 0; j < 6; ++j) {
        double d = A[j][j];
        for (int k = 0; k < j; ++k) d -= (L[j][k] * L[j][k]) * D[k];
        if (!(d > floor_piv)) return false;
        D[j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double v = A[i][j];
            for (int k = 0; k < j; ++k) v -= (L[i][k] * L[j][k]) * D[k];
            L[i][j] = v / d;
        }
    }
    for (int i = 0; i < 6; ++i) {
        double v = b[i];
        for (int k = 0; k < i; ++k) v -= L[i][k] * y[k];
        y[i] = v;
    }
    for (int i = 0; i < 6; ++i) y[i] = y[i] / D[i];
    for (int i = 5; i >= 0; --i) {
        double v = y[i];
        for (int k = i + 1; k < 6; ++k) v -= L[k][i] * x[k];
        x[i] = v;
    }
    return true;
}

// 1 solved, 2 solved after damping, 0 failed
__device__ inline int solve6(const double *U, const double *Atb, double *x)
{
    double A[6][6];
    int k = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) { A[r][c] = U[k]; A[c][r] = U[k]; ++k; }
    double tr = 0.0;
    for (int r = 0; r < 6; ++r) tr += A[r][r];
    if (!(tr > 0.0)) return 0;
    if (ldl6(A, Atb, tr, x)) return 1;
    const double lam = 1e-9 * tr / 6.0;
    for (int r = 0; r < 6; ++r) A[r][r] = A[r][r] + lam;
    if (ldl6(A, Atb, tr, x)) return 2;
    return 0;
}

// one-sided Jacobi SVD of H (at most 12 sweeps; round 6: rotations of columns orthogonal to 2^-50 are skipped, a sweep without a
// rotation ends it -- oracle/icp_oracle.c::orc_svd3_rotation) -> R = V U^T with det +1 (spec S5, Kabsch)
__device__ inline void svd3_rotation(const double *H, double *R)
{
    double g[3][3], v[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { g[r][c] = H[r * 3 + c]; v[r][c] = r == c ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 12; ++sweep) {
        bool rotated = false;
        for (int k = 0; k < 3; ++k) {
            const int p = k == 2 ? 1 : 0, q = k == 0 ? 1 : 2;
            const double al = (g[0][p] * g[0][p] + g[1][p] * g[1][p]) + g[2][p] * g[2][p];
            const double be = (g[0][q] * g[0][q] + g[1][q] * g[1][q]) + g[2][q] * g[2][q];
            const double ga = (g[0][p] * g[0][q] + g[1][p] * g[1][q]) + g[2][p] * g[2][q];
            if (ga * ga <= 0x1p-100 * (al * be)) continue;
            rotated = true;
            const double zeta = (be - al) / (2.0 * ga);
            double t = 1.0 / (fabs(zeta) + sqrt(zeta * zeta + 1.0));
            if (zeta < 0.0) t = -t;
            const double c = 1.0 / sqrt(t * t + 1.0);
            const double s = c * t;
            for (int m = 0; m < 3; ++m) {
                const double gp = g[m][p], gq = g[m][q];
                g[m][p] = c * gp - s * gq;
                g[m][q] = s * gp + c * gq;
                const double vp = v[m][p], vq = v[m][q];
                v[m][p] = c * vp - s * vq;
                v[m][q] = s * vp + c * vq;
            }
        }
        if (!rotated) break;
    }
    double sg[3];
    for (int k = 0; k < 3; ++k) sg[k] = sqrt((g[0][k] * g[0][k] + g[1][k] * g[1][k]) + g[2][k] * g[2][k]);
    int i0 = 0, i1 = 1, i2 = 2, tmp;
    if (sg[i1] > sg[i0]) { tmp = i0; i0 = i1; i1 = tmp; }
    if (sg[i2] > sg[i1]) { tmp = i1; i1 = i2; i2 = tmp; }
    if (sg[i1] > sg[i0]) { tmp = i0; i0 = i1; i1 = tmp; }
    for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if (!(sg[i0] > 0.0) || !(sg[i1] > 1e-14 * sg[i0])) return;
    double u0[3], u1[3], u2[3], v0[3], v1[3], v2[3];
    for (int m = 0; m < 3; ++m) {
        u0[m] = g[m][i0] / sg[i0]; u1[m] = g[m][i1] / sg[i1];
        v0[m] = v[m][i0];          v1[m] = v[m][i1];
    }
    u2[0] = u0[1] * u1[2] - u0[2] * u1[1]; u2[1] = u0[2] * u1[0] - u0[0] * u1[2]; u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
    v2[0] = v0[1] * v1[2] - v0[2] * v1[1]; v2[1] = v0[2] * v1[0] - v0[0] * v1[2]; v2[2] = v0[0] * v1[1] - v0[1] * v1[0];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[r * 3 + c] = (v0[r] * u0[c] + v1[r] * u1[c]) + v2[r] * u2[c];
}

__device__ inline void compose(const double *dR, const double *dt, double *T)
{
    double Tn[16];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
            Tn[r * 4 + c] = (dR[r * 3 + 0] * T[0 * 4 + c] + dR[r * 3 + 1] * T[1 * 4 + c]) + dR[r * 3 + 2] * T[2 * 4 + c];
        Tn[r * 4 + 3] = ((dR[r * 3 + 0] * T[3] + dR[r * 3 + 1] * T[7]) + dR[r * 3 + 2] * T[11]) + dt[r];
    }
    Tn[12] = 0.0; Tn[13] = 0.0; Tn[14] = 0.0; Tn[15] = 1.0;
    for (int k = 0; k < 16; ++k) T[k] = Tn[k];
}

// one solve + SE(3) update step from the 29 sums on a local pose T (one thread): 1 solved, 2 solved after damping, 0 no update (T unchanged)
__device__ inline int solve_step_one(const double *__restrict__ sums, double *T, int estimator)
{
    int rc = 0;
    double dR[9], dt[3];
    if (estimator == 0) {
        if (!(sums[27] < 6.0)) {
            double x[6];
            rc = solve6(sums, sums + 21, x);
            if (rc) {
                double sa, ca, sb, cb, sg, cg;
                spec_sincos(x[0], sa, ca); spec_sincos(x[1], sb, cb); spec_sincos(x[2], sg, cg);
                dR[0] = cg * cb; dR[1] = (cg * sb) * sa - sg * ca; dR[2] = (cg * sb) * ca + sg * sa;
                dR[3] = sg * cb; dR[4] = (sg * sb) * sa + cg * ca; dR[5] = (sg * sb) * ca - cg * sa;
                dR[6] = -sb;     dR[7] = cb * sa;                  dR[8] = cb * ca;
                dt[0] = x[3]; dt[1] = x[4]; dt[2] = x[5];
            }
        }
    } else {
        const double n = sums[27];
        if (!(n < 3.0)) {
            const double pm[3] = { sums[0] / n, sums[1] / n, sums[2] / n };
            const double qm[3] = { sums[3] / n, sums[4] / n, sums[5] / n };
            double H[9];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) H[r * 3 + c] = sums[6 + r * 3 + c] - (n * pm[r]) * qm[c];
            svd3_rotation(H, dR);
            for (int r = 0; r < 3; ++r)
                dt[r] = qm[r] - ((dR[r * 3 + 0] * pm[0] + dR[r * 3 + 1] * pm[1]) + dR[r * 3 + 2] * pm[2]);
            rc = 1;
        }
    }
    if (rc) compose(dR, dt, T);
    return rc;
}

// solve + SE(3) update from the 29 sums (one thread), trace bookkeeping
__device__ inline void solve_update_one(const double *__restrict__ sums, double *__restrict__ Tcur_b,
                                        double *__restrict__ trace_T_b, double *__restrict__ trace_S_b,
                                        int *__restrict__ flag, int it, int estimator)
{
    double T[16];
    for (int k = 0; k < 16; ++k) T[k] = Tcur_b[k];
    for (int k = 0; k < NSUMS; ++k) trace_S_b[(size_t)it * NSUMS + k] = sums[k];
    const int rc = solve_step_one(sums, T, estimator);
    if (rc) {
        for (int k = 0; k < 16; ++k) Tcur_b[k] = T[k];
        if (rc == 2) *flag = *flag | 1;
    } else {
        *flag = *flag | 2;       // no update in this iteration (too few rows, or the system could not be solved): never a silent "ok"
    }
    for (int k = 0; k < 16; ++k) trace_T_b[(size_t)(it + 1) * 16 + k] = T[k];
}

// ---- wave-parallel form of solve_update_one for the point-to-plane estimator (k_solve_acc's single wave).  Same
// operations in the same order per value as the serial code above (which the svd estimator and the dense mode's k_solve
// still run, and which the oracle restates), so every bit of T is the same -- but no 6x6 arrays in one lane's registers
// (the serial form needed 124 VGPRs and 192 B of scratch):
//   LDL^T       lane i < 6 owns row i of A and of L; at step j every lane forms v_i = A_ij - sum_k (L_ik L_jk) D_k with
//               L_jk broadcast from lane j -- for i == j that is the pivot d_j, for i > j it is L_ij d_j;
//   forward     y_i -= L_ik y_k with y_k broadcast, k ascending (the serial order per row);
//   backward    x_5 .. x_0 on broadcast values, terms k ascending (the serial order);
//   sin / cos   lanes 0, 1, 2; compose: lane l < 12 forms entry l of the new T.
// (Tried and dropped: running this in the tail of the NN launch, by the block that takes the pair's last ticket.  The
// chain ticket -> collect the accumulators -> solve is as long as the launch it replaces (un-overlapped latency 0.91 ->
// 0.94 ms) and the call cost the throughput build 40 SGPR spills: 58 k -> 54 k it/s at 64 pairs per launch.)
__device__ __forceinline__ double bcast_d(double v, int src_lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// tot: the 29 sums (LDS), sh: T of the pair (LDS, 16 doubles); all 64 lanes of the wave call this.  Returns in lane
// l < 16 entry l of the updated pose (the old one when the solve failed) and rc: 1 solved, 2 solved after damping, 0 no
// update.  Values only -- who stores what is the caller's business (k_solve_acc; the head of the NN launch).
__device__ __forceinline__ double wave_solve_point2plane(const double *tot, const double *sh, int &rc_out)
{
    const int lane = threadIdx.x & 63;
    const int i = lane < 6 ? lane : 5;                      // lanes >= 6 shadow row 5 (their values are never used)
    int rc = 0;
    double x0 = 0, x1 = 0, x2 = 0, x3 = 0, x4 = 0, x5 = 0;
    if (!(tot[27] < 6.0)) {
        // row i of the symmetric matrix from the 21 upper-triangle sums: entry (r, c), r <= c, sits at first[r] + (c - r)
        double Ar[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int r0 = i < c ? i : c, c0 = i < c ? c : i;
            const int first = r0 == 0 ? 0 : (r0 == 1 ? 6 : (r0 == 2 ? 11 : (r0 == 3 ? 15 : (r0 == 4 ? 18 : 20))));
            Ar[c] = tot[first + (c0 - r0)];
        }
        const double bi = tot[21 + i];
        double tr = 0.0;
        tr += tot[0]; tr += tot[6]; tr += tot[11]; tr += tot[15]; tr += tot[18]; tr += tot[20];
        if (tr > 0.0) {
            const double floor_piv = 1e-12 * tr / 6.0;
#pragma unroll 1
            for (int attempt = 0; attempt < 2 && rc == 0; ++attempt) {
                if (attempt == 1) {                          // A[r][r] = A[r][r] + lam on every diagonal entry
                    const double lam = 1e-9 * tr / 6.0;
#pragma unroll
                    for (int c = 0; c < 6; ++c) if (c == i) Ar[c] = Ar[c] + lam;
                }
                double L[6] = { 0, 0, 0, 0, 0, 0 }, D[6] = { 0, 0, 0, 0, 0, 0 };
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    if (ok) {
                        double v = Ar[j];
#pragma unroll
                        for (int k = 0; k < j; ++k) v -= (L[k] * bcast_d(L[k], j)) * D[k];
                        const double d = bcast_d(v, j);
                        if (!(d > floor_piv)) ok = false;
                        else { D[j] = d; L[j] = v / d; }
                    }
                }
                if (!ok) continue;
                double yv = bi;                              // forward: y_i = b_i - sum_{k<i} L_ik y_k
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const double yk = bcast_d(yv, k);
                    if (i > k) yv -= L[k] * yk;
                }
                {                                            // y_i = y_i / D_i
                    double Di = D[0];
#pragma unroll
                    for (int k = 1; k < 6; ++k) if (i == k) Di = D[k];
                    yv = yv / Di;
                }
                // backward on broadcast values: x_i = y_i - sum_{k>i} L_ki x_k, k ascending
                x5 = bcast_d(yv, 5);
                x4 = bcast_d(yv, 4); x4 -= bcast_d(L[4], 5) * x5;
                x3 = bcast_d(yv, 3); x3 -= bcast_d(L[3], 4) * x4; x3 -= bcast_d(L[3], 5) * x5;
                x2 = bcast_d(yv, 2); x2 -= bcast_d(L[2], 3) * x3; x2 -= bcast_d(L[2], 4) * x4; x2 -= bcast_d(L[2], 5) * x5;
                x1 = bcast_d(yv, 1); x1 -= bcast_d(L[1], 2) * x2; x1 -= bcast_d(L[1], 3) * x3; x1 -= bcast_d(L[1], 4) * x4; x1 -= bcast_d(L[1], 5) * x5;
                x0 = bcast_d(yv, 0); x0 -= bcast_d(L[0], 1) * x1; x0 -= bcast_d(L[0], 2) * x2; x0 -= bcast_d(L[0], 3) * x3; x0 -= bcast_d(L[0], 4) * x4;
                x0 -= bcast_d(L[0], 5) * x5;
                rc = attempt == 0 ? 1 : 2;
            }
        }
    }
    double Tn = lane < 16 ? sh[lane] : 0.0;                 // unchanged when the solve failed
    if (rc) {
        double s_, c_;
        spec_sincos(lane == 0 ? x0 : (lane == 1 ? x1 : x2), s_, c_);
        const double sa = bcast_d(s_, 0), ca = bcast_d(c_, 0), sb = bcast_d(s_, 1), cb = bcast_d(c_, 1), sg = bcast_d(s_, 2), cg = bcast_d(c_, 2);
        const int r = (lane >> 2) & 3, c = lane & 3;
        // row r of dR = Rz(g) Ry(b) Rx(a), the expressions of solve_update_one
        const double d0 = r == 0 ? cg * cb : (r == 1 ? sg * cb : -sb);
        const double d1 = r == 0 ? (cg * sb) * sa - sg * ca : (r == 1 ? (sg * sb) * sa + cg * ca : cb * sa);
        const double d2 = r == 0 ? (cg * sb) * ca + sg * sa : (r == 1 ? (sg * sb) * ca - cg * sa : cb * ca);
        const double dtr = r == 0 ? x3 : (r == 1 ? x4 : x5);
        if (lane < 12) {
            Tn = (d0 * sh[c] + d1 * sh[4 + c]) + d2 * sh[8 + c];
            if (c == 3) Tn = Tn + dtr;
        } else if (lane < 16) {
            Tn = lane == 15 ? 1.0 : 0.0;
        }
    }
    rc_out = rc;
    return Tn;
}

__device__ __forceinline__ void wave_solve_update_point2plane(const double *tot, const double *sh, double *__restrict__ Tcur_b,
                                                              double *__restrict__ trace_T_b, double *__restrict__ trace_S_b,
                                                              int *__restrict__ flag_b, int it, double *__restrict__ res_rec /* nullable */,
                                                              const PairPtrs *pp /* read only with res_rec */)
{
    const int lane = threadIdx.x & 63;
    if (lane < NSUMS) trace_S_b[(size_t)it * NSUMS + lane] = tot[lane];
    int rc;
    const double Tn = wave_solve_point2plane(tot, sh, rc);
    if (lane < 16) Tcur_b[lane] = Tn;                    // (a failed solve returns the old pose)
    if (lane < 16) trace_T_b[(size_t)(it + 1) * 16 + lane] = Tn;
    int flag = 0;
    if (lane == 0) {
        flag = *flag_b;
        if (rc == 2) flag |= 1;
        if (rc == 0) flag |= 2;        // no update in this iteration: never a silent "ok"
        *flag_b = flag;
    }
    if (res_rec) {
        if (lane < 16) res_rec[lane] = Tn;
        if (lane < NSUMS) res_rec[16 + lane] = tot[lane];
        if (lane == 0) { res_rec[45] = (double)flag; res_rec[46] = (double)pp->src_counts[0]; res_rec[47] = (double)pp->tgt_counts[1]; }
    }
}

// The iteration's tail: grid (B), block 64.  Lane k adds the replicas of component k (integers: any order),
// converts to double, clears the accumulators for the next launch; lane 0 solves and updates T (do_solve) or the
// raw integer sums go to `raw_out` for the caller's all-reduce (dense mode).  In the final iteration the pose
// record goes straight to host-mapped memory, so fetch_results is a stream synchronisation with no copies.
template <int EST>
__global__ __launch_bounds__(64) void k_solve_acc(long long *__restrict__ acc, long long *__restrict__ raw_out,
                                                  double *__restrict__ Tcur, double *__restrict__ trace_T,
                                                  double *__restrict__ trace_S, int *__restrict__ flags,
                                                  const PairPtrs *__restrict__ pairs, double *__restrict__ res_host,
                                                  int it, int iters, int do_solve, StampRing sring, int stamp_idx,
                                                  int nsets, int set /* which accumulator set of the pair: 0, or `it` after head-solved launches */,
                                                  int from_trace /* T_it from trace_T[it] (head-solved launches leave Tcur at T_0) */,
                                                  int *__restrict__ end_run /* the device's run counter when this is the last launch of a slam3d_icp_run (one run less in flight), else null */,
                                                  int eb /* spec S4: exponent of the residual component */)
{
    if (end_run && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(end_run, -1);
    __shared__ double tot[32], Tsh[16];
    __shared__ long long Gs[NRAW];
    const int b = blockIdx.x, k = threadIdx.x;
    unsigned long long *const stamp = stamp_row(sring, stamp_idx);
    if (k == 0) stamp_start(stamp, b);
    long long *__restrict__ A = acc + ((size_t)b * nsets + set) * ACC_R * ACC_STRIDE;
    if (k < NRAW) {
        long long q = 0;
#pragma unroll
        for (int r = 0; r < ACC_R; ++r) q += __hip_atomic_load(A + r * ACC_STRIDE + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int r = 0; r < ACC_R; ++r) A[r * ACC_STRIDE + k] = 0;
        if (raw_out) raw_out[b * NRAW + k] = q;
        Gs[k] = q;
    }
    if (k < 16) Tsh[k] = from_trace ? trace_T[((size_t)b * (iters + 1) + it) * 16 + k] : Tcur[b * 16 + k];
    __syncthreads();
    if (k < NSUMS) tot[k] = derive_sum(EST, eb, k, Gs);
    __syncthreads();
    if constexpr (EST == 0) {                 // point-to-plane: the whole wave solves (lane-parallel LDL^T, no scratch)
        if (do_solve)
            wave_solve_update_point2plane(tot, Tsh, Tcur + b * 16, trace_T + (size_t)b * (iters + 1) * 16, trace_S + (size_t)b * iters * NSUMS,
                                          flags + b, it, (res_host && it == iters - 1) ? res_host + (size_t)b * RES_REC : nullptr, pairs + b);
        if (k == 0) stamp_end(stamp, b);
        return;
    }
    if (k == 0 && do_solve) {
        solve_update_one(tot, Tcur + b * 16, trace_T + (size_t)b * (iters + 1) * 16, trace_S + (size_t)b * iters * NSUMS,
                         flags + b, it, EST);
        if (res_host && it == iters - 1) {
            double *__restrict__ r = res_host + (size_t)b * RES_REC;
            for (int j = 0; j < 16; ++j) r[j] = Tcur[b * 16 + j];
            for (int j = 0; j < NSUMS; ++j) r[16 + j] = tot[j];
            r[45] = (double)flags[b]; r[46] = (double)pairs[b].src_counts[0]; r[47] = (double)pairs[b].tgt_counts[1];
        }
    }
    if (k == 0) stamp_end(stamp, b);
}

// dense mode, three-step exchange: solve from externally reduced (all-reduced) integer Gram totals.  grid (B), block 64: the
// point-to-plane estimator solves with the whole wave like k_solve_acc (the one-thread form took 128 VGPRs, 72 spills and 272 B
// of scratch), the svd estimator in lane 0.
template <int EST>
__global__ __launch_bounds__(64) void k_solve(const long long *__restrict__ sums_all, double *__restrict__ Tcur,
                                              double *__restrict__ trace_T, double *__restrict__ trace_S,
                                              int *__restrict__ flags, int it, int iters, int eb)
{
    __shared__ double tot[32], Tsh[16];
    const int b = blockIdx.x, k = threadIdx.x;
    if (k < NSUMS) tot[k] = derive_sum(EST, eb, k, sums_all + (size_t)b * NRAW);
    if (k < 16) Tsh[k] = Tcur[b * 16 + k];
    __syncthreads();
    if constexpr (EST == 0) {
        wave_solve_update_point2plane(tot, Tsh, Tcur + b * 16, trace_T + (size_t)b * (iters + 1) * 16, trace_S + (size_t)b * iters * NSUMS,
                                      flags + b, it, nullptr, nullptr);
    } else {
        if (k == 0)
            solve_update_one(tot, Tcur + b * 16, trace_T + (size_t)b * (iters + 1) * 16, trace_S + (size_t)b * iters * NSUMS, flags + b, it, EST);
    }
}

// slot-order correspondences -> original pixel order (for get_correspondences)
__global__ __launch_bounds__(256) void k_fill_corr(int *__restrict__ idx, float *__restrict__ d2, int N)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) { idx[i] = -1; d2[i] = __int_as_float(0x7f800000); }
}

__global__ __launch_bounds__(256) void k_scatter_corr(const PairPtrs *__restrict__ pairs, const int *__restrict__ corr /* [nslots] of the pair */,
                                                      const float *__restrict__ cd2 /* nullable */, int b, TileGrid tg,
                                                      int *__restrict__ idx, float *__restrict__ d2)
{
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= tg.ntiles * TILE_SLOTS) return;
    const int pix = __float_as_int(pairs[b].srcT[slot].w);
    if (pix < 0) return;
    idx[pix] = corr[slot];
    if (cd2) d2[pix] = cd2[slot];
}

} // namespace s3d
