// plane_seg.hpp -- batched RANSAC plane segmentation on gfx950.
//
// Replaces the pcl::SACSegmentation loop of GraphicEnd::extractPlanesAndGenerateImage
// (src/GraphicEnd.cpp:353-430): up to max_planes rounds of {H seeded hypotheses, consensus count over the
// unassigned points, least-squares refinement of the consensus set, re-selection, sign rule d >= 0 (:383-387),
// removal of the inliers (:419-420)} while more than plane_percent of the cloud is unassigned (:372).
// Spec P1-P5; oracle/seg_oracle.c is the CPU twin and gives the same bits:
// the consensus counts are integers, the moments are integer fixed point (order-free, so wave reductions and
// atomics are exact) and the 3x3 eigen solve is the spec's cyclic Jacobi run by one thread.
//
// One launch sequence serves B frames (blockIdx.y / blockIdx.x = frame).  Nothing returns to the host between
// the rounds: the loop condition lives in SegState.done on the device.
#pragma once

#include "icp_kernels.hpp"

namespace s3d {

constexpr int SEG_H = 64;          // hypotheses per round (lane h of a wave owns hypothesis h)
constexpr int SEG_DRAWS = 32;      // PRNG draws a hypothesis may spend on its three points
constexpr int SEG_MAXP = 8;        // planes per frame
constexpr int SEG_PTS = 4;         // points per thread in the consensus / moment kernels of ONE frame (and of k_fit_moments)
constexpr int SEG_PTS_BATCH = 8;   // ... of a batch of frames: half the blocks and half the atomics per frame (64 frames: 32.4 -> 37.1 k frames/s; a frame alone: 137 -> 149 us)
constexpr int SEG_BLOCK = 256;
constexpr int SEG_HGROUP = 16;     // hypotheses per k_seg_count block (grid.z = SEG_H / SEG_HGROUP)
constexpr int SEG_CR = 8;          // replicas of the consensus counts (block x adds into replica x % SEG_CR: atomics on one cache line serialise)

struct SegParams { float thr, percent; int max_planes, hypotheses; unsigned long long seed; };

struct SegHyp { float nx, ny, nz, dd, thr2nn; int ok; float p0x, p0y, p0z; };

struct SegPlane { float a, b, c, d, cx, cy, cz; int count; };

// Round 4: a frame ALONE takes THREE launches per round instead of five (k_seg_hyp goes into the head of k_seg_count,
// k_seg_refine into the head of k_seg_label; 158 -> 142 us): every block redoes the few hundred scalar operations of the bookkeeping / hypotheses / eigen solve from
// inputs that no block of the same launch writes, so all blocks agree and block 0 alone records the result.  What one
// launch accumulates while another generation is still being read lives in two copies, indexed by the round's parity:
//   rs[r & 1]          bookkeeping as round r finds it (remaining, nplanes, done); round r's count launch writes rs[(r+1) & 1]
//   lab_count[r & 1]   points labelled in round r          (zeroed by round r's count launch, added by its label launch)
//   mom[r & 1]         moments of round r                  (zeroed by round r's count launch, added by its moments launch)
//   counts[r & 1]      consensus counts of round r         (zeroed by round r-1's moments launch; rounds 0 / 1: the initial memset)
// Frames in batches keep the five launches (the redundant heads cost a batch of 64 frames 29 % of its rate), on the same state.
struct SegRound { int remaining, nplanes, done, pad; };
struct SegState {                  // one per frame, zeroed before k_seg_init
    int n_valid, best, nplanes, pad0;          // nplanes: the final count (k_seg_final)
    SegRound rs[2];
    int lab_count[2], pad1[2];
    int counts[2][SEG_CR][SEG_H];
    long long mom[2][10];
    SegHyp hyp[SEG_H];
    SegPlane planes[SEG_MAXP];
};

__device__ __forceinline__ unsigned long long seg_mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ bool seg_inlier(float nx, float ny, float nz, float dd, float thr2nn, float4 q)
{
    const float e = __fmaf_rn(nx, q.x, __fmaf_rn(ny, q.y, nz * q.z)) + dd;
    return e * e <= thr2nn;
}

// sum over the 256 threads of a block (valid in thread 0): wave shuffles, then LDS across the four waves
__device__ __forceinline__ int block_sum_int(int v)
{
    __shared__ int part[SEG_BLOCK / 64];
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    return part[0] + part[1] + part[2] + part[3];
}

// P1: labels = -1 (valid, unassigned) / -2 (invalid); n_valid.  grid (ceil(N/1024), B)
template <int PTS>
__global__ __launch_bounds__(SEG_BLOCK) void k_seg_init(const float4 *const *__restrict__ clouds, int *__restrict__ labels,
                                                        SegState *__restrict__ st, int N, float zmax)
{
    const int b = blockIdx.y;
    const float4 *__restrict__ cloud = clouds[b];
    int *__restrict__ lab = labels + (size_t)b * N;
    int nv = 0;
#pragma unroll
    for (int k = 0; k < PTS; ++k) {
        const int i = (blockIdx.x * PTS + k) * SEG_BLOCK + threadIdx.x;
        if (i < N) {
            const float4 q = cloud[i];
            const bool ok = pt_valid(q.x, q.y, q.z, zmax);
            lab[i] = ok ? -1 : -2;
            nv += ok ? 1 : 0;
        }
    }
    nv = block_sum_int(nv);
    if (threadIdx.x == 0 && nv) atomicAdd(&st[b].n_valid, nv);     // one same-address atomic per block, not per wave
}

// bookkeeping of the round that just ended (P4 tail) + the loop condition of round r (P2 head): values only -- every
// block of a launch evaluates this from the same inputs.  `got`: points the previous round labelled.
__device__ __forceinline__ SegRound seg_open_round(const SegState &s, int r, float percent, int &plane_closed, int &plane_count)
{
    SegRound c = s.rs[r & 1];
    plane_closed = -1; plane_count = 0;
    if (r == 0) c.remaining = s.n_valid;
    else if (!c.done) {
        const int got = s.lab_count[(r - 1) & 1];
        if (got == 0) c.done = 1;
        else { plane_closed = r - 1; plane_count = got; c.nplanes = r; c.remaining -= got; }
    }
    return c;
}

// hypothesis h of round r (P2): three distinct unassigned points drawn with a counter-based generator
__device__ __forceinline__ SegHyp seg_make_hyp(const float4 *__restrict__ cloud, const int *__restrict__ lab, int N, const SegParams &sp, int r, int h)
{
    SegHyp hy;
    hy.nx = hy.ny = hy.nz = hy.dd = hy.thr2nn = 0.0f; hy.ok = 0; hy.p0x = hy.p0y = hy.p0z = 0.0f;
    if (h < sp.hypotheses) {
        unsigned long long x = sp.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(1 + r * 4096 + h);
        int pick0 = -1, pick1 = -1, pick2 = -1, np = 0;
        // The draws are a counter-based sequence: the first SEG_AHEAD of them are formed at once and their labels fetched TOGETHER (one
        // round trip instead of up to one per draw -- a count launch's head was three to five dependent trips), then walked in order with
        // the spec's acceptance rule; the rare hypothesis that needs more draws goes on one by one.  Same picks by construction.
        constexpr int SEG_AHEAD = 8;
        int px_[SEG_AHEAD], lb_[SEG_AHEAD];
#pragma unroll
        for (int t = 0; t < SEG_AHEAD; ++t) {
            x += 0x9E3779B97F4A7C15ull;
            const unsigned long long o = seg_mix64(x);
            px_[t] = (int)(((o >> 32) * (unsigned long long)N) >> 32);
        }
#pragma unroll
        for (int t = 0; t < SEG_AHEAD; ++t) lb_[t] = lab[px_[t]];
#pragma unroll
        for (int t = 0; t < SEG_AHEAD; ++t) {
            const int pix = px_[t];
            const bool take = np < 3 && lb_[t] == -1 && !((np > 0 && pick0 == pix) || (np > 1 && pick1 == pix));
            if (take) { if (np == 0) pick0 = pix; else if (np == 1) pick1 = pix; else pick2 = pix; ++np; }
        }
        for (int t = SEG_AHEAD; t < SEG_DRAWS && np < 3; ++t) {
            x += 0x9E3779B97F4A7C15ull;
            const unsigned long long o = seg_mix64(x);
            const int pix = (int)(((o >> 32) * (unsigned long long)N) >> 32);
            if (lab[pix] != -1) continue;
            if ((np > 0 && pick0 == pix) || (np > 1 && pick1 == pix)) continue;
            if (np == 0) pick0 = pix; else if (np == 1) pick1 = pix; else pick2 = pix;
            ++np;
        }
        if (np == 3) {
            const float4 p0 = cloud[pick0], p1 = cloud[pick1], p2 = cloud[pick2];
            const float ax = p1.x - p0.x, ay = p1.y - p0.y, az = p1.z - p0.z;
            const float bx = p2.x - p0.x, by = p2.y - p0.y, bz = p2.z - p0.z;
            const float nx = __fmaf_rn(ay, bz, -(az * by));
            const float ny = __fmaf_rn(az, bx, -(ax * bz));
            const float nz = __fmaf_rn(ax, by, -(ay * bx));
            const float nn = __fmaf_rn(nz, nz, __fmaf_rn(ny, ny, nx * nx));
            if (nn > 1e-16f) {
                hy.nx = nx; hy.ny = ny; hy.nz = nz;
                hy.dd = -__fmaf_rn(nx, p0.x, __fmaf_rn(ny, p0.y, nz * p0.z));
                hy.thr2nn = (sp.thr * sp.thr) * nn;
                hy.ok = 1;
                hy.p0x = p0.x; hy.p0y = p0.y; hy.p0z = p0.z;
            }
        }
    }
    return hy;
}

// consensus count of hypothesis h: the replicas' sum
__device__ __forceinline__ int seg_count_of(const SegState &s, int r, int h)
{
    int c = 0;
#pragma unroll
    for (int k = 0; k < SEG_CR; ++k) c += s.counts[r & 1][k][h];
    return c;
}

// P2: bookkeeping + loop condition + the block's SEG_HGROUP hypotheses, then their consensus counts.
// grid (ceil(N/1024), B, H / SEG_HGROUP), block 256.  Each thread keeps SEG_PTS points in registers; lane h of wave 0
// builds hypothesis h0 + h and the block shares them through LDS; the wave-uniform loop broadcasts hypothesis h with
// v_readlane; a ballot + popcount gives the wave's count, which lane h keeps; the four waves meet in LDS and the block ends
// with one 64-lane atomic (same-address global atomics serialise, so there is one per block, not per wave).
// Block (0, b, z) records: the bookkeeping (z = 0), its hypotheses, and the zeroes of the round's other accumulators.
// the round's head on its own (frames in BATCHES: one block per frame does it once instead of every count block): grid (B), block 64
__global__ __launch_bounds__(64) void k_seg_hyp(const float4 *const *__restrict__ clouds, const int *__restrict__ labels,
                                                SegState *__restrict__ st, int N, SegParams sp, int r)
{
    const int b = blockIdx.x, h = threadIdx.x;
    SegState &s = st[b];
    int plane_closed, plane_count;
    SegRound c = seg_open_round(s, r, sp.percent, plane_closed, plane_count);
    if (!c.done && (s.n_valid < 3 || !((double)c.remaining > (double)sp.percent * (double)s.n_valid))) c.done = 1;
    if (h == 0) {
        s.rs[(r + 1) & 1] = c;
        if (plane_closed >= 0) s.planes[plane_closed].count = plane_count;
        s.lab_count[r & 1] = 0;
#pragma unroll
        for (int k = 0; k < 10; ++k) s.mom[r & 1][k] = 0;
    }
    if (c.done) return;
    s.hyp[h] = seg_make_hyp(clouds[b], labels + (size_t)b * N, N, sp, r, h);
}

#ifdef SEGC_DBG
__device__ long long g_segc_dbg[3][1200 * 8];
#define SEGCT(k) do { if (threadIdx.x == 0 && blockIdx.y == 0 && r < 3) g_segc_dbg[r][((int)blockIdx.z * (int)gridDim.x + (int)blockIdx.x) * 8 + (k)] = (long long)wall_clock64(); } while (0)
#else
#define SEGCT(k) do { } while (0)
#endif
template <bool FUSED, int PTS>
__global__ __launch_bounds__(SEG_BLOCK) void k_seg_count(const float4 *const *__restrict__ clouds, const int *__restrict__ labels,
                                                         SegState *__restrict__ st, int N, SegParams sp, int r)
{
    const int b = blockIdx.y;
    SegState &s = st[b];
    const float4 *__restrict__ cloud = clouds[b];
    const int *__restrict__ lab = labels + (size_t)b * N;
    const int H = sp.hypotheses;
    const int lane = threadIdx.x & 63;
    const int h0 = blockIdx.z * SEG_HGROUP, h1 = min(H, h0 + SEG_HGROUP);
    __shared__ SegHyp hy_sh[SEG_HGROUP];
    __shared__ int bc[SEG_H];
    SEGCT(0);
    // this thread's points first: their loads are in flight while the head below draws the hypotheses (dependent trips of its own)
    float4 q[PTS];
    bool live[PTS];
#pragma unroll
    for (int k = 0; k < PTS; ++k) {
        const int i = (blockIdx.x * PTS + k) * SEG_BLOCK + threadIdx.x;
        live[k] = i < N && lab[i] == -1;
        q[k] = i < N ? cloud[i] : make_float4(0, 0, 0, 0);
    }
    if constexpr (FUSED) {
        int plane_closed, plane_count;
        SegRound c = seg_open_round(s, r, sp.percent, plane_closed, plane_count);
        if (!c.done && (s.n_valid < 3 || !((double)c.remaining > (double)sp.percent * (double)s.n_valid))) c.done = 1;
        const bool recorder = blockIdx.x == 0 && threadIdx.x == 0;
        if (recorder && blockIdx.z == 0) {
            s.rs[(r + 1) & 1] = c;
            if (plane_closed >= 0) s.planes[plane_closed].count = plane_count;
            s.lab_count[r & 1] = 0;
#pragma unroll
            for (int k = 0; k < 10; ++k) s.mom[r & 1][k] = 0;
        }
        if (c.done) return;
        if (threadIdx.x < SEG_HGROUP) {
            const SegHyp hy = seg_make_hyp(cloud, lab, N, sp, r, h0 + (int)threadIdx.x);
            hy_sh[threadIdx.x] = hy;
            if (blockIdx.x == 0 && h0 + (int)threadIdx.x < SEG_H) s.hyp[h0 + threadIdx.x] = hy;
        }
    } else {
        if (s.rs[(r + 1) & 1].done) return;                  // (k_seg_hyp ran before this launch)
        if (threadIdx.x < SEG_HGROUP) hy_sh[threadIdx.x] = s.hyp[h0 + threadIdx.x];
    }
    SEGCT(1);
    if (threadIdx.x < SEG_H) bc[threadIdx.x] = 0;
    __syncthreads();
    SEGCT(2);
    // lane l < SEG_HGROUP holds hypothesis h0 + l; the loop broadcasts it with v_readlane (no dependent scalar loads)
    SegHyp mh = hy_sh[lane < SEG_HGROUP ? lane : 0];
    if (lane >= SEG_HGROUP) mh.ok = 0;
    const unsigned long long okm = __ballot(mh.ok != 0);
    int mine = 0;
    for (int h = h0; h < h1; ++h) {
        const int l = h - h0;
        if (!((okm >> l) & 1ull)) continue;
        const float nx = rdlane(mh.nx, l), ny = rdlane(mh.ny, l), nz = rdlane(mh.nz, l), dd = rdlane(mh.dd, l),
                    t2 = rdlane(mh.thr2nn, l);
        int cc = 0;
#pragma unroll
        for (int k = 0; k < PTS; ++k) cc += __popcll(__ballot(live[k] && seg_inlier(nx, ny, nz, dd, t2, q[k])));
        if (lane == l) mine = cc;
    }
    SEGCT(3);
    if (mine) atomicAdd(&bc[h0 + lane], mine);            // (mine != 0 only in lanes < SEG_HGROUP)
    __syncthreads();
    SEGCT(4);
    if (threadIdx.x < SEG_H && bc[threadIdx.x]) atomicAdd(&s.counts[r & 1][blockIdx.x % SEG_CR][threadIdx.x], bc[threadIdx.x]);
#ifdef SEGC_DBG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SEGCT(5);
}

// P2 tail + P3: every block finds the best hypothesis (max count, smallest h on ties), block 0 records it, then
// the block adds its points' fixed-point moments about the hypothesis' first sample.  grid (ceil(N/1024), B)
template <int PTS>
__global__ __launch_bounds__(SEG_BLOCK) void k_seg_moments(const float4 *const *__restrict__ clouds, const int *__restrict__ labels,
                                                           SegState *__restrict__ st, int N, int H, int r)
{
    const int b = blockIdx.y;
    SegState &s = st[b];
    if (s.rs[(r + 1) & 1].done) return;
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x < SEG_H) {          // the counts of the NEXT round start from zero (nobody reads or adds to them now)
#pragma unroll
        for (int k = 0; k < SEG_CR; ++k) s.counts[(r + 1) & 1][k][threadIdx.x] = 0;
    }
    // argmax over (count desc, h asc): key = count * 64 + (63 - h)
    int key = lane < H ? seg_count_of(s, r, lane) * 64 + (63 - lane) : -1;
    for (int o = 32; o >= 1; o >>= 1) key = max(key, __shfl_xor(key, o));
    const int best = 63 - (key & 63), bc = key >> 6;
    if (bc < 3) return;                       // the label launch raises done
    if (blockIdx.x == 0 && threadIdx.x == 0) s.best = best;
    const SegHyp &hy = s.hyp[best];
    const float nx = hy.nx, ny = hy.ny, nz = hy.nz, dd = hy.dd, t2 = hy.thr2nn;
    const double ox = hy.p0x, oy = hy.p0y, oz = hy.p0z;
    const float4 *__restrict__ cloud = clouds[b];
    const int *__restrict__ lab = labels + (size_t)b * N;
    long long m[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) m[k] = 0;
#pragma unroll
    for (int k = 0; k < PTS; ++k) {
        const int i = (blockIdx.x * PTS + k) * SEG_BLOCK + threadIdx.x;
        if (i < N && lab[i] == -1) {
            const float4 q = cloud[i];
            if (seg_inlier(nx, ny, nz, dd, t2, q)) {
                const long long qx = __double2ll_rn(((double)q.x - ox) * 65536.0), qy = __double2ll_rn(((double)q.y - oy) * 65536.0),
                                qz = __double2ll_rn(((double)q.z - oz) * 65536.0);
                m[0] += 1; m[1] += qx; m[2] += qy; m[3] += qz;
                m[4] += qx * qx; m[5] += qx * qy; m[6] += qx * qz; m[7] += qy * qy; m[8] += qy * qz; m[9] += qz * qz;
            }
        }
    }
    __shared__ unsigned long long bm[10];
    if (threadIdx.x < 10) bm[threadIdx.x] = 0;
    __syncthreads();
    if (__any(m[0] != 0)) {
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            long long v = m[k];
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
            m[k] = v;
        }
        long long mine = 0;
#pragma unroll
        for (int k = 0; k < 10; ++k) if (lane == k) mine = m[k];
        if (lane < 10) atomicAdd(&bm[lane], (unsigned long long)mine);
    }
    __syncthreads();
    if (threadIdx.x < 10 && bm[threadIdx.x]) atomicAdd(reinterpret_cast<unsigned long long *>(&s.mom[r & 1][threadIdx.x]), bm[threadIdx.x]);
}

// P3 tail + P4: covariance -> eigenvector of the smallest eigenvalue -> (n, d), sign rule (thread 0 of every block: same
// inputs, same bits; block 0 records the plane), then the plane's points = unassigned points within thr of it.
// grid (ceil(N/1024), B)
// the refined plane of round r from its moments (P3 tail): values only
__device__ __forceinline__ SegPlane seg_refined_plane(const SegState &s, int r)
{
    const SegHyp &hy = s.hyp[s.best];
    const long long *__restrict__ mom = s.mom[r & 1];
    const double ox = hy.p0x, oy = hy.p0y, oz = hy.p0z;
    const double inv = 1.0 / (double)mom[0];
    const double mx = (double)mom[1] * inv, my = (double)mom[2] * inv, mz = (double)mom[3] * inv;
    Sym3 C;
    C.a00 = (double)mom[4] * inv - mx * mx; C.a01 = (double)mom[5] * inv - mx * my; C.a02 = (double)mom[6] * inv - mx * mz;
    C.a11 = (double)mom[7] * inv - my * my; C.a12 = (double)mom[8] * inv - my * mz; C.a22 = (double)mom[9] * inv - mz * mz;
    double nx, ny, nz;
    eig3_smallest(C, nx, ny, nz);
    const double cx = ox + mx / 65536.0, cy = oy + my / 65536.0, cz = oz + mz / 65536.0;
    double d = -((nx * cx + ny * cy) + nz * cz);
    if (d < 0.0) { nx = -nx; ny = -ny; nz = -nz; d = -d; }      // src/GraphicEnd.cpp:383-387
    SegPlane P;
    P.a = (float)nx; P.b = (float)ny; P.c = (float)nz; P.d = (float)d;
    P.cx = (float)cx; P.cy = (float)cy; P.cz = (float)cz; P.count = 0;
    return P;
}

// the refinement on its own (frames in batches).  grid (B), block 64
__global__ __launch_bounds__(64) void k_seg_refine(SegState *__restrict__ st, int H, int r)
{
    SegState &s = st[blockIdx.x];
    if (s.rs[(r + 1) & 1].done) return;
    const int lane = threadIdx.x;
    int key = lane < H ? seg_count_of(s, r, lane) * 64 + (63 - lane) : -1;
    for (int o = 32; o >= 1; o >>= 1) key = max(key, __shfl_xor(key, o));
    if (lane != 0) return;
    if ((key >> 6) < 3) { s.rs[(r + 1) & 1].done = 1; return; }
    s.planes[r] = seg_refined_plane(s, r);
}

template <bool FUSED, int PTS>
__global__ __launch_bounds__(SEG_BLOCK) void k_seg_label(const float4 *const *__restrict__ clouds, int *__restrict__ labels,
                                                         SegState *__restrict__ st, int N, int H, float thr, int r)
{
    const int b = blockIdx.y;
    SegState &s = st[b];
    if (s.rs[(r + 1) & 1].done) return;
    __shared__ float pl[4];
    if constexpr (FUSED) {
        const int lane = threadIdx.x & 63;
        int key = lane < H ? seg_count_of(s, r, lane) * 64 + (63 - lane) : -1;
        for (int o = 32; o >= 1; o >>= 1) key = max(key, __shfl_xor(key, o));
        if ((key >> 6) < 3) {                                    // no consensus: the loop ends (every block sees the same counts)
            if (blockIdx.x == 0 && threadIdx.x == 0) s.rs[(r + 1) & 1].done = 1;
            return;
        }
        if (threadIdx.x == 0) {
            const SegPlane P = seg_refined_plane(s, r);
            pl[0] = P.a; pl[1] = P.b; pl[2] = P.c; pl[3] = P.d;
            if (blockIdx.x == 0) s.planes[r] = P;
        }
    } else {
        if (threadIdx.x == 0) { const SegPlane &P = s.planes[r]; pl[0] = P.a; pl[1] = P.b; pl[2] = P.c; pl[3] = P.d; }     // (k_seg_refine ran before this launch)
    }
    __syncthreads();
    const float a = pl[0], bb = pl[1], c = pl[2], d = pl[3];
    const float4 *__restrict__ cloud = clouds[b];
    int *__restrict__ lab = labels + (size_t)b * N;
    int got = 0;
#pragma unroll
    for (int k = 0; k < PTS; ++k) {
        const int i = (blockIdx.x * PTS + k) * SEG_BLOCK + threadIdx.x;
        if (i < N && lab[i] == -1) {
            const float4 q = cloud[i];
            const float e = __fmaf_rn(a, q.x, __fmaf_rn(bb, q.y, c * q.z)) + d;
            if (fabsf(e) <= thr) { lab[i] = r; ++got; }
        }
    }
    got = block_sum_int(got);
    if (threadIdx.x == 0 && got) atomicAdd(&s.lab_count[r & 1], got);
}

// bookkeeping of the last round.  grid (B), block 1
__global__ void k_seg_final(SegState *__restrict__ st, int rounds, float percent)
{
    SegState &s = st[blockIdx.x];
    int plane_closed, plane_count;
    const SegRound c = seg_open_round(s, rounds, percent, plane_closed, plane_count);
    if (plane_closed >= 0) s.planes[plane_closed].count = plane_count;
    s.nplanes = c.nplanes;
}

// ------------------------------------------------------------------------------------ round 6: ONE persistent launch per pass (frames alone)
// A frame alone paid 1 + 3 x 3 dependent launches (init; count | moments | label per round): 137 us per frame, 180 us for the two
// frames of a SLAM3D_EST_PLANE alignment -- every launch re-reads the cloud and the labels, and every launch boundary is a drain.
// Here the frame's pixels stay in REGISTERS (point + label, SEG_PPT_MAX per thread) from the first load to the last label; what the
// blocks exchange -- the valid count, the 64 consensus counts, the ten moments, the labelled count -- are device-scope integer atomics
// read back with device-scope loads behind a grid barrier (three per round; no fence: list_icp.hpp explains why), into accumulators of
// their own per round (SegScratch, zeroed by one memset in front of the launch).  The bookkeeping, the hypotheses and the refinement are
// redone by every block from the same numbers, as in the fused launches above; block 0 records what the consumers of SegState read.
// A hypothesis draws RANDOM pixels and needs to know whether they are still unassigned: the labels of other blocks' pixels live in
// their registers, so the draw decides it from the pixel itself -- valid, and within thr of none of the planes refined so far, tested in
// round order -- which is exactly how the label launches assign (spec P4).  Same integers, same floats, same labels and planes as the
// launches above (tests/test_segmentation.py, tests/test_plane_icp.py compare with the oracle and the numpy restatement).
// Co-residency as in list_icp.hpp: G x frames <= 256 blocks of 256 threads, <= 128 VGPRs.
// MEASURED (640x480, one frame, tools/variants -DSEG_DBG, thread 0 of block 0, sums over the three rounds): load 1.5 us, first barrier 7 |
// hypotheses 25, consensus 87, barrier 24 | moments 11, barrier 24 | refinement 19, labels 2.5, barrier 35 = 263 us per frame against
// 139 us for the ten launches.  The consensus phase is THROUGHPUT work (307 k pixels x 64 hypotheses): the launches run it on 4,800
// blocks at full occupancy, the persistent grid -- capped at a quarter of the chip's wave slots so that four launches are always
// co-resident -- runs it with ONE wave per SIMD, where a dependent VALU / ballot chain issues every 10-27 cycles; and every barrier
// waits for the slowest block of such a phase.  A persistent launch pays where an iteration is latency (list_icp.hpp: 0.97 -> 0.66 ms),
// not where it is arithmetic.  Kept OFF (SLAM3D_SEG_PERSIST=1 enables it; tests/test_segmentation.py runs both forms).
constexpr int SEG_PPT_MAX = 10;
struct SegScratch {
    int n_valid; unsigned int ticket; int lab_count[SEG_MAXP]; int pad[6];
    long long mom[SEG_MAXP][10];
    int counts[SEG_MAXP][SEG_CR][SEG_H];
};

// (the pixel is unassigned at the head of round r <=> valid and in none of the planes of rounds 0 .. r-1)
__device__ __forceinline__ bool seg_free_at(const float4 q, float zmax, const SegPlane *pl, int r, float thr)
{
    if (!pt_valid(q.x, q.y, q.z, zmax)) return false;
    for (int k = 0; k < r; ++k) {
        const float e = __fmaf_rn(pl[k].a, q.x, __fmaf_rn(pl[k].b, q.y, pl[k].c * q.z)) + pl[k].d;
        if (fabsf(e) <= thr) return false;
    }
    return true;
}

// seg_make_hyp with the labels decided from the pixels (same draws, same acceptance rule, same picks)
__device__ __forceinline__ SegHyp seg_make_hyp_free(const float4 *__restrict__ cloud, int N, const SegParams &sp, int r, int h, float zmax, const SegPlane *pl)
{
    SegHyp hy;
    hy.nx = hy.ny = hy.nz = hy.dd = hy.thr2nn = 0.0f; hy.ok = 0; hy.p0x = hy.p0y = hy.p0z = 0.0f;
    if (h < sp.hypotheses) {
        unsigned long long x = sp.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(1 + r * 4096 + h);
        int pick0 = -1, pick1 = -1, pick2 = -1, np = 0;
        float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;
        constexpr int SEG_AHEAD = 8;
        int px_[SEG_AHEAD];
        float4 pq_[SEG_AHEAD];
#pragma unroll
        for (int t = 0; t < SEG_AHEAD; ++t) {
            x += 0x9E3779B97F4A7C15ull;
            const unsigned long long o = seg_mix64(x);
            px_[t] = (int)(((o >> 32) * (unsigned long long)N) >> 32);
        }
#pragma unroll
        for (int t = 0; t < SEG_AHEAD; ++t) pq_[t] = cloud[px_[t]];
#pragma unroll
        for (int t = 0; t < SEG_AHEAD; ++t) {
            const int pix = px_[t];
            const bool take = np < 3 && seg_free_at(pq_[t], zmax, pl, r, sp.thr) && !((np > 0 && pick0 == pix) || (np > 1 && pick1 == pix));
            if (take) { if (np == 0) { pick0 = pix; p0 = pq_[t]; } else if (np == 1) { pick1 = pix; p1 = pq_[t]; } else { pick2 = pix; p2 = pq_[t]; } ++np; }
        }
        for (int t = SEG_AHEAD; t < SEG_DRAWS && np < 3; ++t) {
            x += 0x9E3779B97F4A7C15ull;
            const unsigned long long o = seg_mix64(x);
            const int pix = (int)(((o >> 32) * (unsigned long long)N) >> 32);
            const float4 qq = cloud[pix];
            if (!seg_free_at(qq, zmax, pl, r, sp.thr)) continue;
            if ((np > 0 && pick0 == pix) || (np > 1 && pick1 == pix)) continue;
            if (np == 0) { pick0 = pix; p0 = qq; } else if (np == 1) { pick1 = pix; p1 = qq; } else { pick2 = pix; p2 = qq; }
            ++np;
        }
        (void)pick2;
        if (np == 3) {
            const float ax = p1.x - p0.x, ay = p1.y - p0.y, az = p1.z - p0.z;
            const float bx = p2.x - p0.x, by = p2.y - p0.y, bz = p2.z - p0.z;
            const float nx = __fmaf_rn(ay, bz, -(az * by));
            const float ny = __fmaf_rn(az, bx, -(ax * bz));
            const float nz = __fmaf_rn(ax, by, -(ay * bx));
            const float nn = __fmaf_rn(nz, nz, __fmaf_rn(ny, ny, nx * nx));
            if (nn > 1e-16f) {
                hy.nx = nx; hy.ny = ny; hy.nz = nz;
                hy.dd = -__fmaf_rn(nx, p0.x, __fmaf_rn(ny, p0.y, nz * p0.z));
                hy.thr2nn = (sp.thr * sp.thr) * nn;
                hy.ok = 1;
                hy.p0x = p0.x; hy.p0y = p0.y; hy.p0z = p0.z;
            }
        }
    }
    return hy;
}

// the refined plane of a round from its moments and the winning hypothesis' first sample (seg_refined_plane's arithmetic)
__device__ __noinline__ SegPlane seg_refine_from(const long long *mom, double ox, double oy, double oz)      // (out of line: the eigen solve's registers must not push the resident pixels into scratch)
{
    const double inv = 1.0 / (double)mom[0];
    const double mx = (double)mom[1] * inv, my = (double)mom[2] * inv, mz = (double)mom[3] * inv;
    Sym3 C;
    C.a00 = (double)mom[4] * inv - mx * mx; C.a01 = (double)mom[5] * inv - mx * my; C.a02 = (double)mom[6] * inv - mx * mz;
    C.a11 = (double)mom[7] * inv - my * my; C.a12 = (double)mom[8] * inv - my * mz; C.a22 = (double)mom[9] * inv - mz * mz;
    double nx, ny, nz;
    eig3_smallest(C, nx, ny, nz);
    const double cx = ox + mx / 65536.0, cy = oy + my / 65536.0, cz = oz + mz / 65536.0;
    double d = -((nx * cx + ny * cy) + nz * cz);
    if (d < 0.0) { nx = -nx; ny = -ny; nz = -nz; d = -d; }      // src/GraphicEnd.cpp:383-387
    SegPlane P;
    P.a = (float)nx; P.b = (float)ny; P.c = (float)nz; P.d = (float)d;
    P.cx = (float)cx; P.cy = (float)cy; P.cz = (float)cz; P.count = 0;
    return P;
}

// grid (G, frames), block 256; ppt = pixels per thread (pixel i = (blockIdx.x * ppt + k) * 256 + thread, k < ppt <= SEG_PPT_MAX).
// The block's pixels live in LDS (x, y, z planes + a label byte: 33 KB at ten pixels per thread; kept in registers they and the
// eigen solve / the eight look-ahead draws did not fit 128 VGPRs -- 48 spilled registers).  Every phase walks k = 0 .. ppt-1 once.

#ifdef SEG_DBG
#define SEGT(k) do { if (rec && b == 0) { const long long n_ = (long long)wall_clock64(); tph[k] += n_ - tprev; tprev = n_; } } while (0)
#else
#define SEGT(k) do { } while (0)
#endif
__global__ __launch_bounds__(SEG_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_seg_persist(
    const float4 *const *__restrict__ clouds, int *__restrict__ labels, SegState *__restrict__ st, SegScratch *__restrict__ scr, int N, int ppt, float zmax, SegParams sp)
{
    __shared__ float sx[SEG_PPT_MAX][SEG_BLOCK], sy[SEG_PPT_MAX][SEG_BLOCK], sz[SEG_PPT_MAX][SEG_BLOCK];
    __shared__ signed char sl[SEG_PPT_MAX][SEG_BLOCK];           // -3 no pixel, -2 invalid, -1 unassigned, r >= 0 plane
    __shared__ SegHyp hy_sh[SEG_H];
    __shared__ SegPlane pl_sh[SEG_MAXP];
    __shared__ int bc[SEG_H];
    __shared__ unsigned long long bm[10];
    __shared__ int part[SEG_BLOCK / 64];
    const int b = blockIdx.y, G = gridDim.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    SegState &s = st[b];
    SegScratch &x = scr[b];
    const float4 *__restrict__ cloud = clouds[b];
    const int H = sp.hypotheses;
    const bool rec = blockIdx.x == 0 && tid == 0;
    unsigned int epoch = 0;
    long long tph[10] = {0,0,0,0,0,0,0,0,0,0}, tprev = (long long)wall_clock64(); (void)tph; (void)tprev;
    auto grid_barrier = [&]() __attribute__((always_inline)) {
        ls_barrier();
        if (w == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (all of a block's global atomics are issued by wave 0: they have been performed)
            epoch += 1;
            if (lane == 0) __hip_atomic_fetch_add(&x.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ls_wait_ticket(&x.ticket, epoch * (unsigned int)G);       // (list_icp.hpp: bounded -- after ~2 s the abort bit opens every barrier; the end of the kernel reports it)
        }
        ls_barrier();
    };
    auto block_sum = [&](int v) __attribute__((always_inline)) {      // valid in every thread
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        ls_barrier();
        if (lane == 0) part[w] = v;
        ls_barrier();
        return part[0] + part[1] + part[2] + part[3];
    };
    // ---- P1: the block's pixels, resident from here on
    int nv = 0;
    for (int k = 0; k < ppt; ++k) {
        const int i = (blockIdx.x * ppt + k) * SEG_BLOCK + tid;
        float4 qq = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        int l = -3;
        if (i < N) {
            qq = cloud[i];
            const bool ok = pt_valid(qq.x, qq.y, qq.z, zmax);
            l = ok ? -1 : -2;
            nv += ok ? 1 : 0;
        }
        sx[k][tid] = qq.x; sy[k][tid] = qq.y; sz[k][tid] = qq.z; sl[k][tid] = (signed char)l;
    }
    nv = block_sum(nv);
    if (tid == 0 && nv) __hip_atomic_fetch_add(&x.n_valid, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SEGT(0);
    grid_barrier();
    SEGT(1);
    const int n_valid = __hip_atomic_load(&x.n_valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SegRound c;
    c.remaining = 0; c.nplanes = 0; c.done = 0; c.pad = 0;
    int best = 0, last_got = 0;
    for (int r = 0; r < sp.max_planes; ++r) {
        // ---- bookkeeping of the round that ended + this round's loop condition (seg_open_round on the block's own copy)
        last_got = 0;
        if (r == 0) c.remaining = n_valid;
        else if (!c.done) {
            const int got = __hip_atomic_load(&x.lab_count[r - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got == 0) c.done = 1;
            else { c.nplanes = r; c.remaining -= got; if (rec) s.planes[r - 1].count = got; }
        }
        if (!c.done && (n_valid < 3 || !((double)c.remaining > (double)sp.percent * (double)n_valid))) c.done = 1;
        if (c.done) break;
        // ---- P2: the 64 hypotheses (every block the same), then their consensus among this block's unassigned pixels
        if (tid < SEG_H) {
            const SegHyp hy = seg_make_hyp_free(cloud, N, sp, r, tid, zmax, pl_sh);
            hy_sh[tid] = hy;
            if (blockIdx.x == 0) s.hyp[tid] = hy;
            bc[tid] = 0;
        }
        ls_barrier();
        SEGT(2);
        {
            const SegHyp mh = hy_sh[lane];
            const unsigned long long okm = __ballot(mh.ok != 0 && lane < H);
            int mine = 0;
            // (hypothesis outer, pixels inner and in registers for this phase: one v_readlane set per hypothesis serves all the thread's
            //  pixels, whose inlier tests are independent chains -- with ONE wave per SIMD a dependent chain issues every ~10 cycles)
            float px_[SEG_PPT_MAX], py_[SEG_PPT_MAX], pz_[SEG_PPT_MAX];
            bool live[SEG_PPT_MAX];
#pragma unroll
            for (int k = 0; k < SEG_PPT_MAX; ++k) {
                live[k] = k < ppt && sl[k][tid] == -1;
                px_[k] = k < ppt ? sx[k][tid] : 0.0f; py_[k] = k < ppt ? sy[k][tid] : 0.0f; pz_[k] = k < ppt ? sz[k][tid] : 0.0f;
            }
            for (int h = 0; h < H; ++h) {
                if (!((okm >> h) & 1ull)) continue;
                const float nx = rdlane(mh.nx, h), ny = rdlane(mh.ny, h), nz = rdlane(mh.nz, h), dd = rdlane(mh.dd, h), t2 = rdlane(mh.thr2nn, h);
                int cc = 0;
#pragma unroll
                for (int k = 0; k < SEG_PPT_MAX; ++k) cc += __popcll(__ballot(live[k] && seg_inlier(nx, ny, nz, dd, t2, make_float4(px_[k], py_[k], pz_[k], 0.0f))));
                if (lane == h) mine = cc;
            }
            if (mine) atomicAdd(&bc[lane], mine);
        }
        ls_barrier();
        if (tid < SEG_H && bc[tid]) __hip_atomic_fetch_add(&x.counts[r][blockIdx.x % SEG_CR][tid], bc[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SEGT(3);
        grid_barrier();
        SEGT(4);
        // ---- P2 tail: the best hypothesis (max count, smallest h on ties); P3: moments of its inliers about its first sample
        {
            int cnt = 0;
            if (lane < H)
#pragma unroll
                for (int k = 0; k < SEG_CR; ++k) cnt += __hip_atomic_load(&x.counts[r][k][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int key = lane < H ? cnt * 64 + (63 - lane) : -1;
            for (int o = 32; o >= 1; o >>= 1) key = max(key, __shfl_xor(key, o));
            best = 63 - (key & 63);
            if ((key >> 6) < 3) { c.done = 1; break; }             // no consensus: the loop ends (every block sees the same counts)
        }
        if (rec) s.best = best;
        const SegHyp hb = hy_sh[best];
        {
            const double ox = hb.p0x, oy = hb.p0y, oz = hb.p0z;
            long long m[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) m[k] = 0;
            for (int k = 0; k < ppt; ++k) {
                const float4 qq = make_float4(sx[k][tid], sy[k][tid], sz[k][tid], 0.0f);
                if (sl[k][tid] == -1 && seg_inlier(hb.nx, hb.ny, hb.nz, hb.dd, hb.thr2nn, qq)) {
                    const long long qx = __double2ll_rn(((double)qq.x - ox) * 65536.0), qy = __double2ll_rn(((double)qq.y - oy) * 65536.0),
                                    qz = __double2ll_rn(((double)qq.z - oz) * 65536.0);
                    m[0] += 1; m[1] += qx; m[2] += qy; m[3] += qz;
                    m[4] += qx * qx; m[5] += qx * qy; m[6] += qx * qz; m[7] += qy * qy; m[8] += qy * qz; m[9] += qz * qz;
                }
            }
            if (tid < 10) bm[tid] = 0;
            ls_barrier();
            if (__any(m[0] != 0)) {
#pragma unroll
                for (int k = 0; k < 10; ++k) {
                    long long v = m[k];
                    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
                    m[k] = v;
                }
                long long mine = 0;
#pragma unroll
                for (int k = 0; k < 10; ++k) if (lane == k) mine = m[k];
                if (lane < 10) atomicAdd(&bm[lane], (unsigned long long)mine);
            }
            ls_barrier();
            if (tid < 10 && bm[tid]) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&x.mom[r][tid]), bm[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        SEGT(5);
        grid_barrier();
        SEGT(6);
        // ---- P3 tail + P4: the refined plane (thread 0 of every block: same moments, same bits), then the plane's pixels
        if (tid == 0) {
            long long mom[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) mom[k] = (long long)__hip_atomic_load(reinterpret_cast<unsigned long long *>(&x.mom[r][k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const SegPlane P = seg_refine_from(mom, hb.p0x, hb.p0y, hb.p0z);
            pl_sh[r] = P;
            if (blockIdx.x == 0) s.planes[r] = P;
        }
        ls_barrier();
        SEGT(7);
        {
            const float a = pl_sh[r].a, bb = pl_sh[r].b, cc_ = pl_sh[r].c, d = pl_sh[r].d;
            int got = 0;
            for (int k = 0; k < ppt; ++k) {
                if (sl[k][tid] == -1) {
                    const float e = __fmaf_rn(a, sx[k][tid], __fmaf_rn(bb, sy[k][tid], cc_ * sz[k][tid])) + d;
                    if (fabsf(e) <= sp.thr) { sl[k][tid] = (signed char)r; ++got; }
                }
            }
            got = block_sum(got);
            if (tid == 0 && got) __hip_atomic_fetch_add(&x.lab_count[r], got, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        SEGT(8);
        grid_barrier();
        SEGT(9);
        last_got = r == sp.max_planes - 1 ? __hip_atomic_load(&x.lab_count[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    }
    // ---- the labels, and the state seg_open_round(s, max_planes) -- k_seg_final / k_plane_normals -- closes the last round from
    int *__restrict__ lab = labels + (size_t)b * N;
    for (int k = 0; k < ppt; ++k) {
        const int i = (blockIdx.x * ppt + k) * SEG_BLOCK + tid;
        if (i < N) lab[i] = (int)sl[k][tid];
    }
#ifdef SEG_DBG
    if (rec && b == 0) printf("seg phases us: load %.1f bar %.1f | hyp %.1f count %.1f bar %.1f | mom %.1f bar %.1f | refine %.1f label %.1f bar %.1f\n", tph[0]*.01, tph[1]*.01, tph[2]*.01, tph[3]*.01, tph[4]*.01, tph[5]*.01, tph[6]*.01, tph[7]*.01, tph[8]*.01, tph[9]*.01);
#endif
    if (rec) {
        // (a barrier that timed out -- the blocks were not all resident -- left garbage everywhere: the state says so, the host reader refuses it)
        s.n_valid = (__hip_atomic_load(&x.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & LS_ABORT) ? -1 : n_valid;
        s.rs[0] = c; s.rs[1] = c;
        s.lab_count[0] = 0; s.lab_count[1] = 0;
        if (!c.done) s.lab_count[(sp.max_planes - 1) & 1] = last_got;      // (the loop ran its last round to the end: its labelled count is still to be booked)
    }
}

// ------------------------------------------------------------------------------------ a6: planes from labels
// slam3d_fit_planes: per-plane least-squares fit for GIVEN labels (the refinement PCL runs inside
// SACSegmentation::segment, src/GraphicEnd.cpp:360-375).  Same arithmetic as P3 with the sensor origin as the
// moment origin: integer fixed-point (2^-16 m) moments -> covariance -> smallest eigenvector -> (n, d), d >= 0.
constexpr int FIT_MAXP = 16;
struct FitState { long long mom[FIT_MAXP][10]; SegPlane out[FIT_MAXP]; };

// grid (ceil(N/1024)), block 256: every plane present in the block is summed (wave shuffles -> LDS -> one set of
// atomics per block and plane)
__global__ __launch_bounds__(SEG_BLOCK) void k_fit_moments(const float4 *__restrict__ cloud, const int *__restrict__ labels, int N,
                                                           int nplanes, FitState *__restrict__ st)
{
    __shared__ unsigned long long bm[10];
    __shared__ int present;
    const int lane = threadIdx.x & 63;
    long long q[SEG_PTS][3];
    int lab[SEG_PTS];
#pragma unroll
    for (int k = 0; k < SEG_PTS; ++k) {
        const int i = (blockIdx.x * SEG_PTS + k) * SEG_BLOCK + threadIdx.x;
        lab[k] = -1;
        q[k][0] = q[k][1] = q[k][2] = 0;
        if (i < N) {
            lab[k] = labels[i];
            if (lab[k] >= 0 && lab[k] < nplanes) {
                const float4 p = cloud[i];
                q[k][0] = __double2ll_rn((double)p.x * 65536.0); q[k][1] = __double2ll_rn((double)p.y * 65536.0);
                q[k][2] = __double2ll_rn((double)p.z * 65536.0);
            }
        }
    }
    for (int pl = 0; pl < nplanes; ++pl) {
        __syncthreads();
        if (threadIdx.x < 10) bm[threadIdx.x] = 0;
        if (threadIdx.x == 0) present = 0;
        __syncthreads();
        long long m[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) m[k] = 0;
#pragma unroll
        for (int k = 0; k < SEG_PTS; ++k)
            if (lab[k] == pl) {
                const long long x = q[k][0], y = q[k][1], z = q[k][2];
                m[0] += 1; m[1] += x; m[2] += y; m[3] += z;
                m[4] += x * x; m[5] += x * y; m[6] += x * z; m[7] += y * y; m[8] += y * z; m[9] += z * z;
            }
        if (__any(m[0] != 0)) {
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                long long v = m[k];
                for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
                m[k] = v;
            }
            long long mine = 0;
#pragma unroll
            for (int k = 0; k < 10; ++k) if (lane == k) mine = m[k];
            if (lane < 10) atomicAdd(&bm[lane], (unsigned long long)mine);
            if (lane == 0) present = 1;
        }
        __syncthreads();
        if (present && threadIdx.x < 10) atomicAdd(reinterpret_cast<unsigned long long *>(&st->mom[pl][threadIdx.x]), bm[threadIdx.x]);
    }
}

// one lane per plane
__global__ __launch_bounds__(64) void k_fit_refine(FitState *__restrict__ st, int nplanes)
{
    const int pl = threadIdx.x;
    if (pl >= nplanes) return;
    const long long *m = st->mom[pl];
    SegPlane &P = st->out[pl];
    P.a = P.b = P.c = P.d = P.cx = P.cy = P.cz = 0.0f;
    P.count = (int)m[0];
    if (m[0] < 3) return;
    const double inv = 1.0 / (double)m[0];
    const double mx = (double)m[1] * inv, my = (double)m[2] * inv, mz = (double)m[3] * inv;
    Sym3 C;
    C.a00 = (double)m[4] * inv - mx * mx; C.a01 = (double)m[5] * inv - mx * my; C.a02 = (double)m[6] * inv - mx * mz;
    C.a11 = (double)m[7] * inv - my * my; C.a12 = (double)m[8] * inv - my * mz; C.a22 = (double)m[9] * inv - mz * mz;
    double nx, ny, nz;
    eig3_smallest(C, nx, ny, nz);
    const double cx = mx / 65536.0, cy = my / 65536.0, cz = mz / 65536.0;
    double d = -((nx * cx + ny * cy) + nz * cz);
    if (d < 0.0) { nx = -nx; ny = -ny; nz = -nz; d = -d; }      // src/GraphicEnd.cpp:383-387
    P.a = (float)nx; P.b = (float)ny; P.c = (float)nz; P.d = (float)d;
    P.cx = (float)cx; P.cy = (float)cy; P.cz = (float)cz;
}

// ------------------------------------------------------------------------------------ S2p: per-plane normals (SLAM3D_EST_PLANE)
// SURVEY.md App. C2, per-plane variant: "points take their plane's normal".  One task = one frame that was just segmented
// (labels + SegState of scratch slot k): the pixel labelled with plane r gets (a, b, c, 1 + r) -- the fit's normal, d >= 0, i.e.
// toward the camera like the window normals --, a pixel on no plane keeps the 7x7-window normal k_normals left there with
// w = 0.75 ("a normal, no plane"; plane_only: nothing), and the frame's plane table is recorded for the pair gate / the caller.
// oracle/icp_oracle.c::orc_plane_normals.  grid (ceil(N / 256), tasks), block 256
struct PlaneTask { const int *lab; const SegState *st; float4 *nrm; FramePlanes *out; int window, pad; };      // window: pixels on no plane keep the 7x7-window normal already in nrm
constexpr int PLANE_ARGS = 32;
struct PlaneTasks { PlaneTask t[PLANE_ARGS]; };
__global__ __launch_bounds__(256) void k_plane_normals(PlaneTasks a, int N, int rounds, float percent)
{
    const PlaneTask &t = a.t[blockIdx.y];
    const SegState &s = *t.st;
    // the bookkeeping of the last round (what k_seg_final does for slam3d_segment_planes*): values only, every block the same
    int plane_closed, plane_count;
    const SegRound fin = seg_open_round(s, rounds, percent, plane_closed, plane_count);
    const int np = fin.nplanes;
    if (blockIdx.x == 0 && threadIdx.x <= SEG_MAXP) {
        if (threadIdx.x == SEG_MAXP) t.out->n = np;
        else {
            const SegPlane &q = s.planes[threadIdx.x];
            FramePlane o = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0 };
            if ((int)threadIdx.x < np) {
                o.a = q.a; o.b = q.b; o.c = q.c; o.d = q.d; o.cx = q.cx; o.cy = q.cy; o.cz = q.cz;
                o.count = (int)threadIdx.x == plane_closed ? plane_count : q.count;
            }
            t.out->pl[threadIdx.x] = o;
        }
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int r = t.lab[i];
    float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (r >= 0 && r < np) {
        const SegPlane &q = s.planes[r];
        o = make_float4(q.a, q.b, q.c, (float)(1 + r));
    } else if (t.window) {
        const float4 w = t.nrm[i];
        if (w.w > 0.5f) o = make_float4(w.x, w.y, w.z, 0.75f);
    }
    t.nrm[i] = o;
}

// (spec S4p's association of the pair gate runs inside k_pair_init, icp_kernels.hpp.)

constexpr int PTR_ARGS = 32;
struct PtrArgs { const float4 *p[PTR_ARGS]; };
__global__ void k_set_ptrs(const float4 **__restrict__ dst, PtrArgs a, int n)
{
    if ((int)threadIdx.x < n) dst[threadIdx.x] = a.p[threadIdx.x];
}

}  // namespace s3d
