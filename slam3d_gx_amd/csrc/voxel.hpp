// voxel.hpp -- PassThrough + VoxelGrid of GraphicEnd::readimage on gfx950.
//
// Replaces pcl::PassThrough (z in [0, z_filter]) and pcl::VoxelGrid (cubic leaf grid_leaf = 0.03) of
// src/GraphicEnd.cpp:283-295 for the 16-byte {x, y, z, rgba} records of the reference's binary PCD files.
// oracle/voxel_oracle.c is the CPU twin.  The centroids come from integer fixed-point sums (2^-20 m) and
// integer colour sums, so the atomics below give the same bits whatever order they land in.
//
//   k_voxel_insert   one thread per point: 64-bit voxel key (iz,iy,ix) -> open-addressing hash table in HBM
//                    (capacity >= 2n, linear probing, atomicCAS on the key), atomic adds into the entry
//   k_voxel_compact  occupied slots -> dense list (one global ticket per block) + histogram of the (iz, iy) rows
//   k_voxel_scan1/2 / k_voxel_scatter   counting sort of the list by row (iz, iy are the most significant key fields)
//   k_voxel_rank     output order = ascending key, like PCL's sorted linear voxel index: rank = start of the
//                    rows a block touches + number of smaller keys among them (LDS-tiled compares), centroid
//                    written at that rank.  No host round trip anywhere: the entry count stays on the device.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace s3d {

constexpr unsigned long long VOX_EMPTY = ~0ull;
constexpr int VOX_BLOCK = 256;
constexpr int VOX_TILE = 2048;        // keys staged in LDS per step of the ranking kernel

struct VoxTable {                     // SoA hash table, `cap` slots (power of two)
    unsigned long long *key;
    long long *sx, *sy, *sz;
    unsigned int *c0, *c1, *c2, *c3, *n;
    int cap;
};

__device__ __forceinline__ unsigned long long vox_key(float x, float y, float z, float inv_leaf)
{
    const long long ix = (long long)floorf(x * inv_leaf) + 1048576, iy = (long long)floorf(y * inv_leaf) + 1048576,
                    iz = (long long)floorf(z * inv_leaf) + 1048576;
    const unsigned long long cx = (unsigned long long)(ix < 0 ? 0 : (ix > 2097151 ? 2097151 : ix));
    const unsigned long long cy = (unsigned long long)(iy < 0 ? 0 : (iy > 2097151 ? 2097151 : iy));
    const unsigned long long cz = (unsigned long long)(iz < 0 ? 0 : (iz > 2097151 ? 2097151 : iz));
    return (cz << 42) | (cy << 21) | cx;
}

__device__ __forceinline__ unsigned int vox_hash(unsigned long long k)
{
    k = (k ^ (k >> 30)) * 0xBF58476D1CE4E5B9ull;
    k = (k ^ (k >> 27)) * 0x94D049BB133111EBull;
    return (unsigned int)(k ^ (k >> 31));
}

// inclusive sum over the lanes of the same run (runs = maximal stretches of consecutive lanes with equal keys,
// numbered by `seg`): Hillis-Steele with a segment test; integer adds, so exact
template <typename T> __device__ __forceinline__ T run_scan(T v, int seg)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T u = __shfl_up(v, o);
        const int su = __shfl_up(seg, o);
        if (lane >= o && su == seg) v += u;
    }
    return v;
}

// One thread per point.  Neighbouring pixels of an organized cloud fall into the same voxel 3-10 times in a row,
// and atomics of one wave to one address serialise, so each run of equal keys inside the wave is summed first
// (integers: same bits) and only the last lane of the run touches the table.
__global__ __launch_bounds__(VOX_BLOCK) void k_voxel_insert(const float4 *__restrict__ pts, int n, float inv_leaf, float zmin, float zmax,
                                                            VoxTable t)
{
    const int i = blockIdx.x * VOX_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    float4 p = make_float4(0.0f, 0.0f, -1.0f, 0.0f);
    if (i < n) p = pts[i];
    const bool ok = i < n && isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && p.z >= zmin && p.z <= zmax;     // PassThrough
    const unsigned long long key = ok ? vox_key(p.x, p.y, p.z, inv_leaf) : VOX_EMPTY;
    const unsigned long long prev = __shfl_up(key, 1);
    const bool head = lane == 0 || prev != key;
    const unsigned long long heads = __ballot(head);
    const int seg = __popcll(heads & ((2ull << lane) - 1ull));                  // run number of this lane
    const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    const unsigned int rgba = (unsigned int)__float_as_int(p.w);
    const long long qx = run_scan(ok ? __double2ll_rn((double)p.x * 1048576.0) : 0ll, seg);
    const long long qy = run_scan(ok ? __double2ll_rn((double)p.y * 1048576.0) : 0ll, seg);
    const long long qz = run_scan(ok ? __double2ll_rn((double)p.z * 1048576.0) : 0ll, seg);
    // colour channels and the count travel packed: (c0 | c1 << 32), (c2 | c3 << 32) and the count stay below 2^32 each
    // inside a wave (64 x 255)
    const unsigned long long c01 = run_scan((unsigned long long)(rgba & 0xffu) | ((unsigned long long)((rgba >> 8) & 0xffu) << 32), seg);
    const unsigned long long c23 = run_scan((unsigned long long)((rgba >> 16) & 0xffu) | ((unsigned long long)(rgba >> 24) << 32), seg);
    const int cnt = run_scan(1, seg);
    if (!ok || !tail) return;
    unsigned int s = vox_hash(key) & (unsigned int)(t.cap - 1);
    for (;;) {
        const unsigned long long was = atomicCAS(t.key + s, VOX_EMPTY, key);
        if (was == VOX_EMPTY || was == key) break;
        s = (s + 1) & (unsigned int)(t.cap - 1);
    }
    atomicAdd(reinterpret_cast<unsigned long long *>(t.sx + s), (unsigned long long)qx);
    atomicAdd(reinterpret_cast<unsigned long long *>(t.sy + s), (unsigned long long)qy);
    atomicAdd(reinterpret_cast<unsigned long long *>(t.sz + s), (unsigned long long)qz);
    atomicAdd(t.c0 + s, (unsigned int)c01);
    atomicAdd(t.c1 + s, (unsigned int)(c01 >> 32));
    atomicAdd(t.c2 + s, (unsigned int)c23);
    atomicAdd(t.c3 + s, (unsigned int)(c23 >> 32));
    atomicAdd(t.n + s, (unsigned int)cnt);
}

constexpr int VOX_BZ = 512, VOX_BY = 256;   // ordering bins: (iz, iy) rows, both clamped (the order stays monotone)
constexpr int zbias = 128;                   // iz in [-128, 383] has its own slab (-3.8 m .. 11.5 m at a 3 cm leaf)
constexpr int VOX_BINS = VOX_BZ * VOX_BY;
constexpr int VOX_SPT = 16;           // table slots per thread in k_voxel_compact

// bin = the (iz, iy) row of the voxel: a monotone function of the key (clamping only merges rows at the ends), so
// bins are ordered like keys and a row holds at most one voxel per ix -- a few hundred entries even for a wall
// that fills a whole z slab
__device__ __forceinline__ int vox_bin(unsigned long long key)
{
    const int iz = (int)(key >> 42) - 1048576 + (int)zbias;      // clouds in a world frame may have negative z
    const int iy = (int)((key >> 21) & 0x1FFFFF) - 1048576 + VOX_BY / 2;
    // slabs outside the table collapse into the first / last BIN (not row: rows of different slabs must not interleave)
    if (iz < 0) return 0;
    if (iz > VOX_BZ - 1) return VOX_BINS - 1;
    const int by = iy < 0 ? 0 : (iy > VOX_BY - 1 ? VOX_BY - 1 : iy);
    return iz * VOX_BY + by;
}

// occupied slots -> dense (key, slot) list + histogram of the (iz, iy) rows.  Each block scans 4096 slots and
// takes ONE ticket from the global counter (same-address global atomics serialise: one per wave cost 166 us).
__global__ __launch_bounds__(VOX_BLOCK) void k_voxel_compact(VoxTable t, unsigned long long *__restrict__ lkey, int *__restrict__ lslot,
                                                             int *__restrict__ m, int *__restrict__ hist)
{
    __shared__ int wsum[VOX_BLOCK / 64], base_sh;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int s0 = blockIdx.x * VOX_BLOCK * VOX_SPT;
    unsigned long long k[VOX_SPT];
    int mine = 0;
#pragma unroll
    for (int j = 0; j < VOX_SPT; ++j) {
        const int s = s0 + j * VOX_BLOCK + threadIdx.x;
        k[j] = s < t.cap ? t.key[s] : VOX_EMPTY;
        mine += k[j] != VOX_EMPTY ? 1 : 0;
    }
    int incl = mine;                                             // inclusive prefix over the wave
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        base_sh = tot ? atomicAdd(m, tot) : 0;
    }
    __syncthreads();
    int pos = base_sh + incl - mine;
    for (int q = 0; q < w; ++q) pos += wsum[q];
#pragma unroll
    for (int j = 0; j < VOX_SPT; ++j) {
        if (k[j] != VOX_EMPTY) {
            lkey[pos] = k[j];
            lslot[pos] = s0 + j * VOX_BLOCK + threadIdx.x;
            atomicAdd(hist + vox_bin(k[j]), 1);                  // 65536 rows, hashed order: no address is hot
            ++pos;
        }
    }
}

// exclusive prefix of the row histogram -> start[]; cursor[] reset.  Two launches of VOX_BINS / 1024 blocks:
// scan1 scans 1024 rows per block (wave shuffles + one LDS step) and leaves the block totals, scan2 adds to every
// block the sum of the totals before it (128 values, read by every block).
constexpr int VOX_SCAN_BLOCKS = VOX_BINS / 1024;
__global__ __launch_bounds__(1024) void k_voxel_scan1(const int *__restrict__ hist, int *__restrict__ start, int *__restrict__ cursor,
                                                      int *__restrict__ btot)
{
    __shared__ int wtot[16];
    const int i = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int v = hist[i];
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    int before = 0;
    for (int q = 0; q < w; ++q) before += wtot[q];
    start[i] = before + incl - v;                      // exclusive prefix inside the block
    cursor[i] = 0;
    if (threadIdx.x == 1023) btot[blockIdx.x] = before + incl;
}
__global__ __launch_bounds__(1024) void k_voxel_scan2(int *__restrict__ start, const int *__restrict__ btot)
{
    __shared__ int off_sh;
    if (threadIdx.x < 64) {
        int a = 0;
        for (int q = threadIdx.x; q < (int)blockIdx.x; q += 64) a += btot[q];
        for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
        if (threadIdx.x == 0) off_sh = a;
    }
    __syncthreads();
    start[blockIdx.x * 1024 + threadIdx.x] += off_sh;
    if (blockIdx.x == VOX_SCAN_BLOCKS - 1 && threadIdx.x == 1023) start[VOX_BINS] = off_sh + btot[blockIdx.x];
}

// group the list by row (order inside a row is arbitrary; k_voxel_rank fixes it)
__global__ __launch_bounds__(VOX_BLOCK) void k_voxel_scatter(const unsigned long long *__restrict__ lkey, const int *__restrict__ lslot,
                                                             const int *__restrict__ m, const int *__restrict__ start,
                                                             int *__restrict__ cursor, unsigned long long *__restrict__ gkey,
                                                             int *__restrict__ gslot)
{
    const int e = blockIdx.x * VOX_BLOCK + threadIdx.x;
    const bool live = e < *m;
    const unsigned long long k = live ? lkey[e] : 0ull;
    const int b = live ? vox_bin(k) : 0;
    if (!live) return;
    const int pos = start[b] + atomicAdd(cursor + b, 1);
    gkey[pos] = k;
    gslot[pos] = lslot[e];
}

// Output order = ascending key, like PCL's sorted linear voxel index.  Keys are grouped by (iz, iy) row and rows
// are ordered, so an entry's rank = start of the first row its block touches + the keys of the block's rows below
// its own: the block stages the union of its entries' rows through LDS.  grid covers the worst case (n entries);
// blocks beyond *m leave at once.
__global__ __launch_bounds__(VOX_BLOCK) void k_voxel_rank(VoxTable t, const unsigned long long *__restrict__ gkey,
                                                          const int *__restrict__ gslot, const int *__restrict__ m,
                                                          const int *__restrict__ start, float4 *__restrict__ out)
{
    __shared__ unsigned long long tile[VOX_TILE];
    __shared__ int lo_sh, hi_sh;
    const int M = *m;
    const int e0 = blockIdx.x * VOX_BLOCK;
    if (e0 >= M) return;
    const int e = e0 + threadIdx.x;
    const bool live = e < M;
    const unsigned long long mine = live ? gkey[e] : 0ull;
    const int b = live ? vox_bin(mine) : 0;
    const int my_lo = live ? start[b] : 0, my_hi = live ? start[b + 1] : 0;
    // (dead lanes: mine = 0 is below every key, their count is unused)
    if (threadIdx.x == 0) lo_sh = my_lo;                                   // first entry's slab starts the union
    if (e == min(M, e0 + VOX_BLOCK) - 1) hi_sh = my_hi;                    // last entry's slab ends it
    __syncthreads();
    // everything before the union is smaller and everything after it larger, so counting the smaller keys of the
    // whole union (fixed-length, unrolled, LDS-broadcast reads; padding = EMPTY never counts) gives the rank
    const int lo = lo_sh, hi = hi_sh;
    int rank = lo;
    for (int t0 = lo; t0 < hi; t0 += VOX_TILE) {
        const int cnt = min(VOX_TILE, hi - t0);
        const int padded = (cnt + 7) & ~7;
        __syncthreads();
        for (int k = threadIdx.x; k < padded; k += VOX_BLOCK) tile[k] = k < cnt ? gkey[t0 + k] : VOX_EMPTY;
        __syncthreads();
        for (int k = 0; k < padded; k += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += tile[k + u] < mine ? 1 : 0;
        }
    }
    if (!live) return;
    const int s = gslot[e];
    const unsigned int ni = t.n[s];
    const double cnt = (double)ni;
    float4 o;
    o.x = (float)(((double)t.sx[s] / cnt) / 1048576.0);
    o.y = (float)(((double)t.sy[s] / cnt) / 1048576.0);
    o.z = (float)(((double)t.sz[s] / cnt) / 1048576.0);
    const unsigned int rgba = (t.c0[s] / ni) | ((t.c1[s] / ni) << 8) | ((t.c2[s] / ni) << 16) | ((t.c3[s] / ni) << 24);
    o.w = __int_as_float((int)rgba);
    out[rank] = o;
}

// src/saveOutput.cpp:84-92: PassThrough z in [0, z_max] on a (down-sampled) keyframe cloud, then
// pcl::transformPointCloud by the keyframe's pose.  Spec T1: a kept record becomes (float)(R p + t), each
// coordinate = fma(R_r2, z, fma(R_r1, y, R_r0 * x)) + t_r in double; a dropped record becomes NaN (the voxel
// grid that follows ignores it), so the output keeps the input order and no compaction is needed.
struct Pose34 { double m[12]; };
__global__ __launch_bounds__(VOX_BLOCK) void k_pass_transform(const float4 *__restrict__ pts, int n, float zmax, Pose34 P,
                                                              float4 *__restrict__ out, int *__restrict__ kept)
{
    const int i = blockIdx.x * VOX_BLOCK + threadIdx.x;
    bool ok = false;
    float4 o = make_float4(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000), __int_as_float(0x7fc00000), 0.0f);
    if (i < n) {
        const float4 p = pts[i];
        ok = isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && p.z >= 0.0f && p.z <= zmax;
        if (ok) {
            const double x = p.x, y = p.y, z = p.z;
            o.x = (float)(fma(P.m[2], z, fma(P.m[1], y, P.m[0] * x)) + P.m[3]);
            o.y = (float)(fma(P.m[6], z, fma(P.m[5], y, P.m[4] * x)) + P.m[7]);
            o.z = (float)(fma(P.m[10], z, fma(P.m[9], y, P.m[8] * x)) + P.m[11]);
            o.w = p.w;
        }
        out[i] = o;
    }
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(kept, __popcll(m));
}

}  // namespace s3d
