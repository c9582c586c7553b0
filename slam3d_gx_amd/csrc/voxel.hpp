// voxel.hpp -- PassThrough + VoxelGrid of GraphicEnd::readimage on gfx950.
//
// Replaces pcl::PassThrough (z in [0, z_filter]) and pcl::VoxelGrid (cubic leaf grid_leaf = 0.03) of
// src/GraphicEnd.cpp:283-295 for the 16-byte {x, y, z, rgba} records of the reference's binary PCD files.
// oracle/voxel_oracle.c is the CPU twin.  The centroids come from integer fixed-point sums (2^-20 m) and
// integer colour sums, so the atomics below give the same bits whatever order they land in.
//
//   k_voxel_insert   one thread per point: 64-bit voxel key (iz,iy,ix) -> open-addressing hash table in HBM
//                    (capacity >= 2n, linear probing, atomicCAS on the key), atomic adds into the entry
//   k_voxel_compact  occupied slots -> dense list (one global ticket per block) + histogram of the z slabs
//   k_voxel_scan / k_voxel_scatter   counting sort of the list by slab (iz is the most significant key field)
//   k_voxel_rank     output order = ascending key, like PCL's sorted linear voxel index: rank = start of the
//                    slab + number of smaller keys inside the slab (LDS-tiled compares), centroid written at
//                    that rank.  No host round trip anywhere: the entry count stays on the device.
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

__global__ __launch_bounds__(VOX_BLOCK) void k_voxel_insert(const float4 *__restrict__ pts, int n, float inv_leaf, float zmax,
                                                            VoxTable t)
{
    const int i = blockIdx.x * VOX_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && p.z >= 0.0f && p.z <= zmax)) return;     // PassThrough
    const unsigned long long key = vox_key(p.x, p.y, p.z, inv_leaf);
    unsigned int s = vox_hash(key) & (unsigned int)(t.cap - 1);
    for (;;) {
        const unsigned long long prev = atomicCAS(t.key + s, VOX_EMPTY, key);
        if (prev == VOX_EMPTY || prev == key) break;
        s = (s + 1) & (unsigned int)(t.cap - 1);
    }
    const unsigned int rgba = (unsigned int)__float_as_int(p.w);
    atomicAdd(reinterpret_cast<unsigned long long *>(t.sx + s), (unsigned long long)__double2ll_rn((double)p.x * 1048576.0));
    atomicAdd(reinterpret_cast<unsigned long long *>(t.sy + s), (unsigned long long)__double2ll_rn((double)p.y * 1048576.0));
    atomicAdd(reinterpret_cast<unsigned long long *>(t.sz + s), (unsigned long long)__double2ll_rn((double)p.z * 1048576.0));
    atomicAdd(t.c0 + s, rgba & 0xffu);
    atomicAdd(t.c1 + s, (rgba >> 8) & 0xffu);
    atomicAdd(t.c2 + s, (rgba >> 16) & 0xffu);
    atomicAdd(t.c3 + s, rgba >> 24);
    atomicAdd(t.n + s, 1u);
}

constexpr int VOX_BINS = 8192;        // z slabs of the ordering pass (iz clamped: the order stays monotone)
constexpr int VOX_SPT = 16;           // table slots per thread in k_voxel_compact

__device__ __forceinline__ int vox_bin(unsigned long long key)
{
    const long long iz = (long long)(key >> 42) - 1048576;       // >= 0 after PassThrough
    return (int)(iz < 0 ? 0 : (iz > VOX_BINS - 1 ? VOX_BINS - 1 : iz));
}

// counter[bin] += 1 for every active lane, returning the lane's ticket; lanes of the wave that hit the same bin
// share ONE atomic (a wall at constant depth puts 40 % of the voxels into one slab: per-lane atomics on that
// address serialise)
__device__ __forceinline__ int wave_ticket(int *__restrict__ counter, int bin, bool active)
{
    const int lane = threadIdx.x & 63;
    int ticket = 0;
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int b0 = __shfl(bin, leader);
        const unsigned long long grp = __ballot(active && bin == b0) & todo;
        int base = 0;
        if (lane == leader) base = atomicAdd(counter + b0, __popcll(grp));
        base = __shfl(base, leader);
        if ((grp >> lane) & 1ull) ticket = base + __popcll(grp & ((1ull << lane) - 1ull));
        todo &= ~grp;
    }
    return ticket;
}

// occupied slots -> dense (key, slot) list + histogram of the z slabs.  Each block scans 4096 slots, takes ONE
// ticket from the global counter and flushes ONE block-local histogram (same-address global atomics serialise:
// one per wave cost 166 us here).
__global__ __launch_bounds__(VOX_BLOCK) void k_voxel_compact(VoxTable t, unsigned long long *__restrict__ lkey, int *__restrict__ lslot,
                                                             int *__restrict__ m, int *__restrict__ hist)
{
    __shared__ int wsum[VOX_BLOCK / 64], base_sh;
    __shared__ int lh[VOX_BINS];                                 // block-local slab histogram (32 KB of LDS)
    for (int k = threadIdx.x; k < VOX_BINS; k += VOX_BLOCK) lh[k] = 0;
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
            atomicAdd(&lh[vox_bin(k[j])], 1);                    // LDS atomic; flushed once per block below
            ++pos;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < VOX_BINS; k += VOX_BLOCK)
        if (lh[k]) atomicAdd(hist + k, lh[k]);
}

// exclusive prefix of the slab histogram -> start[]; cursor[] reset.  one block of 1024
__global__ __launch_bounds__(1024) void k_voxel_scan(const int *__restrict__ hist, int *__restrict__ start, int *__restrict__ cursor)
{
    __shared__ int part[1024];
    constexpr int PER = VOX_BINS / 1024;
    int loc[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { loc[j] = hist[threadIdx.x * PER + j]; sum += loc[j]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        start[threadIdx.x * PER + j] = run;
        cursor[threadIdx.x * PER + j] = 0;
        run += loc[j];
    }
    if (threadIdx.x == 1023) start[VOX_BINS] = run;
}

// group the list by slab (order inside a slab is arbitrary; k_voxel_rank fixes it)
__global__ __launch_bounds__(VOX_BLOCK) void k_voxel_scatter(const unsigned long long *__restrict__ lkey, const int *__restrict__ lslot,
                                                             const int *__restrict__ m, const int *__restrict__ start,
                                                             int *__restrict__ cursor, unsigned long long *__restrict__ gkey,
                                                             int *__restrict__ gslot)
{
    const int e = blockIdx.x * VOX_BLOCK + threadIdx.x;
    const bool live = e < *m;
    const unsigned long long k = live ? lkey[e] : 0ull;
    const int b = live ? vox_bin(k) : 0;
    const int tk = wave_ticket(cursor, b, live);
    if (!live) return;
    const int pos = start[b] + tk;
    gkey[pos] = k;
    gslot[pos] = lslot[e];
}

// Output order = ascending key, like PCL's sorted linear voxel index.  Keys are grouped by slab and slabs are
// ordered, so an entry's rank = start of its slab + the keys of that slab below its own: the block stages the
// union of its entries' slabs through LDS and every thread compares inside its own slab only.  grid covers the
// worst case (n entries); blocks beyond *m leave at once.
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

}  // namespace s3d
