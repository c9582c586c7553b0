// voxel.hpp -- PassThrough + VoxelGrid of GraphicEnd::readimage on gfx950.
//
// Replaces pcl::PassThrough (z in [0, z_filter]) and pcl::VoxelGrid (cubic leaf grid_leaf = 0.03) of
// src/GraphicEnd.cpp:283-295 for the 16-byte {x, y, z, rgba} records of the reference's binary PCD files.
// oracle/voxel_oracle.c is the CPU twin.  The centroids come from integer fixed-point sums (2^-20 m) and
// integer colour sums, so the atomics below give the same bits whatever order they land in.
//
//   k_voxel_insert   one thread per point: 64-bit voxel key (iz,iy,ix); runs of equal keys summed in the wave, then in a block-local LDS
//                    table, then ONE update of the open-addressing table in HBM per (block, voxel) (two-line slots, see VoxSlot: atomicCAS
//                    on the key; the block that claims a slot stores its sums with plain writes, blocks that find it claimed add theirs
//                    atomically into the key's own line).  The lane that claims a slot lists it (per block, through an LDS counter) and
//                    sets the voxel's BIT in the occupancy bitmap of its (iz, iy) row -- ORed per block in LDS first.
//   k_voxel_scan<true>   row popcounts -> row starts (prefix inside blocks of 1,024 rows + block totals); occupied rows copied aside, bitmap cleared.
//   k_voxel_finalize     output order = ascending key, like PCL's sorted linear voxel index: rank = start of the voxel's row + occupied
//                    bits below its own; centroid written at that rank; the slot is reset as it is read; the count goes to host-mapped memory.
//   general path (a frame with a voxel outside the bitmap's key range -- flagged on the device, run by the host in the same call):
//   k_voxel_hist / k_voxel_scan<false> / k_voxel_scatter / k_voxel_rank: counting sort of the listed slots by row, rank by key compares.
// The tables are SELF-CLEANING: every call leaves every slot empty, the bitmap and the histogram zero (round 1 cleared 52 MB of
// table per call and scanned its 2^20 slots for the occupied ones: 0.4 % of the HBM roofline).  No host round trip until the final count.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace s3d {

constexpr unsigned long long VOX_EMPTY = ~0ull;
constexpr int VOX_BLOCK = 256;
constexpr int VOX_TILE = 2048;        // keys staged in LDS per step of the ranking kernel

// A voxel slot is TWO cache lines (round 4).  Line 0 is the atomic line: the key every lane CASes on, and the sums of
// the blocks that found the voxel already claimed ("late" sums; zero between calls).  Line 1 belongs to the ONE block whose
// CAS claimed the slot: it writes its sums there with plain 16-byte stores -- no atomics, never cleared (a claim always
// overwrites all of it before anything reads it).  Most voxels of an organized cloud lie inside one image tile, so most
// updates are one returning atomic + one line store instead of one returning atomic + six atomic adds; the sums are
// integers, so first + late is the same total whatever the order.
struct __attribute__((aligned(64))) VoxSums {
    long long sx, sy, sz;                          // fixed-point coordinate sums (2^-20 m)
    unsigned long long c01, c23;                   // colour channel sums, two per word (c0 | c1 << 32), (c2 | c3 << 32): each < 2^27
    unsigned int n;                                // point count
    unsigned int pad[5];
};
struct __attribute__((aligned(128))) VoxSlot {
    unsigned long long key;                        // line 0: key + late sums
    long long sx, sy, sz;
    unsigned long long c01, c23;
    unsigned int n;
    unsigned int pad[3];
    VoxSums first;                                 // line 1: the claiming block's sums
};
static_assert(sizeof(VoxSums) == 64 && sizeof(VoxSlot) == 128 && offsetof(VoxSlot, first) == 64, "a voxel slot is two 64-byte lines");
struct VoxTable { VoxSlot *slot; int cap; };       // `cap` slots (power of two) PER FRAME: frame f of a batch owns slot[f * cap, (f + 1) * cap)
// Round 5: one launch sequence serves a BATCH of frames (blockIdx.y = frame; saveOutput merges every keyframe, src/saveOutput.cpp:58-96,
// and loop closure touches 30 at a time): every table below exists once per frame, `*_stride` apart.  A single frame is the batch of
// one whose record travels as a kernel argument (no pointer table to upload first).
struct VoxFrame { const float4 *pts; float4 *out; int n, pad; };
struct VoxLayout {
    VoxTable t;
    unsigned long long *lkey; int *lslot; int *bcount; int blk_stride;      // claim lists: blk_stride insert blocks per frame (bcount: blk_stride + 1 ints)
    int *hist; int hist_stride;                                            // hist | start | cursor | btot | boff | ticket, VOX_HIST_INTS per frame
    unsigned long long *gkey; int *gslot; int g_stride;                    // sorted lists: g_stride entries per frame
    int *m_host;                                                           // host-mapped voxel counts, one per frame (-2: the frame needs the general ordering path)
    unsigned long long *bits, *rowbits;                                    // round 6: occupancy bitmaps of the dense key range, VOX_BINS x VOX_BW words per frame (see k_voxel_finalize)
    int *flags;                                                            // per frame: 1 = a voxel outside the dense key range was claimed
};
__device__ __forceinline__ VoxFrame vox_frame(const VoxFrame *__restrict__ frames, const VoxFrame &f0) { return frames ? frames[blockIdx.y] : f0; }

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

// every slot empty, every sum zero: at allocation and after a call that failed half way
__global__ __launch_bounds__(VOX_BLOCK) void k_voxel_clear(VoxTable t)
{
    const int s = blockIdx.x * VOX_BLOCK + threadIdx.x;
    if (s >= t.cap) return;
    uint4 *q = reinterpret_cast<uint4 *>(t.slot + (size_t)blockIdx.y * t.cap + s);                      // (line 0 only: line 1 needs no initial state)
    q[0] = make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u);
    q[1] = make_uint4(0u, 0u, 0u, 0u); q[2] = make_uint4(0u, 0u, 0u, 0u); q[3] = make_uint4(0u, 0u, 0u, 0u);
}

// inclusive sum over the lanes of the same run (runs = maximal stretches of consecutive lanes with equal keys,
// numbered by `seg`): Hillis-Steele with a segment test; integer adds, so exact
template <int CTRL> __device__ __forceinline__ int vox_dpp(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ long long vox_dpp(long long x)
{
    const int lo = vox_dpp<CTRL>((int)(unsigned int)((unsigned long long)x & 0xffffffffull)), hi = vox_dpp<CTRL>((int)((unsigned long long)x >> 32));
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
}
template <int CTRL> __device__ __forceinline__ unsigned long long vox_dpp(unsigned long long x) { return (unsigned long long)vox_dpp<CTRL>((long long)x); }
__device__ __forceinline__ int vox_rdlane(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ long long vox_rdlane(long long x, int l)
{
    const int lo = __builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)x & 0xffffffffull), l);
    const int hi = __builtin_amdgcn_readlane((int)((unsigned long long)x >> 32), l);
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ unsigned long long vox_rdlane(unsigned long long x, int l) { return (unsigned long long)vox_rdlane((long long)x, l); }
// No LDS traffic: a segmented Hillis-Steele scan inside each 16-lane row with DPP row_shr moves (lanes shifted in from
// outside the row are excluded by the lane test), then the rows are chained in order -- a run that continues from the
// previous row adds that row's last, already final, value (v_readlane broadcasts).
template <typename T> __device__ __forceinline__ T run_scan(T v, int seg)
{
    const int lane = threadIdx.x & 63, rl = lane & 15;
    { const T u = vox_dpp<0x111>(v); const int su = vox_dpp<0x111>(seg); if (rl >= 1 && su == seg) v += u; }
    { const T u = vox_dpp<0x112>(v); const int su = vox_dpp<0x112>(seg); if (rl >= 2 && su == seg) v += u; }
    { const T u = vox_dpp<0x114>(v); const int su = vox_dpp<0x114>(seg); if (rl >= 4 && su == seg) v += u; }
    { const T u = vox_dpp<0x118>(v); const int su = vox_dpp<0x118>(seg); if (rl >= 8 && su == seg) v += u; }
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        const T carry = vox_rdlane(v, 16 * r - 1);
        const int cseg = vox_rdlane(seg, 16 * r - 1);
        if ((lane >> 4) == r && seg == cseg) v += carry;
    }
    return v;
}

constexpr int VOX_BZ = 512, VOX_BY = 256;   // ordering bins: (iz, iy) rows, both clamped (the order stays monotone)
constexpr int zbias = 128;                   // iz in [-128, 383] has its own slab (-3.8 m .. 11.5 m at a 3 cm leaf)
constexpr int VOX_BINS = VOX_BZ * VOX_BY;
constexpr int VOX_SCAN_BLOCKS = VOX_BINS / 1024;
constexpr int VOX_HIST_INTS = 3 * VOX_BINS + 16 + 2 * VOX_SCAN_BLOCKS;      // per frame: hist | start (+8) | cursor | btot | boff | ticket (+8)

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

// Round 6: the ORDER without a sort.  Inside the dense key range -- iz and iy inside the row table above, ix in [-256, 255] (7.7 m either
// side at a 3 cm leaf: every cloud in a camera frame) -- a voxel is one BIT: row (iz, iy), bit ix of the row's VOX_BW 64-bit words.  The
// lane that claims a voxel sets its bit (ORed per block in LDS first: one memory-side atomic per (block, word), 11.8 k instead of the
// 35 k histogram increments of a 640x480 frame); k_voxel_scan<true> turns the rows' popcounts into row starts, copies the occupied rows
// to `rowbits` and clears the bitmap; k_voxel_finalize ranks every claimed voxel by start[row] + popcount(bits below its own) -- ascending
// key, PCL's order -- and writes its centroid.  Three launches, no scatter, no key compares.  A claimed voxel outside the range raises the
// frame's flag: the host then runs the general path (k_voxel_hist from the claim lists, scan, scatter, rank) for that frame.
constexpr int VOX_BW = 8;                    // 64-bit words per row: 512 ix values
__device__ __forceinline__ bool vox_dense(unsigned long long key, int &row, int &ixr)
{
    const int iz = (int)(key >> 42) - 1048576 + (int)zbias;
    const int iy = (int)((key >> 21) & 0x1FFFFF) - 1048576 + VOX_BY / 2;
    ixr = (int)(key & 0x1FFFFF) - 1048576 + 32 * VOX_BW;
    row = iz * VOX_BY + iy;
    return iz >= 0 && iz < VOX_BZ && iy >= 0 && iy < VOX_BY && ixr >= 0 && ixr < 64 * VOX_BW;
}

// One thread per point, one block per 16x16-pixel tile of an organized cloud (ORG; a voxel of the 3 cm grid covers
// ~6x6 pixels at 2.5 m, so a tile holds about a dozen voxels) or per 256 consecutive records otherwise.
//   1. runs of equal keys inside a wave are summed with a segmented scan (integers: same bits);
//   2. the run tails add into a block-local hash table in LDS (512 entries; LDS atomics);
//   3. every occupied LDS entry becomes ONE update of the global table: atomicCAS on the key, then either three plain
//      16-byte stores (the slot was empty: this block's sums are the voxel's first) or six atomic adds (it was not).
//      Device-scope atomics are the scarce resource here (~1-2 returning ones per ns scattered; 29 of the insert's 36 us
//      when every update made seven): the two aggregation levels cut the updates from one per point run (77 k per
//      640x480 frame) to one per (tile, voxel) (~15 k), the first-writer line cuts the atomics per update.
// The lane whose CAS claims an empty global slot lists it at lkey / lslot[block * VOX_BLOCK + k] (k from an LDS
// counter; bcount[block] = entries listed) and bumps the row histogram.
#ifdef VOX_DBG       // developer build: thread 0 of every insert block books the 100 MHz clock at its phase boundaries (tools/vox_phases.py)
__device__ long long g_vox_dbg[4096 * 8];
#define VOXT(k) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) g_vox_dbg[blockIdx.x * 8 + (k)] = (long long)wall_clock64(); } while (0)
#else
#define VOXT(k) do { } while (0)
#endif
constexpr int VOX_LH = 512;            // LDS hash entries per block (256 points: load <= 0.5)
constexpr int VOX_TW = 16;             // tile edge in pixels (ORG)
constexpr int VOX_LIST_BLOCK = 1024;   // threads of a LIST insert block: a run is 1,024 consecutive records, summed in one pass
constexpr int VOX_MAX_PASSES = 4;      // runs a list block may take
constexpr int VOX_SEG_ALIGN = VOX_MAX_PASSES * VOX_LIST_BLOCK / 256;      // the claim lists are allocated in multiples of a list block's segments

// Point LISTS (round 6): a block of 1,024 threads takes `passes` runs of 1,024 consecutive records into ONE LDS table before it touches the
// global one.  The reference's PCD frames are raster-ordered lists without the invalid pixels: 256 records (round 5's block) are half an image
// row, a 3 cm voxel at 2 m spans six rows, so every voxel was claimed once and then updated by five other blocks with six atomics each (302 k
// memory-side atomics per frame; an organized frame cut into 16x16 tiles needs 66 k).  4,096 records are nine rows: 58 k.  The table (2,048
// entries: a run cannot fill more than half) is flushed early whenever the next run could fill it, so any input is handled; the block's
// claim list is its passes x 4 segments of 256 entries, filled in order.
template <bool ORG>
__global__ __launch_bounds__(ORG ? VOX_BLOCK : VOX_LIST_BLOCK) void k_voxel_insert(VoxFrame f0, const VoxFrame *__restrict__ frames, int W, int H, float inv_leaf, float zmin,
                                                            float zmax, VoxLayout L, int passes /* runs of 256 records per block (lists; 1 for tiles) */)
{
    constexpr int BS = ORG ? VOX_BLOCK : VOX_LIST_BLOCK;       // threads = records per run
    constexpr int LH = 2 * BS;                                 // LDS hash entries: a run cannot fill more than half
    constexpr int SEGS = BS / VOX_BLOCK;                       // claim-list segments (of VOX_BLOCK entries) a run may fill
    const VoxFrame fr = vox_frame(frames, f0);
    const float4 *__restrict__ pts = fr.pts;
    const int n = fr.n, fb = blockIdx.y;
    VoxTable t = L.t; t.slot += (size_t)fb * t.cap;
    unsigned long long *__restrict__ lkey = L.lkey + ((size_t)fb * L.blk_stride + (size_t)blockIdx.x * passes * SEGS) * VOX_BLOCK;
    int *__restrict__ lslot = L.lslot + ((size_t)fb * L.blk_stride + (size_t)blockIdx.x * passes * SEGS) * VOX_BLOCK;
    int *__restrict__ bcount = L.bcount + (size_t)fb * (L.blk_stride + 1) + 1 + (size_t)blockIdx.x * passes * SEGS;
    unsigned long long *__restrict__ bits = L.bits + (size_t)fb * VOX_BINS * VOX_BW;
    // colour sums: a tile block holds 256 points, so its four channel sums fit 16-bit fields of ONE word (hc01; 255 * 256 < 2^16); a list block
    // sums up to 4,096 points per voxel: two words of two 32-bit fields, as in the global table
    __shared__ unsigned long long hk[LH], hc01[LH], hc23[ORG ? 1 : LH];
    __shared__ long long hsx[LH], hsy[LH], hsz[LH];
    __shared__ unsigned int hn[LH];
    __shared__ int occ[LH];                                      // the occupied entries, compacted; then the bitmap word of each claimed one
    __shared__ unsigned char obit[LH];                           // ... and its bit
    __shared__ int bcnt, nocc, nkeys;
    VOXT(0);
    for (int k = threadIdx.x; k < LH; k += BS) { hk[k] = VOX_EMPTY; hc01[k] = 0; if constexpr (!ORG) hc23[k] = 0; hsx[k] = 0; hsy[k] = 0; hsz[k] = 0; hn[k] = 0; }
    if (threadIdx.x == 0) { bcnt = 0; nocc = 0; nkeys = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    auto record_of = [&](int pass) __attribute__((always_inline)) {
        int i = -1;
        if constexpr (ORG) {
            const int tiles_x = (W + VOX_TW - 1) / VOX_TW;
            const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
            const int u = tx * VOX_TW + (threadIdx.x & (VOX_TW - 1)), v = ty * VOX_TW + (threadIdx.x / VOX_TW);
            if (u < W && v < H) i = v * W + u;
        } else {
            i = (blockIdx.x * passes + pass) * BS + threadIdx.x;
            if (i >= n) i = -1;
        }
        return i;
    };
    // level 3: one global update per (block, voxel) in the table, then the table is empty again.  The occupied entries are compacted first, so
    // that each has a thread of its own and the block pays one round of returning-atomic latency per 256 of them.
    auto flush = [&](bool last) __attribute__((always_inline)) {
        for (int k = threadIdx.x; k < LH; k += BS)
            if (hk[k] != VOX_EMPTY) occ[atomicAdd(&nocc, 1)] = k;
        __syncthreads();
        const int n_occ = nocc;
        VOXT(4);
        for (int e = threadIdx.x; e < n_occ; e += BS) {
            const int k = occ[e];
            const unsigned long long gk = hk[k];
            unsigned int s = vox_hash(gk) & (unsigned int)(t.cap - 1);
            bool claimed = false;
            for (;;) {
                const unsigned long long was = atomicCAS(&t.slot[s].key, VOX_EMPTY, gk);
                if (was == VOX_EMPTY) { claimed = true; break; }
                if (was == gk) break;
                s = (s + 1) & (unsigned int)(t.cap - 1);
            }
            VoxSlot *q = t.slot + s;
            VOXT(5);
            int wi = -1, wb = 0;                                     // the bitmap word / bit of the voxel this entry claimed (-1: none)
            unsigned long long c01, c23;                             // the global table's form: (c0 | c1 << 32), (c2 | c3 << 32)
            if constexpr (ORG) { const unsigned long long c = hc01[k]; c01 = (c & 0xffffull) | (((c >> 16) & 0xffffull) << 32); c23 = ((c >> 32) & 0xffffull) | ((c >> 48) << 32); }
            else { c01 = hc01[k]; c23 = hc23[k]; }
            if (claimed) {                                               // this block owns line 1: three plain 16-byte stores
                const unsigned long long sx = (unsigned long long)hsx[k], sy = (unsigned long long)hsy[k], sz = (unsigned long long)hsz[k];
                uint4 *f = reinterpret_cast<uint4 *>(&q->first);
                f[0] = make_uint4((unsigned int)sx, (unsigned int)(sx >> 32), (unsigned int)sy, (unsigned int)(sy >> 32));
                f[1] = make_uint4((unsigned int)sz, (unsigned int)(sz >> 32), (unsigned int)c01, (unsigned int)(c01 >> 32));
                f[2] = make_uint4((unsigned int)c23, (unsigned int)(c23 >> 32), hn[k], 0u);
                const int c = atomicAdd(&bcnt, 1);                       // LDS: at most one claim per record of the block
                lkey[c] = gk;
                lslot[c] = (int)s;
                int row, ixr;
                if (vox_dense(gk, row, ixr)) { wi = row * VOX_BW + (ixr >> 6); wb = ixr & 63; }
                else L.flags[fb] = 1;                                    // (same value from every writer)
            } else {                                                     // another block claimed the voxel: late sums, line 0
                atomicAdd(reinterpret_cast<unsigned long long *>(&q->sx), (unsigned long long)hsx[k]);
                atomicAdd(reinterpret_cast<unsigned long long *>(&q->sy), (unsigned long long)hsy[k]);
                atomicAdd(reinterpret_cast<unsigned long long *>(&q->sz), (unsigned long long)hsz[k]);
                atomicAdd(&q->c01, c01);
                atomicAdd(&q->c23, c23);
                atomicAdd(&q->n, hn[k]);
            }
            occ[e] = wi; obit[e] = (unsigned char)wb;                    // (entry e is this thread's alone)
        }
        // the claimed voxels' bits, ORed per bitmap word in LDS (the hash arrays are free by now: hk = word index, hsx = bits)
        __syncthreads();
        for (int k = threadIdx.x; k < LH; k += BS) { hk[k] = VOX_EMPTY; hsx[k] = 0; }
        __syncthreads();
        for (int e = threadIdx.x; e < n_occ; e += BS) {
            const int wi = occ[e];
            if (wi < 0) continue;
            unsigned int s = vox_hash((unsigned long long)wi) & (LH - 1);
            for (;;) {
                const unsigned long long was = atomicCAS(&hk[s], VOX_EMPTY, (unsigned long long)wi);
                if (was == VOX_EMPTY || was == (unsigned long long)wi) break;
                s = (s + 1) & (LH - 1);
            }
            atomicOr(reinterpret_cast<unsigned long long *>(&hsx[s]), 1ull << obit[e]);
        }
        __syncthreads();
        for (int k = threadIdx.x; k < LH; k += BS) {
            if (hk[k] != VOX_EMPTY) atomicOr(bits + hk[k], (unsigned long long)hsx[k]);
            if (!last) { hk[k] = VOX_EMPTY; hc01[k] = 0; if constexpr (!ORG) hc23[k] = 0; hsx[k] = 0; hsy[k] = 0; hsz[k] = 0; hn[k] = 0; }      // an empty table for the runs that follow
        }
        if (last) return;
        if (threadIdx.x == 0) { nocc = 0; nkeys = 0; }
        __syncthreads();
    };
    int i = record_of(0);
    float4 pn = make_float4(0.0f, 0.0f, -1.0f, 0.0f);
    if (i >= 0) pn = pts[i];
    for (int pass = 0; pass < passes; ++pass) {
        const float4 p = pn;
        const bool have = i >= 0;
        if (pass + 1 < passes) {                                     // the next run's record is in flight while this one is summed
            i = record_of(pass + 1);
            pn = make_float4(0.0f, 0.0f, -1.0f, 0.0f);
            if (i >= 0) pn = pts[i];
        }
#ifdef VOX_DBG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        VOXT(1);
        const bool ok = have && isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && p.z >= zmin && p.z <= zmax;     // PassThrough
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
        // the four colour channels in 16-bit fields of one word (a run is at most 64 lanes: 64 * 255 < 2^16): one scan instead of two; the run's
        // length needs none -- its lanes are consecutive, the tail knows where its head is
        const unsigned long long c4 = run_scan((unsigned long long)(rgba & 0xffu) | ((unsigned long long)((rgba >> 8) & 0xffu) << 16) |
                                               ((unsigned long long)((rgba >> 16) & 0xffu) << 32) | ((unsigned long long)(rgba >> 24) << 48), seg);
        const int cnt = lane - (63 - __builtin_clzll(heads & ((2ull << lane) - 1ull))) + 1;
        VOXT(2);
        if (ok && tail) {                                            // level 2: the block's LDS table
            unsigned int s = vox_hash(key) & (LH - 1);
            for (;;) {
                const unsigned long long was = atomicCAS(&hk[s], VOX_EMPTY, key);
                if (was == VOX_EMPTY) { if constexpr (!ORG) atomicAdd(&nkeys, 1); break; }
                if (was == key) break;
                s = (s + 1) & (LH - 1);
            }
            atomicAdd(reinterpret_cast<unsigned long long *>(&hsx[s]), (unsigned long long)qx);
            atomicAdd(reinterpret_cast<unsigned long long *>(&hsy[s]), (unsigned long long)qy);
            atomicAdd(reinterpret_cast<unsigned long long *>(&hsz[s]), (unsigned long long)qz);
            if constexpr (ORG) atomicAdd(&hc01[s], c4);
            else {
                atomicAdd(&hc01[s], (c4 & 0xffffull) | (((c4 >> 16) & 0xffffull) << 32));
                atomicAdd(&hc23[s], ((c4 >> 32) & 0xffffull) | ((c4 >> 48) << 32));
            }
            atomicAdd(&hn[s], (unsigned int)cnt);
        }
        __syncthreads();
        VOXT(3);
        // the next run may bring 256 new keys: the table must not fill up (load <= 3/4), so it is emptied early when more than half is taken
        if (!ORG && pass + 1 < passes && nkeys > LH / 2) flush(false);
    }
    flush(true);                                                    // (bcnt is final behind the barriers inside)
    VOXT(6);
#ifdef VOX_DBG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    VOXT(7);
    if ((int)threadIdx.x < passes * SEGS) bcount[threadIdx.x] = min(VOX_BLOCK, max(0, bcnt - (int)threadIdx.x * VOX_BLOCK));      // the block's segments, filled in order
}

// general ordering path only: the row histogram of a flagged frame, from its claim lists.  grid (insert blocks, frames)
__global__ __launch_bounds__(VOX_BLOCK) void k_voxel_hist(VoxLayout L)
{
    const int fb = blockIdx.y;
    if (!L.flags[fb]) return;
    const int nb = L.bcount[(size_t)fb * (L.blk_stride + 1) + 1 + blockIdx.x];
    if ((int)threadIdx.x < nb) atomicAdd(L.hist + (size_t)fb * L.hist_stride + vox_bin(L.lkey[((size_t)fb * L.blk_stride + blockIdx.x) * VOX_BLOCK + threadIdx.x]), 1);
}

// exclusive prefix of the row histogram; cursor[] and the histogram itself reset for the next call.  ONE launch of
// VOX_BINS / 1024 blocks: every block scans its 1024 rows (wave shuffles + one LDS step) into start[] (prefix inside the
// block) and leaves its total; the block that arrives last at the ticket (no spinning: it is simply the last one) turns
// the 128 totals into the blocks' exclusive offsets boff[] and the voxel count M = start[VOX_BINS], which it also writes
// straight into host-mapped memory (no copy launch behind the pipeline).  Consumers read vox_start(): start[b] + boff[b / 1024].
struct VoxHist { int *hist, *start, *cursor, *btot, *boff, *ticket; };
__device__ __forceinline__ VoxHist vox_hist_of(const VoxLayout &L, int fb)
{
    VoxHist h;
    h.hist = L.hist + (size_t)fb * L.hist_stride; h.start = h.hist + VOX_BINS; h.cursor = h.start + VOX_BINS + 8; h.btot = h.cursor + VOX_BINS;
    h.boff = h.btot + VOX_SCAN_BLOCKS; h.ticket = h.boff + VOX_SCAN_BLOCKS;
    return h;
}
// grid (VOX_SCAN_BLOCKS, frames).  <BITS>: a row's count is the popcount of its bitmap words; occupied rows are copied to `rowbits` (what
// k_voxel_finalize reads) and cleared in place -- the bitmap is empty again when the launch ends; a flagged frame reports -2 instead of its
// count (the host then runs the general path, whose scan is the <false> instance, on it).  <false> only works on flagged frames.
// <true> runs on the scan blocks [sb0, sb0 + gridDim.x) only: PassThrough keeps z inside [zmin, zmax], so the host knows which slabs of the row
// table can hold a bit at all (z in [0, 7 m] at a 3 cm leaf: 59 of the 128 blocks) -- the rest of the bitmap is neither read nor counted.
template <bool BITS>
__global__ __launch_bounds__(1024) void k_voxel_scan(VoxLayout L, int sb0)
{
    if (!BITS && !L.flags[blockIdx.y]) return;
    const VoxHist vh = vox_hist_of(L, blockIdx.y);
    int *__restrict__ hist = vh.hist, *__restrict__ start = vh.start, *__restrict__ cursor = vh.cursor, *__restrict__ btot = vh.btot,
        *__restrict__ boff = vh.boff, *__restrict__ ticket = vh.ticket, *__restrict__ m_host = L.m_host + blockIdx.y;
    __shared__ int wtot[16];
    __shared__ int last_sh;
    const int sblk = (int)blockIdx.x + sb0;
    const int i = sblk * 1024 + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int v;
    if constexpr (BITS) {
        uint4 *__restrict__ rb = reinterpret_cast<uint4 *>(L.bits + ((size_t)blockIdx.y * VOX_BINS + i) * VOX_BW);
        uint4 r[VOX_BW / 2];
        v = 0;
#pragma unroll
        for (int k = 0; k < VOX_BW / 2; ++k) { r[k] = rb[k]; v += __popc(r[k].x) + __popc(r[k].y) + __popc(r[k].z) + __popc(r[k].w); }
        if (v) {
            uint4 *__restrict__ cp = reinterpret_cast<uint4 *>(L.rowbits + ((size_t)blockIdx.y * VOX_BINS + i) * VOX_BW);
#pragma unroll
            for (int k = 0; k < VOX_BW / 2; ++k) { cp[k] = r[k]; rb[k] = make_uint4(0u, 0u, 0u, 0u); }
        }
    } else {
        v = hist[i];
        if (v) hist[i] = 0;
    }
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    int before = 0;
    for (int q = 0; q < w; ++q) before += wtot[q];
    start[i] = before + incl - v;                      // exclusive prefix inside the block
    if constexpr (!BITS) cursor[i] = 0;
    if constexpr (BITS) {       // the dense path stops here: k_voxel_finalize sums the block totals itself (no ticket, no last-block pass on the critical path)
        if (threadIdx.x == 1023) btot[sblk] = before + incl;
        return;
    }
    if (threadIdx.x == 1023) {
        // (round 6: no __threadfence() on either side of the ticket -- on gfx950 it is buffer_wbl2 + buffer_inv, an L2 write-back and
        //  invalidate.  The block total is a device-scope store and the last block reads the totals with device-scope loads: the store
        //  only has to have been PERFORMED before the ticket is taken, which is s_waitcnt vmcnt(0).)
        __hip_atomic_store(btot + blockIdx.x, before + incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last_sh = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == VOX_SCAN_BLOCKS - 1;
    }
    __syncthreads();
    if (!last_sh || threadIdx.x >= 64) return;
    // the last block: exclusive scan of the VOX_SCAN_BLOCKS totals by one wave (two values per lane at 128 blocks)
    constexpr int PER = (VOX_SCAN_BLOCKS + 63) / 64;
    int tv[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int q = lane * PER + k;
        tv[k] = q < VOX_SCAN_BLOCKS ? __hip_atomic_load(btot + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        sum += tv[k];
    }
    int inc = sum;
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    int run = inc - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int q = lane * PER + k;
        if (q < VOX_SCAN_BLOCKS) boff[q] = run;
        run += tv[k];
    }
    if (lane == 63) {
        start[VOX_BINS] = inc;                         // = number of voxels
        *ticket = 0;                                   // ready for the next call
        __hip_atomic_store(m_host, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // host-mapped: the host polls it while the later launches run
    }
}
__device__ __forceinline__ int vox_start(const int *__restrict__ start, const int *__restrict__ boff, int b)
{
    return b >= VOX_BINS ? start[VOX_BINS] : start[b] + boff[b >> 10];
}

// group the listed slots by row (order inside a row is arbitrary; k_voxel_rank fixes it).  Thread e looks at entry
// e % VOX_BLOCK of insert block e / VOX_BLOCK.  The claims of one insert block come from one image tile, i.e. from a
// handful of rows: lanes with the same row share ONE returning atomic on its cursor.
__global__ __launch_bounds__(VOX_BLOCK) void k_voxel_scatter(VoxLayout L)
{
    const int fb = blockIdx.y;
    if (!L.flags[fb]) return;                                          // general ordering path: flagged frames only
    const VoxHist vh = vox_hist_of(L, fb);
    const unsigned long long *__restrict__ lkey = L.lkey + (size_t)fb * L.blk_stride * VOX_BLOCK;
    const int *__restrict__ lslot = L.lslot + (size_t)fb * L.blk_stride * VOX_BLOCK;
    const int *__restrict__ bcount = L.bcount + (size_t)fb * (L.blk_stride + 1) + 1;
    const int *__restrict__ start = vh.start, *__restrict__ boff = vh.boff;
    int *__restrict__ cursor = vh.cursor;
    unsigned long long *__restrict__ gkey = L.gkey + (size_t)fb * L.g_stride;
    int *__restrict__ gslot = L.gslot + (size_t)fb * L.g_stride;
    const int nb = bcount[blockIdx.x];
    if ((int)(threadIdx.x & ~63u) >= nb) return;                       // whole wave beyond the list
    const bool live = (int)threadIdx.x < nb;
    const int lane = threadIdx.x & 63;
    const size_t e = (size_t)blockIdx.x * VOX_BLOCK + threadIdx.x;
    const unsigned long long k = live ? lkey[e] : 0ull;
    const int b = live ? vox_bin(k) : -1;
    // group the lanes by row WITHOUT touching memory (leader lane, rank in the group, group size), then all leaders issue
    // their returning atomics in one instruction: one round trip per wave instead of one per distinct row
    int leader = lane, rank = 0, size = 1;
    unsigned long long rest = __ballot(live);
    while (rest) {
        const int l0 = __builtin_ctzll(rest);
        const int b0 = __shfl(b, l0);
        const unsigned long long same = __ballot(b == b0) & rest;
        if (b == b0) { leader = l0; rank = __popcll(same & ((1ull << lane) - 1ull)); size = __popcll(same); }
        rest &= ~same;
    }
    int base = 0;
    if (live && leader == lane) base = atomicAdd(cursor + b, size);
    base = __shfl(base, leader);
    const int pos = live ? vox_start(start, boff, b) + base + rank : 0;
    if (!live) return;
    gkey[pos] = k;
    gslot[pos] = lslot[e];
}

// Output order = ascending key, like PCL's sorted linear voxel index.  Keys are grouped by (iz, iy) row and rows
// are ordered, so an entry's rank = start of the first row its block touches + the keys of the block's rows below
// its own: the block stages the union of its entries' rows through LDS.  grid covers the worst case (n entries);
// blocks beyond the voxel count leave at once.  Every slot is reset as it is read (self-cleaning table).
// centroid of slot q (first + late sums) -> out[rank]; the slot is left empty for the next call
__device__ __forceinline__ void vox_emit(VoxSlot *q, float4 *__restrict__ out, int rank)
{
    const uint4 a0 = reinterpret_cast<const uint4 *>(q)[0], a1 = reinterpret_cast<const uint4 *>(q)[1],
                a2 = reinterpret_cast<const uint4 *>(q)[2], a3 = reinterpret_cast<const uint4 *>(q)[3];        // key + late sums
    const uint4 f0 = reinterpret_cast<const uint4 *>(&q->first)[0], f1 = reinterpret_cast<const uint4 *>(&q->first)[1],
                f2 = reinterpret_cast<const uint4 *>(&q->first)[2];                                             // the claiming block's sums
    const long long sx = (long long)((((unsigned long long)a0.w << 32) | a0.z) + (((unsigned long long)f0.y << 32) | f0.x));
    const long long sy = (long long)((((unsigned long long)a1.y << 32) | a1.x) + (((unsigned long long)f0.w << 32) | f0.z));
    const long long sz = (long long)((((unsigned long long)a1.w << 32) | a1.z) + (((unsigned long long)f1.y << 32) | f1.x));
    const unsigned int c0 = a2.x + f1.z, c1 = a2.y + f1.w, c2 = a2.z + f2.x, c3 = a2.w + f2.y;
    const unsigned int ni = a3.x + f2.z;
    const double cnt = (double)ni;
    float4 o;
    o.x = (float)(((double)sx / cnt) / 1048576.0);
    o.y = (float)(((double)sy / cnt) / 1048576.0);
    o.z = (float)(((double)sz / cnt) / 1048576.0);
    const unsigned int rgba = (c0 / ni) | ((c1 / ni) << 8) | ((c2 / ni) << 16) | ((c3 / ni) << 24);
    o.w = __int_as_float((int)rgba);
    out[rank] = o;
    uint4 *w = reinterpret_cast<uint4 *>(q);                              // leave the slot empty for the next call
    w[0] = make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u);
    w[1] = make_uint4(0u, 0u, 0u, 0u); w[2] = make_uint4(0u, 0u, 0u, 0u); w[3] = make_uint4(0u, 0u, 0u, 0u);
}

// The dense path's last launch: one wave per insert block walks that block's claim list.  rank = start of the voxel's row + occupied bits
// below its own in the row's copy of the bitmap.  grid (list segments, frames), block 256: one thread per entry of the segment (a tile's
// segment holds ~30 claims: waves 1-3 leave at once; a list block's first segments are full).
// Workgroups go to the XCDs round robin, and a list block fills its first segments only: taken in list order, the busy segments of blocks
// with 16 segments each would all sit on XCDs 0 and 1 (measured: 3x the launch time).  So workgroup g takes segment g / nins of insert block
// g % nins: every block's first segment, then every block's second, ...
__global__ __launch_bounds__(VOX_BLOCK) void k_voxel_finalize(VoxFrame fr0, const VoxFrame *__restrict__ frames, VoxLayout L, int nins, int passes, int sb0, int sb1 /* the scan blocks that ran */)
{
    const int fb = blockIdx.y;
    const int seg = ((int)blockIdx.x % nins) * passes + (int)blockIdx.x / nins;
    const unsigned long long *__restrict__ lkey = L.lkey + ((size_t)fb * L.blk_stride + seg) * VOX_BLOCK;
    const int *__restrict__ lslot = L.lslot + ((size_t)fb * L.blk_stride + seg) * VOX_BLOCK;
    // the first 64 list entries are fetched together with the flag and the count (entries beyond the count are stale, never used):
    // one round trip less in front of the gathers below
    unsigned long long key0 = 0ull;
    int slot0 = 0;
    if (threadIdx.x < 64) { key0 = lkey[threadIdx.x]; slot0 = lslot[threadIdx.x]; }
    // every wave turns the scan blocks' totals into their exclusive offsets by itself: lane l holds blocks 2l and 2l + 1
    static_assert(VOX_SCAN_BLOCKS == 128, "two scan blocks per lane");
    const VoxHist vh = vox_hist_of(L, fb);
    const int lane = threadIdx.x & 63;
    const int t0 = (2 * lane >= sb0 && 2 * lane < sb1) ? vh.btot[2 * lane] : 0, t1 = (2 * lane + 1 >= sb0 && 2 * lane + 1 < sb1) ? vh.btot[2 * lane + 1] : 0;
    const int flagged = L.flags[fb];
    const int nb = L.bcount[(size_t)fb * (L.blk_stride + 1) + 1 + seg];
    int inc = t0 + t1;
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    const int excl = inc - (t0 + t1);
    if (blockIdx.x == 0 && threadIdx.x == 63)      // the frame's voxel count, straight into host-mapped memory (-2: the general path will order this frame)
        __hip_atomic_store(L.m_host + fb, flagged ? -2 : inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (flagged) return;
    if ((int)(threadIdx.x & ~63u) >= nb) return;
    const int *__restrict__ start = vh.start;
    VoxSlot *const slots = L.t.slot + (size_t)fb * L.t.cap;
    float4 *__restrict__ out = vox_frame(frames, fr0).out;
    const int e = threadIdx.x;
    const bool live = e < nb;
    const unsigned long long key = e < 64 ? key0 : (live ? lkey[e] : 0ull);
    int row, ixr;
    (void)vox_dense(key, row, ixr);
    if (!live) { row = 0; ixr = 0; }                                   // (stale list entries: any valid row)
    const int sb = row >> 10;                                          // the row's scan block; the shuffles run with the whole wave active
    const int off_even = __shfl(excl, sb >> 1), first_of_pair = __shfl(t0, sb >> 1);      // (both unconditional: a lane masked off would hand 0 to its readers)
    const int row_off = off_even + ((sb & 1) ? first_of_pair : 0);
    if (live) {
        VoxSlot *q = slots + (e < 64 ? slot0 : lslot[e]);
        const ulonglong2 *__restrict__ rb = reinterpret_cast<const ulonglong2 *>(L.rowbits + ((size_t)fb * VOX_BINS + row) * VOX_BW);
        int rank = start[row] + row_off;
        const int wi = ixr >> 6;
        const unsigned long long below = (1ull << (ixr & 63)) - 1ull;
#pragma unroll
        for (int k = 0; k < VOX_BW / 2; ++k) {
            const ulonglong2 w2 = rb[k];
            rank += 2 * k < wi ? __popcll(w2.x) : (2 * k == wi ? __popcll(w2.x & below) : 0);
            rank += 2 * k + 1 < wi ? __popcll(w2.y) : (2 * k + 1 == wi ? __popcll(w2.y & below) : 0);
        }
        vox_emit(q, out, rank);
    }
}

__global__ __launch_bounds__(VOX_BLOCK) void k_voxel_rank(VoxFrame fr0, const VoxFrame *__restrict__ frames, VoxLayout L)
{
    const int fb = blockIdx.y;
    if (!L.flags[fb]) return;                                          // general ordering path: flagged frames only
    const VoxHist vh = vox_hist_of(L, fb);
    VoxTable t = L.t; t.slot += (size_t)fb * t.cap;
    const unsigned long long *__restrict__ gkey = L.gkey + (size_t)fb * L.g_stride;
    const int *__restrict__ gslot = L.gslot + (size_t)fb * L.g_stride;
    const int *__restrict__ start = vh.start, *__restrict__ boff = vh.boff;
    float4 *__restrict__ out = vox_frame(frames, fr0).out;
    __shared__ unsigned long long tile[VOX_TILE];
    __shared__ int lo_sh, hi_sh;
    const int M = start[VOX_BINS];
    const int e0 = blockIdx.x * VOX_BLOCK;
    if (e0 >= M) return;
    const int e = e0 + threadIdx.x;
    const bool live = e < M;
    const unsigned long long mine = live ? gkey[e] : 0ull;
    const int b = live ? vox_bin(mine) : 0;
    const int my_lo = live ? vox_start(start, boff, b) : 0, my_hi = live ? vox_start(start, boff, b + 1) : 0;
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
    vox_emit(t.slot + gslot[e], out, rank);
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
