/* voxel_oracle.c -- CPU restatement of the frame-ingestion filters of GraphicEnd::readimage
 * (src/GraphicEnd.cpp:283-295): pcl::PassThrough on z in [0, z_filter] followed by pcl::VoxelGrid with a
 * cubic leaf (grid_leaf = 0.03), on the 16-byte {x, y, z, rgba} records of the reference's binary PCD files
 * (data/exp1/pcd/1.pcd header: FIELDS x y z rgba, SIZE 4 4 4 4, TYPE F F F U).  SURVEY.md 8(f) f-1.
 *
 * TEST INFRASTRUCTURE ONLY (checker for tests/, smoke() and bench.py's cpu_baseline leg).
 *
 * PARITY UNPINNED: PCL 1.7 is not in the tree and no down-sampled cloud is committed.  Followed from PCL's
 * documented behaviour [UPSTREAM-KNOWLEDGE]: voxel index = floor(coordinate * inverse_leaf) per axis (float),
 * one output point per occupied voxel = centroid of its points (all fields, downsample_all_data), output
 * ordered by the linear voxel index, i.e. lexicographically by (iz, iy, ix).  [BUILD-SPEC]: the centroid is
 * formed from integer fixed-point sums (2^-20 m) so that any accumulation order gives the same bits; colour
 * channels are integer means (truncated). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "icp_oracle.h"

typedef struct { uint64_t key; int64_t sx, sy, sz; uint32_t c[4]; uint32_t n; } vox_t;

static int cmp_key_idx(const void *a, const void *b)
{
    const uint64_t x = ((const uint64_t *)a)[0], y = ((const uint64_t *)b)[0];
    return x < y ? -1 : (x > y ? 1 : 0);
}

uint64_t orc_voxel_key(float x, float y, float z, float inv_leaf)
{
    const int64_t ix = (int64_t)floorf(x * inv_leaf) + 1048576, iy = (int64_t)floorf(y * inv_leaf) + 1048576,
                  iz = (int64_t)floorf(z * inv_leaf) + 1048576;
    const uint64_t cx = (uint64_t)(ix < 0 ? 0 : (ix > 2097151 ? 2097151 : ix));
    const uint64_t cy = (uint64_t)(iy < 0 ? 0 : (iy > 2097151 ? 2097151 : iy));
    const uint64_t cz = (uint64_t)(iz < 0 ? 0 : (iz > 2097151 ? 2097151 : iz));
    return (cz << 42) | (cy << 21) | cx;
}

int orc_voxel_grid(const float *pts /* n x {x,y,z,rgba bits} */, int n, float leaf, float zmax, float *out)
{
    return orc_voxel_grid_range(pts, n, leaf, 0.0f, zmax, out);
}

/* src/saveOutput.cpp:84-92 (spec T1): PassThrough z in [0, zmax], then pcl::transformPointCloud by T (row-major 4x4):
 * each coordinate = fma(R_r2, z, fma(R_r1, y, R_r0 * x)) + t_r in double, rounded once to float; dropped -> NaN */
int orc_pass_transform(const float *pts, int n, float zmax, const double *T, float *out)
{
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = pts + 4 * (size_t)i;
        float *o = out + 4 * (size_t)i;
        if (isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]) && p[2] >= 0.0f && p[2] <= zmax) {
            const double x = p[0], y = p[1], z = p[2];
            o[0] = (float)(fma(T[2], z, fma(T[1], y, T[0] * x)) + T[3]);
            o[1] = (float)(fma(T[6], z, fma(T[5], y, T[4] * x)) + T[7]);
            o[2] = (float)(fma(T[10], z, fma(T[9], y, T[8] * x)) + T[11]);
            o[3] = p[3];
            ++kept;
        } else {
            o[0] = o[1] = o[2] = NAN; o[3] = 0.0f;
        }
    }
    return kept;
}

/* zmin = -inf, zmax = +inf: pcl::VoxelGrid alone */
int orc_voxel_grid_range(const float *pts, int n, float leaf, float zmin, float zmax, float *out)
{
    const float inv_leaf = 1.0f / leaf;
    uint64_t *ki = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (size_t)(n > 0 ? n : 1));
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = pts + 4 * (size_t)i;
        if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]) && p[2] >= zmin && p[2] <= zmax)) continue;   /* PassThrough */
        ki[2 * m] = orc_voxel_key(p[0], p[1], p[2], inv_leaf);
        ki[2 * m + 1] = (uint64_t)i;
        ++m;
    }
    qsort(ki, (size_t)m, 2 * sizeof(uint64_t), cmp_key_idx);
    int nv = 0;
    for (int a = 0; a < m;) {
        int b = a;
        int64_t sx = 0, sy = 0, sz = 0;
        uint64_t c[4] = { 0, 0, 0, 0 };
        while (b < m && ki[2 * b] == ki[2 * a]) {
            const float *p = pts + 4 * (size_t)ki[2 * b + 1];
            uint32_t rgba;
            memcpy(&rgba, p + 3, 4);
            sx += llrint((double)p[0] * 1048576.0); sy += llrint((double)p[1] * 1048576.0); sz += llrint((double)p[2] * 1048576.0);
            for (int k = 0; k < 4; ++k) c[k] += (rgba >> (8 * k)) & 0xffu;
            ++b;
        }
        const double cnt = (double)(b - a);
        float *o = out + 4 * (size_t)nv;
        o[0] = (float)(((double)sx / cnt) / 1048576.0);
        o[1] = (float)(((double)sy / cnt) / 1048576.0);
        o[2] = (float)(((double)sz / cnt) / 1048576.0);
        uint32_t rgba = 0;
        for (int k = 0; k < 4; ++k) rgba |= (uint32_t)(c[k] / (uint64_t)(b - a)) << (8 * k);
        memcpy(o + 3, &rgba, 4);
        ++nv;
        a = b;
    }
    free(ki);
    return nv;
}
