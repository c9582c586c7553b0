/* seg_oracle.c -- CPU restatement of the batched plane segmentation.
 *
 * TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
 * the checker; never by the product path.
 *
 * PARITY UNPINNED: the reference delegates this step to pcl::SACSegmentation (PCL 1.7, not in the tree, not
 * buildable here; src/GraphicEnd.cpp:353-430) whose RANSAC draws from rand() after srand(time(0))
 * (src/GraphicEnd.cpp:69), and no expected plane coefficients are committed.  What follows the reference:
 * the loop (while remaining > plane_percent * n and planes < max_planes, :372,:424), the model (plane,
 * distance threshold :365), the least-squares refinement of the consensus set (setOptimizeCoefficients(true),
 * :362) followed by re-selection of the inliers, the sign rule d >= 0 (:383-387) and the removal of the
 * inliers from the working cloud (:419-420).  What is [BUILD-SPEC] (P1-P5 in DESIGN.md): a fixed number of
 * hypotheses per round drawn from a counter-based splitmix64 stream, unnormalised plane test, integer
 * fixed-point moments (order-free, so any parallel schedule gives the same bits), cyclic-Jacobi eigenvector. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "icp_oracle.h"

static inline int seg_valid(const float *q, float zmax)
{
    return isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]) && q[2] > 0.0f && q[2] <= zmax;
}

static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

typedef struct { float nx, ny, nz, dd, thr2nn; int ok; float p0[3]; } hyp_t;

/* P2: hypothesis h of round r */
static hyp_t make_hyp(const float *xyz4, const int32_t *lab, int n, uint64_t seed, int r, int h, float thr)
{
    hyp_t H;
    memset(&H, 0, sizeof H);
    uint64_t st = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(1 + r * 4096 + h);
    int pick[3], np = 0;
    for (int t = 0; t < ORC_SEG_DRAWS && np < 3; ++t) {
        st += 0x9E3779B97F4A7C15ull;
        const uint64_t o = mix64(st);
        const int pix = (int)(((o >> 32) * (uint64_t)n) >> 32);
        if (lab[pix] != -1) continue;
        if ((np > 0 && pick[0] == pix) || (np > 1 && pick[1] == pix)) continue;
        pick[np++] = pix;
    }
    if (np < 3) return H;
    const float *p0 = xyz4 + 4 * (size_t)pick[0], *p1 = xyz4 + 4 * (size_t)pick[1], *p2 = xyz4 + 4 * (size_t)pick[2];
    const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
    const float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
    const float nx = fmaf(ay, bz, -(az * by));
    const float ny = fmaf(az, bx, -(ax * bz));
    const float nz = fmaf(ax, by, -(ay * bx));
    const float nn = fmaf(nz, nz, fmaf(ny, ny, nx * nx));
    if (!(nn > 1e-16f)) return H;
    H.nx = nx; H.ny = ny; H.nz = nz;
    H.dd = -fmaf(nx, p0[0], fmaf(ny, p0[1], nz * p0[2]));
    H.thr2nn = (thr * thr) * nn;
    H.ok = 1;
    H.p0[0] = p0[0]; H.p0[1] = p0[1]; H.p0[2] = p0[2];
    return H;
}

static inline int hyp_inlier(const hyp_t *H, const float *q)
{
    const float e = fmaf(H->nx, q[0], fmaf(H->ny, q[1], H->nz * q[2])) + H->dd;
    return e * e <= H->thr2nn;
}

int orc_segment_planes(const float *xyz4, int n, float zmax, const orc_seg_params *sp,
                       float *planes /* max_planes*8: a b c d cx cy cz count */, int32_t *labels)
{
    const int H = sp->hypotheses, maxp = sp->max_planes;
    const float thr = sp->distance_threshold;
    int n_valid = 0;
    for (int i = 0; i < n; ++i) {
        const int ok = seg_valid(xyz4 + 4 * (size_t)i, zmax);
        labels[i] = ok ? -1 : -2;
        n_valid += ok;
    }
    memset(planes, 0, sizeof(float) * 8 * (size_t)maxp);
    int remaining = n_valid, nplanes = 0;
    hyp_t *hy = (hyp_t *)malloc(sizeof(hyp_t) * (size_t)H);
    int *cnt = (int *)malloc(sizeof(int) * (size_t)H);
    for (int r = 0; r < maxp; ++r) {
        if (n_valid < 3 || !((double)remaining > (double)sp->plane_percent * (double)n_valid)) break;
        for (int h = 0; h < H; ++h) { hy[h] = make_hyp(xyz4, labels, n, sp->seed, r, h, thr); cnt[h] = 0; }
        for (int i = 0; i < n; ++i) {
            if (labels[i] != -1) continue;
            const float *q = xyz4 + 4 * (size_t)i;
            for (int h = 0; h < H; ++h) if (hy[h].ok && hyp_inlier(&hy[h], q)) cnt[h]++;
        }
        int best = 0;
        for (int h = 1; h < H; ++h) if (cnt[h] > cnt[best]) best = h;
        if (cnt[best] < 3) break;
        /* P3: fixed-point moments of the consensus set about the first sample point */
        const hyp_t *B = &hy[best];
        const double ox = B->p0[0], oy = B->p0[1], oz = B->p0[2];
        int64_t S[10] = { 0 };
        for (int i = 0; i < n; ++i) {
            if (labels[i] != -1) continue;
            const float *q = xyz4 + 4 * (size_t)i;
            if (!hyp_inlier(B, q)) continue;
            const int64_t qx = llrint(((double)q[0] - ox) * 65536.0), qy = llrint(((double)q[1] - oy) * 65536.0),
                          qz = llrint(((double)q[2] - oz) * 65536.0);
            S[0] += 1; S[1] += qx; S[2] += qy; S[3] += qz;
            S[4] += qx * qx; S[5] += qx * qy; S[6] += qx * qz; S[7] += qy * qy; S[8] += qy * qz; S[9] += qz * qz;
        }
        const double inv = 1.0 / (double)S[0];
        const double mx = (double)S[1] * inv, my = (double)S[2] * inv, mz = (double)S[3] * inv;
        double C[6], ev[3], V[9];
        C[0] = (double)S[4] * inv - mx * mx; C[1] = (double)S[5] * inv - mx * my; C[2] = (double)S[6] * inv - mx * mz;
        C[3] = (double)S[7] * inv - my * my; C[4] = (double)S[8] * inv - my * mz; C[5] = (double)S[9] * inv - mz * mz;
        orc_eig3(C, ev, V);
        int k = 0;
        if (ev[1] < ev[k]) k = 1;
        if (ev[2] < ev[k]) k = 2;
        double nx = V[0 + k], ny = V[3 + k], nz = V[6 + k];
        const double len = sqrt(nx * nx + ny * ny + nz * nz);
        nx = nx / len; ny = ny / len; nz = nz / len;
        const double cx = ox + mx / 65536.0, cy = oy + my / 65536.0, cz = oz + mz / 65536.0;
        double d = -((nx * cx + ny * cy) + nz * cz);
        if (d < 0.0) { nx = -nx; ny = -ny; nz = -nz; d = -d; }            /* src/GraphicEnd.cpp:383-387 */
        const float a = (float)nx, b = (float)ny, c = (float)nz, df = (float)d;
        /* P4: the plane's points = unassigned points within thr of the refined plane */
        int got = 0;
        for (int i = 0; i < n; ++i) {
            if (labels[i] != -1) continue;
            const float *q = xyz4 + 4 * (size_t)i;
            const float e = fmaf(a, q[0], fmaf(b, q[1], c * q[2])) + df;
            if (fabsf(e) <= thr) { labels[i] = r; ++got; }
        }
        if (got == 0) break;
        float *P = planes + 8 * (size_t)r;
        P[0] = a; P[1] = b; P[2] = c; P[3] = df; P[4] = (float)cx; P[5] = (float)cy; P[6] = (float)cz; P[7] = (float)got;
        nplanes = r + 1;
        remaining -= got;
    }
    free(hy); free(cnt);
    return nplanes;
}
