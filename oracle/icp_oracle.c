/*
 * icp_oracle.c -- CPU ORACLE (test infrastructure, see icp_oracle.h header note).
 *
 * Plain C restatement of the plane-ICP path, DESIGN.md section 3 ("normative
 * spec", stages S1..S6).  PARITY UNPINNED except S1 (see icp_oracle.h).
 *
 * Build: gcc -O2 -mfma -ffp-contract=off -fno-fast-math -fopenmp  (oracle/Makefile)
 * -ffp-contract=off + explicit fmaf() where the spec says so makes every float /
 * double operation here an individually rounded IEEE op in a fixed order, which
 * is what the HIP kernels reproduce.
 */
#include "icp_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ params */
void orc_default_params(orc_params *p)
{
    memset(p, 0, sizeof(*p));
    p->width = 640; p->height = 480;
    /* src/convert2PCD.cpp:19-23 -- the intrinsics the committed fixtures were made with */
    p->fx = 525.0; p->fy = 525.0; p->cx = 319.5; p->cy = 235.5; p->depth_factor = 1000.0;
    p->z_filter = 7.0;                 /* parameters.yaml:65 */
    p->iterations = 20;
    p->max_corr_dist = 0.10;
    p->max_plane_residual2 = 0.0;
    p->min_normal_cos = 0.0;
    p->estimator = ORC_EST_POINT2PLANE;
    p->normal_window = 7;              /* src/planarFeatures.cpp:92 */
    p->normal_min_inliers = 41;        /* src/planarFeatures.cpp:128  (> 40) */
    p->normal_inlier_dist = 0.01;      /* src/planarFeatures.cpp:123 */
    p->min_inliers = 12;               /* src/GraphicEnd.h:134 default arg */
    p->error_threshold = 1.0;          /* parameters.yaml:39 */
    p->nn_method = ORC_NN_KDTREE;
    p->threads = 0;
    p->coarse_iterations = 3;
    p->seg_distance_threshold = 0.04f;   /* round 6: chosen from data; the reference's plane extraction key distance_threshold is 0.08 */
    p->seg_plane_percent = 0.2f;         /* parameters.yaml plane_percent */
    p->seg_max_planes = 3;               /* parameters.yaml max_planes */
    p->seg_hypotheses = 64;
    p->seg_seed = 1;
    p->plane_pair_gate = 0;
    p->plane_only = 0;
}

/* rows / solve of the estimator: ORC_EST_PLANE differs from ORC_EST_POINT2PLANE only in where the target normals come from */
static inline int row_form(int estimator) { return estimator == ORC_EST_SVD ? ORC_EST_SVD : ORC_EST_POINT2PLANE; }

static int n_threads(const orc_params *p)
{
#ifdef _OPENMP
    /* threads <= 0: "all that help" -- at most 32.  Measured on the 256-thread GPU host (tools/cpu_scaling.py): 198 it/s on 16
     * threads, 184 on 32, 68 on 128 and THREE on 256 (a team as large as the machine is oversubscribed by the process's other
     * threads, and libgomp's spinning barriers then cost ~100 ms per region); round 3's default was every hardware thread. */
    if (p->threads > 0) return p->threads;
    const int m = omp_get_max_threads();
    return m > 32 ? 32 : m;
#else
    (void)p; return 1;
#endif
}

/* ------------------------------------------------------------ S1 backproject
 * src/convert2PCD.cpp:65-69: z = d/factor; x = (n-cx)*z/fx; y = (m-cy)*z/fy in
 * double, stored to float.  d==0 is dropped there (:61); the PassThrough of
 * src/GraphicEnd.cpp:283-285 (z in [0, z_filter]) becomes part of the mask. */
void orc_backproject(const uint16_t *depth, const orc_params *p, float *xyz4)
{
    const int W = p->width, H = p->height;
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            const int i = v * W + u;
            const uint16_t d = depth[i];
            double z = (double)d / p->depth_factor;
            float *o = xyz4 + 4 * (size_t)i;
            if (d == 0 || !(z <= p->z_filter)) {
                o[0] = o[1] = o[2] = NAN; o[3] = 0.0f;
                continue;
            }
            double x = ((double)u - p->cx) * z / p->fx;
            double y = ((double)v - p->cy) * z / p->fy;
            o[0] = (float)x; o[1] = (float)y; o[2] = (float)z; o[3] = 1.0f;
        }
}

static inline int point_valid(const float *q, float zmax)
{
    return isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]) && q[2] > 0.0f && q[2] <= zmax;
}

/* ------------------------------------------------------- 3x3 Jacobi eigen
 * cyclic Jacobi, fixed 8 sweeps, rotation order (0,1),(0,2),(1,2). */
void orc_eig3(const double A[6], double evals[3], double V[9])
{
    double a[3][3] = { { A[0], A[1], A[2] }, { A[1], A[3], A[4] }, { A[2], A[4], A[5] } };
    double v[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    static const int PP[3] = { 0, 0, 1 }, QQ[3] = { 1, 2, 2 }, RR[3] = { 2, 1, 0 };
    for (int sweep = 0; sweep < 8; ++sweep)
        for (int k = 0; k < 3; ++k) {
            const int p = PP[k], q = QQ[k], r = RR[k];
            const double apq = a[p][q];
            if (apq == 0.0) continue;
            const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
            double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
            if (theta < 0.0) t = -t;
            const double c = 1.0 / sqrt(t * t + 1.0);
            const double s = t * c;
            a[p][p] = a[p][p] - t * apq;
            a[q][q] = a[q][q] + t * apq;
            a[p][q] = a[q][p] = 0.0;
            const double arp = a[r][p], arq = a[r][q];
            a[r][p] = a[p][r] = c * arp - s * arq;
            a[r][q] = a[q][r] = s * arp + c * arq;
            for (int m = 0; m < 3; ++m) {
                const double vmp = v[m][p], vmq = v[m][q];
                v[m][p] = c * vmp - s * vmq;
                v[m][q] = s * vmp + c * vmq;
            }
        }
    for (int k = 0; k < 3; ++k) evals[k] = a[k][k];
    for (int m = 0; m < 3; ++m) for (int k = 0; k < 3; ++k) V[m * 3 + k] = v[m][k]; /* column k = evec k */
}

/* ----------------------------------------- S2: direction of least variance
 * Spec S2 (round 3): the unit eigenvector of the SMALLEST eigenvalue of the 3x3 window covariance C, by power iteration on
 * the adjugate.  adj(C) has the eigenvectors of C with eigenvalues l1*l2, l0*l2, l0*l1, so its DOMINANT eigenvector is C's
 * smallest one, and repeated squaring raises the dominance ratio l1/l0 to the power 2^K:
 *   M = adj(C) (cofactors, each "product - product");
 *   K = 7 times (round 3: 5):  tr = (M00 + M11) + M22, must be a positive normal number; M *= 2^-ilogb(tr) (exact);  M = M * M;
 *   dominance (round 4): ||M||_F^2 >= (1 - 2^-40) tr(M)^2, else no direction (l1 / l0 below ~1.25);
 *   column j of M with the largest diagonal entry (ties: lowest j), normalised.
 * Only +, -, *, one sqrt and three divisions, every one an individually rounded IEEE operation: bit-identical on the CPU and
 * in k_normals.  Error against the exact eigenvector: ~1e-15 + (l0/l1)^128 <= ~1e-12 for every window that passes the
 * dominance test -- so numpy.linalg.eigh reproduces the FLOAT normal of practically every pixel bit for bit, which is what
 * lets the scipy-only goldens run with their own normals (tests/golden/make_independent_golden.py).
 * (Rounds 1-2 ran 8 sweeps of cyclic Jacobi here, 24 rotations with three divisions and two square roots each: 70 % of
 * k_normals' time.  orc_eig3 stays for the plane fits, which need all three eigenpairs of a handful of matrices.) */
#define ORC_EVEC_SQUARINGS 7
int orc_smallest_evec3(const double C[6], double n[3])
{
    const double c00 = C[0], c01 = C[1], c02 = C[2], c11 = C[3], c12 = C[4], c22 = C[5];
    double m00 = c11 * c22 - c12 * c12, m01 = c02 * c12 - c01 * c22, m02 = c01 * c12 - c02 * c11;
    double m11 = c00 * c22 - c02 * c02, m12 = c01 * c02 - c00 * c12, m22 = c00 * c11 - c01 * c01;
    for (int k = 0; k < ORC_EVEC_SQUARINGS; ++k) {
        const double tr = (m00 + m11) + m22;
        uint64_t bits;
        memcpy(&bits, &tr, sizeof bits);
        const int be = (int)((bits >> 52) & 0x7ff);
        if ((bits >> 63) != 0 || be == 0 || be == 0x7ff) return 0;     /* not a positive normal number: no direction */
        const uint64_t sb = (uint64_t)(2046 - be) << 52;                /* 2^-(be - 1023) */
        double sc;
        memcpy(&sc, &sb, sizeof sc);
        const double a00 = m00 * sc, a01 = m01 * sc, a02 = m02 * sc, a11 = m11 * sc, a12 = m12 * sc, a22 = m22 * sc;
        m00 = (a00 * a00 + a01 * a01) + a02 * a02;
        m01 = (a00 * a01 + a01 * a11) + a02 * a12;
        m02 = (a00 * a02 + a01 * a12) + a02 * a22;
        m11 = (a01 * a01 + a11 * a11) + a12 * a12;
        m12 = (a01 * a02 + a11 * a12) + a12 * a22;
        m22 = (a02 * a02 + a12 * a12) + a22 * a22;
    }
    {   /* dominance (round 4): M must be rank one to 2^-40 -- ||M||_F^2 >= (1 - 2^-40) tr(M)^2, i.e. (l0/l1)^128 below ~5e-13,
         * l1/l0 above ~1.25.  A window whose two smallest eigenvalues are closer than that has no well-defined direction of
         * least variance (round 3 kept whatever five squarings had reached: up to 30 degrees off numpy.linalg.eigh on such
         * windows); it gets no normal.  What passes is within ~1e-12 of the exact eigenvector. */
        const double t = (m00 + m11) + m22;
        const double F = ((m00 * m00 + m11 * m11) + m22 * m22) + 2.0 * ((m01 * m01 + m02 * m02) + m12 * m12);
        if (!(F >= (t * t) * (1.0 - 0x1p-40))) return 0;
    }
    double nx = m00, ny = m01, nz = m02, best = m00;
    if (m11 > best) { best = m11; nx = m01; ny = m11; nz = m12; }
    if (m22 > best) { best = m22; nx = m02; ny = m12; nz = m22; }
    const double len = sqrt((nx * nx + ny * ny) + nz * nz);
    if (!(len > 0.0) || !isfinite(len)) return 0;
    n[0] = nx / len; n[1] = ny / len; n[2] = nz / len;
    return 1;
}

/* ------------------------------------------------------------- S2 normals
 * Spec S2 (round 4b: integer window moments).  Every valid pixel's coordinates are quantised once, Xq = rintf(x * 2^16) (float
 * arithmetic: an exact scaling, then round to nearest even -- an integer below 2^20).  For a valid pixel, over the valid pixels of
 * its w x w window: n, S1 = sum Xq, S2 = sum Xq Xq^T -- exact integers, so any summation order gives the same values (the GPU runs
 * them as a box filter) --; n >= normal_min_inliers; C' = n S2 - S1 S1^T (every entry an exact integer below 2^53); the direction of
 * least variance of C' by orc_smallest_evec3, turned toward the camera against the centre's quantised coordinates; stored as float
 * nf.  Planar iff at least normal_min_inliers valid window pixels satisfy, in float arithmetic,
 *     |fmaf(nf.z, Zq, fmaf(nf.y, Yq, nf.x * Xq)) - dqf| <= (float)(normal_inlier_dist * 2^16),
 * dqf = (float)((nx (S1x / n) + ny (S1y / n)) + nz (S1z / n)) with inv = 1.0 / n, S1x * inv ... in double: the LS plane through the
 * window mean.  (Rounds 1-3: fp64 moments about the centre point accumulated in raster order with fused multiply-adds, and the
 * inlier test in double on the unquantised coordinates.) */
static void normal_at(const float *q4 /* quantised frame: (Xq, Yq, Zq, valid) */, int W, int H, int u, int v, int r,
                      int min_in, double in_dist, float *out)
{
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    const float *c0 = q4 + 4 * ((size_t)v * W + u);
    if (!(c0[3] > 0.5f)) return;
    int64_t n = 0, sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
    for (int dv = -r; dv <= r; ++dv)
        for (int du = -r; du <= r; ++du) {
            const int uu = u + du, vv = v + dv;
            if (uu < 0 || uu >= W || vv < 0 || vv >= H) continue;
            const float *q = q4 + 4 * ((size_t)vv * W + uu);
            if (!(q[3] > 0.5f)) continue;
            const int64_t x = (int64_t)q[0], y = (int64_t)q[1], z = (int64_t)q[2];
            ++n; sx += x; sy += y; sz += z;
            sxx += x * x; sxy += x * y; sxz += x * z; syy += y * y; syz += y * z; szz += z * z;
        }
    if (n < min_in) return;
    double C[6];
    C[0] = (double)(n * sxx - sx * sx); C[1] = (double)(n * sxy - sx * sy); C[2] = (double)(n * sxz - sx * sz);
    C[3] = (double)(n * syy - sy * sy); C[4] = (double)(n * syz - sy * sz); C[5] = (double)(n * szz - sz * sz);
    double nv[3];
    const int have = orc_smallest_evec3(C, nv);
    if (!have) return;
    double nx = nv[0], ny = nv[1], nz = nv[2];
    if ((nx * (double)c0[0] + ny * (double)c0[1]) + nz * (double)c0[2] > 0.0) { nx = -nx; ny = -ny; nz = -nz; } /* toward camera */
    const double inv = 1.0 / (double)n;
    const float dqf = (float)((nx * ((double)sx * inv) + ny * ((double)sy * inv)) + nz * ((double)sz * inv));
    const float nxf = (float)nx, nyf = (float)ny, nzf = (float)nz;
    const float thr = (float)(in_dist * 65536.0);
    int cnt = 0;
    for (int dv = -r; dv <= r; ++dv)
        for (int du = -r; du <= r; ++du) {
            const int uu = u + du, vv = v + dv;
            if (uu < 0 || uu >= W || vv < 0 || vv >= H) continue;
            const float *q = q4 + 4 * ((size_t)vv * W + uu);
            if (!(q[3] > 0.5f)) continue;
            const float e = fmaf(nzf, q[2], fmaf(nyf, q[1], nxf * q[0])) - dqf;
            if (fabsf(e) <= thr) ++cnt;
        }
    if (cnt < min_in) return;
    out[0] = nxf; out[1] = nyf; out[2] = nzf; out[3] = 1.0f;
}

void orc_normals(const float *xyz4, const orc_params *p, float *nrm4)
{
    const int W = p->width, H = p->height, r = p->normal_window / 2;
    const float zmax = (float)p->z_filter;
    const int nt = n_threads(p);
    (void)nt;
    float *q4 = malloc(sizeof(float) * 4 * (size_t)W * H);
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int i = 0; i < W * H; ++i) {
        const float *c = xyz4 + 4 * (size_t)i;
        float *q = q4 + 4 * (size_t)i;
        if (point_valid(c, zmax)) { q[0] = rintf(c[0] * 65536.0f); q[1] = rintf(c[1] * 65536.0f); q[2] = rintf(c[2] * 65536.0f); q[3] = 1.0f; }
        else { q[0] = q[1] = q[2] = q[3] = 0.0f; }
    }
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u)
            normal_at(q4, W, H, u, v, r, p->normal_min_inliers, p->normal_inlier_dist, nrm4 + 4 * ((size_t)v * W + u));
    free(q4);
}

/* ------------------------------------------------------------ spec sincos
 * Deterministic sin/cos from +,-,*,floor only (libm's differ between CPU and GPU):
 * k = floor(x*2/pi + 0.5); r = (x - k*PIO2_HI) - k*PIO2_LO; Taylor in r*r (Horner). */
void orc_sincos(double x, double *s, double *c)
{
    const double kf = floor(x * 0.63661977236758134308 + 0.5);
    const double r = (x - kf * 1.57079632673412561417e+00) - kf * 6.07710050650619224932e-11;
    const double z = r * r;
    /* sin r = r*(1 - z/3! + z^2/5! - ... - z^7/15! + z^8/17!) */
    double ps = 1.0 / 355687428096000.0;             /* 1/17! */
    ps = ps * z - 1.0 / 1307674368000.0;             /* 1/15! */
    ps = ps * z + 1.0 / 6227020800.0;                /* 1/13! */
    ps = ps * z - 1.0 / 39916800.0;                  /* 1/11! */
    ps = ps * z + 1.0 / 362880.0;                    /* 1/9!  */
    ps = ps * z - 1.0 / 5040.0;                      /* 1/7!  */
    ps = ps * z + 1.0 / 120.0;                       /* 1/5!  */
    ps = ps * z - 1.0 / 6.0;                         /* 1/3!  */
    const double sr = r + r * (z * ps);
    /* cos r = 1 - z/2! + z^2/4! - ... + z^8/16! - z^9/18! */
    double pc = -1.0 / 6402373705728000.0;           /* 1/18! */
    pc = pc * z + 1.0 / 20922789888000.0;            /* 1/16! */
    pc = pc * z - 1.0 / 87178291200.0;               /* 1/14! */
    pc = pc * z + 1.0 / 479001600.0;                 /* 1/12! */
    pc = pc * z - 1.0 / 3628800.0;                   /* 1/10! */
    pc = pc * z + 1.0 / 40320.0;                     /* 1/8!  */
    pc = pc * z - 1.0 / 720.0;                       /* 1/6!  */
    pc = pc * z + 1.0 / 24.0;                        /* 1/4!  */
    pc = pc * z - 0.5;                               /* 1/2!  */
    const double cr = 1.0 + z * pc;
    long long k = (long long)kf;
    int quad = (int)(((k % 4) + 4) % 4);
    switch (quad) {
    case 0: *s = sr;  *c = cr;  break;
    case 1: *s = cr;  *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr; break;
    }
}

/* --------------------------------------------------------- 6x6 LDL^T solve */
static int ldl6(const double A[6][6], const double b[6], double tr, double x[6])
{
    double L[6][6], D[6], y[6];
    const double floor_piv = 1e-12 * tr / 6.0;
    memset(L, 0, sizeof(L));
    for (int j = 0; j < 6; ++j) {
        double d = A[j][j];
        for (int k = 0; k < j; ++k) d -= (L[j][k] * L[j][k]) * D[k];
        if (!(d > floor_piv)) return 0;
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
    return 1;
}

/* returns 1 = solved, 2 = solved after Tikhonov damping (degenerate), 0 = failed */
int orc_solve6(const double U[21], const double Atb[6], double x[6])
{
    double A[6][6];
    int k = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) { A[r][c] = A[c][r] = U[k]; ++k; }
    double tr = 0.0;
    for (int r = 0; r < 6; ++r) tr += A[r][r];
    if (!(tr > 0.0)) return 0;
    if (ldl6(A, Atb, tr, x)) return 1;
    const double lam = 1e-9 * tr / 6.0;
    for (int r = 0; r < 6; ++r) A[r][r] = A[r][r] + lam;
    if (ldl6(A, Atb, tr, x)) return 2;
    return 0;
}

/* ----------------------------------------------------- 3x3 SVD -> rotation
 * one-sided (Hestenes) Jacobi on the columns of H, at most 12 sweeps; R = V U^T.
 * Round 6: a rotation is skipped when its two columns are orthogonal to 2^-50 (ga^2 <= 2^-100 al be; ga == 0 is the special case of
 * rounds 1-5) and the sweeps stop after one that rotated nothing -- Jacobi converges quadratically, 4-6 sweeps do what 12 did, and
 * the serial chain of divisions and square roots is what an iteration of the svd estimator waits for on the GPU.
 * with the smallest-singular-value pair replaced by cross products (det = +1). */
void orc_svd3_rotation(const double H[9], double R[9])
{
    double g[3][3], v[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } }; /* [row][col] */
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) g[r][c] = H[r * 3 + c];
    static const int PP[3] = { 0, 0, 1 }, QQ[3] = { 1, 2, 2 };
    for (int sweep = 0; sweep < 12; ++sweep) {
        int rotated = 0;
        for (int k = 0; k < 3; ++k) {
            const int p = PP[k], q = QQ[k];
            const double al = (g[0][p] * g[0][p] + g[1][p] * g[1][p]) + g[2][p] * g[2][p];
            const double be = (g[0][q] * g[0][q] + g[1][q] * g[1][q]) + g[2][q] * g[2][q];
            const double ga = (g[0][p] * g[0][q] + g[1][p] * g[1][q]) + g[2][p] * g[2][q];
            if (ga * ga <= 0x1p-100 * (al * be)) continue;
            rotated = 1;
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
    for (int k = 0; k < 3; ++k)
        sg[k] = sqrt((g[0][k] * g[0][k] + g[1][k] * g[1][k]) + g[2][k] * g[2][k]);
    int i0 = 0, i1 = 1, i2 = 2, tmp;
    /* stable descending sort of 3 */
    if (sg[i1] > sg[i0]) { tmp = i0; i0 = i1; i1 = tmp; }
    if (sg[i2] > sg[i1]) { tmp = i1; i1 = i2; i2 = tmp; }
    if (sg[i1] > sg[i0]) { tmp = i0; i0 = i1; i1 = tmp; }
    (void)i2;
    for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if (!(sg[i0] > 0.0) || !(sg[i1] > 1e-14 * sg[i0])) return; /* rank < 2: identity */
    double u0[3], u1[3], u2[3], v0[3], v1[3], v2[3];
    for (int m = 0; m < 3; ++m) {
        u0[m] = g[m][i0] / sg[i0]; u1[m] = g[m][i1] / sg[i1];
        v0[m] = v[m][i0];          v1[m] = v[m][i1];
    }
    u2[0] = u0[1] * u1[2] - u0[2] * u1[1]; u2[1] = u0[2] * u1[0] - u0[0] * u1[2]; u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
    v2[0] = v0[1] * v1[2] - v0[2] * v1[1]; v2[1] = v0[2] * v1[0] - v0[0] * v1[2]; v2[2] = v0[0] * v1[1] - v0[1] * v1[0];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            R[r * 3 + c] = (v0[r] * u0[c] + v1[r] * u1[c]) + v2[r] * u2[c];
}

/* ---------------------------------------------------------- compaction S3 */
typedef struct {
    int n;
    float *x, *y, *z;      /* SoA coordinates of the compacted list */
    float *nx, *ny, *nz;   /* normals (targets only)                */
    int32_t *orig;         /* original linear index                 */
    int32_t *lab;          /* plane label of the point (normal.w - 1; -1 none): the plane-pair gate */
} clist;

static void clist_free(clist *c)
{
    free(c->x); free(c->y); free(c->z); free(c->nx); free(c->ny); free(c->nz); free(c->orig); free(c->lab);
    memset(c, 0, sizeof(*c));
}

static void clist_build_ex(clist *c, const float *xyz4, const float *nrm4, const float *lab4, int N, float zmax)
{
    memset(c, 0, sizeof(*c));
    if (lab4) c->lab = malloc(sizeof(int32_t) * (size_t)(N + 8));
    c->x = malloc(sizeof(float) * (size_t)(N + 8)); c->y = malloc(sizeof(float) * (size_t)(N + 8));
    c->z = malloc(sizeof(float) * (size_t)(N + 8)); c->orig = malloc(sizeof(int32_t) * (size_t)(N + 8));
    if (nrm4) {
        c->nx = malloc(sizeof(float) * (size_t)(N + 8)); c->ny = malloc(sizeof(float) * (size_t)(N + 8));
        c->nz = malloc(sizeof(float) * (size_t)(N + 8));
    }
    int n = 0;
    for (int i = 0; i < N; ++i) {
        const float *q = xyz4 + 4 * (size_t)i;
        if (!point_valid(q, zmax)) continue;
        if (nrm4 && !(nrm4[4 * (size_t)i + 3] > 0.5f)) continue;
        c->x[n] = q[0]; c->y[n] = q[1]; c->z[n] = q[2]; c->orig[n] = i;
        if (nrm4) { c->nx[n] = nrm4[4 * (size_t)i]; c->ny[n] = nrm4[4 * (size_t)i + 1]; c->nz[n] = nrm4[4 * (size_t)i + 2]; }
        if (lab4) c->lab[n] = (int32_t)lab4[4 * (size_t)i + 3] - 1;      /* normal.w = 1 + plane (0: no plane) */
        ++n;
    }
    c->n = n;
}

static void clist_build(clist *c, const float *xyz4, const float *nrm4, int N, float zmax)
{
    clist_build_ex(c, xyz4, nrm4, NULL, N, zmax);
}

/* ------------------------------------------------------------- kd-tree NN
 * exact; same (d2, index) lexicographic minimum as the brute-force scan.
 * Pruning uses fl(ds*ds) > best, which is a valid lower bound of the canonical
 * d2 = fmaf(dz,dz,fmaf(dy,dy,dx*dx)) because every partial of that chain is
 * monotone non-decreasing under round-to-nearest. */
typedef struct { float split; int axis; int left, right; int lo, hi; } kdnode; /* leaf: axis=-1, [lo,hi) */
typedef struct { kdnode *nodes; int n_nodes, cap; int *perm; const clist *pts; } kdtree;

/* (coordinate, index) order of two compact points along one axis: a strict total order, so the split is unique */
static inline int kd_less(const float *arr, int ia, int ib)
{
    return arr[ia] < arr[ib] || (arr[ia] == arr[ib] && ia < ib);
}

/* nth_element on perm[lo, hi) by kd_less: afterwards perm[k] is the element of rank k - lo, everything before it is not
 * greater, everything after it not smaller.  Quickselect with a median-of-three pivot; O(hi - lo) on average (the
 * previous form sorted every level with qsort through file-scope state: O(n log^2 n), serial, 60 % of a 20-iteration
 * run on all cores -- VERDICT r3: "256 threads buy 2.6x one thread"). */
static void kd_select(int *perm, const float *arr, int lo, int hi, int k)
{
    int l = lo, r = hi - 1;
    while (l < r) {
        const int m = l + (r - l) / 2;
        int a = perm[l], b = perm[m], c = perm[r];
        /* median of three as the pivot value */
        int piv = kd_less(arr, a, b) ? (kd_less(arr, b, c) ? b : (kd_less(arr, a, c) ? c : a))
                                     : (kd_less(arr, a, c) ? a : (kd_less(arr, b, c) ? c : b));
        int i = l, j = r;
        while (i <= j) {
            while (kd_less(arr, perm[i], piv)) ++i;
            while (kd_less(arr, piv, perm[j])) --j;
            if (i <= j) { const int t = perm[i]; perm[i] = perm[j]; perm[j] = t; ++i; --j; }
        }
        if (k <= j) r = j;
        else if (k >= i) l = i;
        else return;
    }
}

/* nodes are taken from a preallocated pool with an atomic counter, so that the two halves of a split can be built by
 * different threads (OpenMP tasks for ranges above KD_TASK_MIN points); a leaf holds 7..12 points, so a tree over n
 * points has fewer than n / 3 + 2 nodes.  The tree's SHAPE never shows in a result: the search is exact and ties go
 * to the smallest compact index whatever the traversal order. */
#define KD_TASK_MIN 4096
static int kd_build_rec(kdtree *t, int lo, int hi)
{
    int id;
#pragma omp atomic capture
    id = t->n_nodes++;
    if (hi - lo <= 12) {
        kdnode nd = { 0.0f, -1, -1, -1, lo, hi };
        t->nodes[id] = nd;
        return id;
    }
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int k = lo; k < hi; ++k) {
        const int i = t->perm[k];
        const float c[3] = { t->pts->x[i], t->pts->y[i], t->pts->z[i] };
        for (int a = 0; a < 3; ++a) { if (c[a] < mn[a]) mn[a] = c[a]; if (c[a] > mx[a]) mx[a] = c[a]; }
    }
    int axis = 0;
    if (mx[1] - mn[1] > mx[axis] - mn[axis]) axis = 1;
    if (mx[2] - mn[2] > mx[axis] - mn[axis]) axis = 2;
    const float *arr = axis == 0 ? t->pts->x : (axis == 1 ? t->pts->y : t->pts->z);
    const int mid = (lo + hi) / 2;
    kd_select(t->perm, arr, lo, hi, mid);
    const float split = arr[t->perm[mid]];
    int l, r;
    if (hi - lo >= KD_TASK_MIN) {
#pragma omp task shared(l) firstprivate(t, lo, mid)
        l = kd_build_rec(t, lo, mid);
#pragma omp task shared(r) firstprivate(t, mid, hi)
        r = kd_build_rec(t, mid, hi);
#pragma omp taskwait
    } else {
        l = kd_build_rec(t, lo, mid);
        r = kd_build_rec(t, mid, hi);
    }
    kdnode nd = { split, axis, l, r, lo, hi };
    t->nodes[id] = nd;
    return id;
}

static void kd_build(kdtree *t, const clist *pts, int nt)
{
    t->pts = pts; t->cap = pts->n / 3 + 16; t->n_nodes = 0;
    t->nodes = malloc(sizeof(kdnode) * (size_t)t->cap);
    t->perm = malloc(sizeof(int) * (size_t)(pts->n + 1));
    for (int i = 0; i < pts->n; ++i) t->perm[i] = i;
    if (pts->n > 0) {
        /* a small team: the tree has ~n / 4096 tasks, and libgomp's task queue is one lock per team -- with 256 threads
         * spinning on it the build took seconds (measured on the 256-thread GPU host) */
        const int team = nt > 32 ? 32 : nt;       /* (the default team of every other region: no resizing of libgomp's pool between regions) */
        (void)team;
#pragma omp parallel num_threads(team)
#pragma omp single
        kd_build_rec(t, 0, pts->n);
    }
}

static void kd_free(kdtree *t) { free(t->nodes); free(t->perm); }

static inline float canon_d2(float px, float py, float pz, float qx, float qy, float qz)
{
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

static void kd_query(const kdtree *t, float px, float py, float pz, float gate2, float *best_d2, int *best_j)
{
    if (t->n_nodes == 0) return;
    int stack[128]; float lbs[128]; int sp = 0;
    stack[sp] = 0; lbs[sp] = 0.0f; ++sp;
    const clist *c = t->pts;
    while (sp > 0) {
        --sp;
        const int id = stack[sp];
        const float lb = lbs[sp];
        if (lb > *best_d2 || lb > gate2) continue;
        const kdnode *nd = &t->nodes[id];
        if (nd->axis < 0) {
            for (int k = nd->lo; k < nd->hi; ++k) {
                const int j = t->perm[k];
                const float d2 = canon_d2(px, py, pz, c->x[j], c->y[j], c->z[j]);
                if (d2 < *best_d2 || (d2 == *best_d2 && j < *best_j)) { *best_d2 = d2; *best_j = j; }
            }
            continue;
        }
        const float pa = nd->axis == 0 ? px : (nd->axis == 1 ? py : pz);
        const float ds = nd->split - pa;
        const float far_lb = ds * ds;
        int nearc, farc;
        if (pa <= nd->split) { nearc = nd->left; farc = nd->right; } else { nearc = nd->right; farc = nd->left; }
        /* children of the left subtree have coordinate <= split, right >= split */
        stack[sp] = farc; lbs[sp] = far_lb > lb ? far_lb : lb; ++sp;
        stack[sp] = nearc; lbs[sp] = lb; ++sp;
    }
}

/* ---------------------------------------------------------------- S4 NN */
static void transform_f(const double *T, float Rf[9], float tf[3])
{
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Rf[r * 3 + c] = (float)T[r * 4 + c];
        tf[r] = (float)T[r * 4 + 3];
    }
}

static inline void xform_pt(const float Rf[9], const float tf[3], float x, float y, float z, float *o)
{
    o[0] = fmaf(Rf[2], z, fmaf(Rf[1], y, Rf[0] * x)) + tf[0];
    o[1] = fmaf(Rf[5], z, fmaf(Rf[4], y, Rf[3] * x)) + tf[1];
    o[2] = fmaf(Rf[8], z, fmaf(Rf[7], y, Rf[6] * x)) + tf[2];
}

/* corr[i] = compact target position or -1, d2c[i] = canonical d2 or +inf */
/* spec S4c: does the source pixel take part in a coarse iteration?  Every fourth 8x8-pixel tile, staggered by rows. */
static inline int coarse_active(int pix, int W)
{
    const int v = pix / W, u = pix - v * W;
    return (((u >> 3) + 2 * (v >> 3)) & 3) == 0;
}

static void nn_pass(const clist *src, const clist *tgt, const kdtree *kd, const double *T,
                    float gate2, int method, int nt, int *corr, float *d2c, int coarse_W /* > 0: a coarse iteration of an image this wide */)
{
    float Rf[9], tf[3];
    transform_f(T, Rf, tf);
    (void)nt;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nt)
    for (int i = 0; i < src->n; ++i) {
        if (coarse_W > 0 && !coarse_active(src->orig[i], coarse_W)) { corr[i] = -1; d2c[i] = INFINITY; continue; }
        float p[3];
        xform_pt(Rf, tf, src->x[i], src->y[i], src->z[i], p);
        float best = INFINITY; int bj = -1;
        if (method == ORC_NN_BRUTE) {
            for (int j = 0; j < tgt->n; ++j) {
                const float d2 = canon_d2(p[0], p[1], p[2], tgt->x[j], tgt->y[j], tgt->z[j]);
                if (d2 < best) { best = d2; bj = j; }   /* ascending j: ties keep the smallest */
            }
        } else {
            kd_query(kd, p[0], p[1], p[2], gate2, &best, &bj);
        }
        if (bj >= 0 && best <= gate2) { corr[i] = bj; d2c[i] = best; }
        else { corr[i] = -1; d2c[i] = INFINITY; }
    }
}

/* ------------------------------------------------ S4g optional gates (point-to-plane only)
 * Applied to the correspondences the distance gate kept; a rejected source gets no correspondence (no second-nearest
 * fallback).  Residual: e = (nx*dx + ny*dy) + nz*dz in double -- the b of row_sums --, kept iff e*e <= (double)(float)r2
 * (src/GraphicEnd.cpp~:484-489).  Normal angle: the source pixel's own normal (S2 on the source frame) rotated by the
 * float rotation of xform_pt with the same fma chain, c = fma(rz,tz, fma(ry,ty, rx*tx)) in float, kept iff the source
 * normal is valid and c >= (float)min_normal_cos (stands in for src/GraphicEnd.cpp:542's RANSAC inlier subset). */
static void apply_gates(const clist *src, const clist *tgt, const double *T, const orc_params *p,
                        const float *snrm4, const int32_t *assoc /* plane-pair gate: target plane of every source plane, or NULL */,
                        int nt, int *corr, float *d2c)
{
    const float r2f = (float)p->max_plane_residual2, cminf = (float)p->min_normal_cos;
    if (p->estimator == ORC_EST_SVD || (!(r2f > 0.0f) && !(cminf > 0.0f) && !assoc)) return;
    float Rf[9], tf[3];
    transform_f(T, Rf, tf);
    (void)nt;
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int i = 0; i < src->n; ++i) {
        const int j = corr[i];
        if (j < 0) continue;
        int keep = 1;
        if (r2f > 0.0f) {
            float pf[3];
            xform_pt(Rf, tf, src->x[i], src->y[i], src->z[i], pf);
            const double dx = (double)tgt->x[j] - (double)pf[0], dy = (double)tgt->y[j] - (double)pf[1],
                         dz = (double)tgt->z[j] - (double)pf[2];
            const double nx = tgt->nx[j], ny = tgt->ny[j], nz = tgt->nz[j];
            const double e = (nx * dx + ny * dy) + nz * dz;
            keep = e * e <= (double)r2f;
        }
        if (keep && cminf > 0.0f) {
            const float *ns = snrm4 + 4 * (size_t)src->orig[i];
            const float rx = fmaf(Rf[2], ns[2], fmaf(Rf[1], ns[1], Rf[0] * ns[0]));
            const float ry = fmaf(Rf[5], ns[2], fmaf(Rf[4], ns[1], Rf[3] * ns[0]));
            const float rz = fmaf(Rf[8], ns[2], fmaf(Rf[7], ns[1], Rf[6] * ns[0]));
            const float c = fmaf(rz, tgt->nz[j], fmaf(ry, tgt->ny[j], rx * tgt->nx[j]));
            keep = ns[3] > 0.5f && c >= cminf;
        }
        if (keep && assoc) {     /* spec S4p: plane to its associated plane, clutter (label -1) to clutter */
            const int ls = src->lab[i];
            keep = (ls >= 0 ? assoc[ls] : -1) == tgt->lab[j];
        }
        if (!keep) { corr[i] = -1; d2c[i] = INFINITY; }
    }
}

/* ------------------------------------------------ S4 rows: quantised row vectors and their Gram matrix (round 4)
 * Every correspondence contributes an 8-component INTEGER vector V, and the iteration's totals are the 36 upper-triangle
 * entries of the Gram matrix G = sum V V^T -- exact int64 sums, so neither the order of the summands nor their grouping
 * (one GPU wave per tile on the fp64 matrix cores, one atomic per block, any number of GPUs) changes a bit.
 *   point-to-plane   a = p' x n, b = n . (q - p')   (double, every operation individually rounded, in the order written)
 *                    V = ( rint(a 2^16) [3], rint(n 2^20) [3], rint(b 2^EB), 1 ),  EB = 20 - min(k, 8) with gate = m 2^k, 0.5 <= m < 1
 *                    (|b| <= gate < 2^k, so |V6| <= 2^20; gate 0.10 m: EB = 23, 0.12 um)
 *   svd (Kabsch)     V = ( rint(p' 2^16) [3], rint(q 2^16) [3], 0, 1 )
 * rint = round to nearest, ties to even.  The 29 doubles of the trace / the solve are DERIVED from G (orc_derive_sums):
 *   point-to-plane   A^T A (r, c) = G[r][c] 2^-(e_r + e_c), e = (16,16,16,20,20,20);  A^T b (r) = G[r][6] 2^-(e_r + EB);
 *                    count = G[7][7];  sum b^2 = G[6][6] 2^-2EB
 *   svd              sum p' = G[i][7] 2^-16, sum q = G[3+i][7] 2^-16, sum p' q^T (r, c) = G[r][3+c] 2^-32, count = G[7][7],
 *                    sum |q - p'|^2 = (G00 + G11 + G22 + G33 + G44 + G55 - 2 (G03 + G14 + G25)) 2^-32 (integer arithmetic)
 * each as (double)integer times a power of two.  (Rounds 1-3 rounded the 29 PRODUCTS to 2^-32 instead; the factors' form
 * is what lets a wave's 64 rows go through sixteen v_mfma_f64_16x16x4 instead of 29 products, 29 roundings and 8 packed wave
 * reductions.)  Ranges: |V_i V_j| <= (r 2^16)^2, r = the farthest valid point; N r^2 < 2^28 (checked where the geometry is
 * declared) keeps every total below 2^60. */
int orc_b_exponent(double max_corr_dist)
{
    int k = 0;
    (void)frexp(max_corr_dist, &k);            /* max_corr_dist = m 2^k, 0.5 <= m < 1 */
    /* |b| <= |q - p'| < 2^8 for any gate (every valid point lies within 90.5 m of the sensor): k is clamped at 8, so that a gate
     * of kilometres (PCL's default is sqrt(DBL_MAX)) does not quantise the residual to metres */
    return 20 - (k < 8 ? k : 8);
}

static inline int tri36(int i, int j) { return i * 8 - (i * (i - 1)) / 2 + (j - i); }   /* (i <= j) of the 8x8 upper triangle, row-major */

static int row_vector(const clist *src, const clist *tgt, const float Rf[9], const float tf[3],
                      int estimator, int eb, int i, int j, int64_t v[8])
{
    if (j < 0) return 0;
    float pf[3];
    xform_pt(Rf, tf, src->x[i], src->y[i], src->z[i], pf);
    const double px = pf[0], py = pf[1], pz = pf[2];
    const double qx = tgt->x[j], qy = tgt->y[j], qz = tgt->z[j];
    if (estimator != ORC_EST_SVD) {
        const double dx = qx - px, dy = qy - py, dz = qz - pz;
        const double nx = tgt->nx[j], ny = tgt->ny[j], nz = tgt->nz[j];
        const double a0 = py * nz - pz * ny, a1 = pz * nx - px * nz, a2 = px * ny - py * nx;
        const double b = (nx * dx + ny * dy) + nz * dz;
        v[0] = llrint(a0 * 65536.0); v[1] = llrint(a1 * 65536.0); v[2] = llrint(a2 * 65536.0);
        v[3] = llrint(nx * 1048576.0); v[4] = llrint(ny * 1048576.0); v[5] = llrint(nz * 1048576.0);
        v[6] = llrint(ldexp(b, eb));
    } else {
        v[0] = llrint(px * 65536.0); v[1] = llrint(py * 65536.0); v[2] = llrint(pz * 65536.0);
        v[3] = llrint(qx * 65536.0); v[4] = llrint(qy * 65536.0); v[5] = llrint(qz * 65536.0);
        v[6] = 0;
    }
    v[7] = 1;
    return 1;
}

void orc_derive_sums(const int64_t G[ORC_NRAW], int estimator, int eb, double s[ORC_NSUMS])
{
    for (int k = 0; k < ORC_NSUMS; ++k) s[k] = 0.0;
    if (estimator != ORC_EST_SVD) {
        int k = 0;
        for (int r = 0; r < 6; ++r)
            for (int c = r; c < 6; ++c) s[k++] = ldexp((double)G[tri36(r, c)], -((r < 3 ? 16 : 20) + (c < 3 ? 16 : 20)));
        for (int r = 0; r < 6; ++r) s[21 + r] = ldexp((double)G[tri36(r, 6)], -((r < 3 ? 16 : 20) + eb));
        s[27] = (double)G[tri36(7, 7)];
        s[28] = ldexp((double)G[tri36(6, 6)], -2 * eb);
    } else {
        for (int i = 0; i < 3; ++i) { s[i] = ldexp((double)G[tri36(i, 7)], -16); s[3 + i] = ldexp((double)G[tri36(3 + i, 7)], -16); }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) s[6 + 3 * r + c] = ldexp((double)G[tri36(r, 3 + c)], -32);
        s[27] = (double)G[tri36(7, 7)];
        const int64_t d2 = (G[tri36(0, 0)] + G[tri36(1, 1)] + G[tri36(2, 2)]) + (G[tri36(3, 3)] + G[tri36(4, 4)] + G[tri36(5, 5)])
                         - 2 * (G[tri36(0, 3)] + G[tri36(1, 4)] + G[tri36(2, 5)]);
        s[28] = ldexp((double)d2, -32);
    }
}

/* G += V V^T (upper triangle) */
void orc_accumulate_gram(const int64_t v[8], int64_t G[ORC_NRAW])
{
    int k = 0;
    for (int i = 0; i < 8; ++i)
        for (int j = i; j < 8; ++j) G[k++] += v[i] * v[j];
}

static void accumulate(const clist *src, const clist *tgt, const double *T, int estimator, int eb,
                       const int *corr, int W, int H, int nt, double *total /*29*/, int64_t *raw /* 36 or NULL */)
{
    float Rf[9], tf[3];
    transform_f(T, Rf, tf);
    (void)W; (void)H; (void)nt;
    int64_t G[ORC_NRAW];
    for (int k = 0; k < ORC_NRAW; ++k) G[k] = 0;
#pragma omp parallel num_threads(nt)
    {
        int64_t g[ORC_NRAW], v[8];
        for (int k = 0; k < ORC_NRAW; ++k) g[k] = 0;
#pragma omp for schedule(static)
        for (int i = 0; i < src->n; ++i) {
            if (corr[i] < 0) continue;
            if (row_vector(src, tgt, Rf, tf, estimator, eb, i, corr[i], v)) orc_accumulate_gram(v, g);
        }
#pragma omp critical
        for (int k = 0; k < ORC_NRAW; ++k) G[k] += g[k];
    }
    orc_derive_sums(G, estimator, eb, total);
    if (raw) memcpy(raw, G, sizeof(G));
}

/* ------------------------------------------------------------- S5 update */
static void compose(const double dR[9], const double dt[3], double *T)
{
    double Tn[16];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
            Tn[r * 4 + c] = (dR[r * 3 + 0] * T[0 * 4 + c] + dR[r * 3 + 1] * T[1 * 4 + c]) + dR[r * 3 + 2] * T[2 * 4 + c];
        Tn[r * 4 + 3] = ((dR[r * 3 + 0] * T[3] + dR[r * 3 + 1] * T[7]) + dR[r * 3 + 2] * T[11]) + dt[r];
    }
    Tn[12] = 0.0; Tn[13] = 0.0; Tn[14] = 0.0; Tn[15] = 1.0;
    memcpy(T, Tn, sizeof(Tn));
}

/* returns 1 ok, 2 ok-but-degenerate, 0 failed (T unchanged) */
static int solve_update(const double *sums, int estimator, double *T)
{
    double dR[9], dt[3];
    if (estimator != ORC_EST_SVD) {
        if (sums[27] < 6.0) return 0;
        double x[6];
        const int rc = orc_solve6(sums, sums + 21, x);
        if (!rc) return 0;
        double sa, ca, sb, cb, sg, cg;
        orc_sincos(x[0], &sa, &ca); orc_sincos(x[1], &sb, &cb); orc_sincos(x[2], &sg, &cg);
        dR[0] = cg * cb; dR[1] = (cg * sb) * sa - sg * ca; dR[2] = (cg * sb) * ca + sg * sa;
        dR[3] = sg * cb; dR[4] = (sg * sb) * sa + cg * ca; dR[5] = (sg * sb) * ca - cg * sa;
        dR[6] = -sb;     dR[7] = cb * sa;                  dR[8] = cb * ca;
        dt[0] = x[3]; dt[1] = x[4]; dt[2] = x[5];
        compose(dR, dt, T);
        return rc;
    }
    const double n = sums[27];
    if (n < 3.0) return 0;
    const double pm[3] = { sums[0] / n, sums[1] / n, sums[2] / n };
    const double qm[3] = { sums[3] / n, sums[4] / n, sums[5] / n };
    double H[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) H[r * 3 + c] = sums[6 + r * 3 + c] - (n * pm[r]) * qm[c];
    orc_svd3_rotation(H, dR);
    for (int r = 0; r < 3; ++r)
        dt[r] = qm[r] - ((dR[r * 3 + 0] * pm[0] + dR[r * 3 + 1] * pm[1]) + dR[r * 3 + 2] * pm[2]);
    compose(dR, dt, T);
    return 1;
}

/* ------------------------------------------------------------ S6 result */
static void finish_result(const double *T, const double *last_sums, int degenerate,
                          const orc_params *p, orc_result *res)
{
    memcpy(res->T, T, sizeof(double) * 16);
    const double tr = T[0] + T[5] + T[10];
    double ca = (tr - 1.0) / 2.0;
    if (ca > 1.0) ca = 1.0;
    if (ca < -1.0) ca = -1.0;
    const double ang = acos(ca);
    const double tn = sqrt(T[3] * T[3] + T[7] * T[7] + T[11] * T[11]);
    /* src/GraphicEnd.cpp:618 */
    res->norm = fabs(fmin(ang, 2.0 * M_PI - ang)) + 0.9 * fabs(tn);
    res->inliers = last_sums ? (int)last_sums[27] : 0;
    res->rmse = (last_sums && last_sums[27] > 0.0) ? sqrt(last_sums[28] / last_sums[27]) : 0.0;
    res->status = ORC_OK;
    if (res->inliers < p->min_inliers) res->status = ORC_TOO_FEW_INLIERS;      /* :599 */
    else if (degenerate) res->status = ORC_DEGENERATE;
    else if (res->norm > p->error_threshold) res->status = ORC_NORM_EXCEEDED;  /* :621 */
    if (res->status != ORC_OK)
        for (int k = 0; k < 16; ++k) res->T[k] = (k % 5 == 0) ? 1.0 : 0.0;     /* failure == Identity, :173 */
}

static float gate2_of(const orc_params *p) { return (float)(p->max_corr_dist * p->max_corr_dist); }

int orc_icp(const float *src4, const float *tgt4, const orc_params *p, const double *T_init,
            orc_result *res, int32_t *idx_out, float *d2_out, double *T_trace, double *sums_trace)
{
    const int N = p->width * p->height;
    const float zmax = (float)p->z_filter;
    const int nt = n_threads(p);
    /* ORC_TIMING=1: phase times of this call on stderr (bench.py's cpu_baseline leg reports where the CPU path's time goes) */
    const int timing = getenv("ORC_TIMING") != NULL;
    double tph[6] = { 0, 0, 0, 0, 0, 0 };
    double tp = omp_get_wtime();
#define ORC_PHASE(k) do { if (timing) { const double now__ = omp_get_wtime(); tph[k] += now__ - tp; tp = now__; } } while (0)
    float *nrm4 = NULL, *snrm4 = NULL;
    int32_t assoc_buf[16], *assoc = NULL;
    if (p->estimator == ORC_EST_POINT2PLANE) {
        nrm4 = malloc(sizeof(float) * 4 * (size_t)N);
        orc_normals(tgt4, p, nrm4);
        if ((float)p->min_normal_cos > 0.0f) {          /* the normal-angle gate needs the source frame's normals too */
            snrm4 = malloc(sizeof(float) * 4 * (size_t)N);
            orc_normals(src4, p, snrm4);
        }
    } else if (p->estimator == ORC_EST_PLANE) {         /* spec S2p: the target's planes give the normals */
        float tpl[16 * 8], spl[16 * 8];
        if (p->seg_max_planes < 1 || p->seg_max_planes > 16) return -1;
        nrm4 = malloc(sizeof(float) * 4 * (size_t)N);
        const int nt_pl = orc_plane_normals(tgt4, p, nrm4, tpl, NULL);
        if ((float)p->min_normal_cos > 0.0f || p->plane_pair_gate) {
            snrm4 = malloc(sizeof(float) * 4 * (size_t)N);
            const int ns_pl = orc_plane_normals(src4, p, snrm4, spl, NULL);
            if (p->plane_pair_gate) { orc_plane_assoc(spl, ns_pl, tpl, nt_pl, T_init, assoc_buf); assoc = assoc_buf; }
        }
    }
    ORC_PHASE(0);
    clist src, tgt;
    clist_build_ex(&src, src4, NULL, assoc ? snrm4 : NULL, N, zmax);
    clist_build_ex(&tgt, tgt4, nrm4, assoc ? nrm4 : NULL, N, zmax);
    ORC_PHASE(1);
    kdtree kd; memset(&kd, 0, sizeof(kd));
    if (p->nn_method == ORC_NN_KDTREE) kd_build(&kd, &tgt, n_threads(p));
    ORC_PHASE(2);

    double T[16];
    if (T_init) memcpy(T, T_init, sizeof(T));
    else for (int k = 0; k < 16; ++k) T[k] = (k % 5 == 0) ? 1.0 : 0.0;
    int *corr = malloc(sizeof(int) * (size_t)(src.n + 1));
    float *d2c = malloc(sizeof(float) * (size_t)(src.n + 1));
    double sums[ORC_NSUMS];
    int degenerate = 0, have_sums = 0;
    const float g2 = gate2_of(p);
    if (T_trace) memcpy(T_trace, T, sizeof(T));
    for (int it = 0; it < p->iterations; ++it) {
        const int coarse = it < p->coarse_iterations && it < p->iterations - 1;     /* spec S4c: never the last iteration */
        nn_pass(&src, &tgt, &kd, T, g2, p->nn_method, nt, corr, d2c, coarse ? p->width : 0);
        ORC_PHASE(3);
        apply_gates(&src, &tgt, T, p, snrm4, assoc, nt, corr, d2c);
        accumulate(&src, &tgt, T, row_form(p->estimator), orc_b_exponent(p->max_corr_dist), corr, p->width, p->height, nt, sums, NULL);
        ORC_PHASE(4);
        have_sums = 1;
        if (sums_trace) memcpy(sums_trace + (size_t)it * ORC_NSUMS, sums, sizeof(sums));
        const int rc = solve_update(sums, row_form(p->estimator), T);
        if (rc == 2 || rc == 0) degenerate = 1;   /* damped, or no update at all in this iteration: never a silent OK */
        if (T_trace) memcpy(T_trace + (size_t)(it + 1) * 16, T, sizeof(T));
    }
    if (idx_out) {
        for (int i = 0; i < N; ++i) idx_out[i] = -1;
        if (p->iterations > 0)
            for (int i = 0; i < src.n; ++i) idx_out[src.orig[i]] = corr[i] >= 0 ? tgt.orig[corr[i]] : -1;
    }
    if (d2_out) {
        for (int i = 0; i < N; ++i) d2_out[i] = INFINITY;
        if (p->iterations > 0)
            for (int i = 0; i < src.n; ++i) d2_out[src.orig[i]] = d2c[i];
    }
    finish_result(T, have_sums ? sums : NULL, degenerate, p, res);
    res->iterations = p->iterations;
    res->n_src = src.n; res->n_tgt = tgt.n;
    ORC_PHASE(5);
    if (timing)
        fprintf(stderr, "orc_icp timing threads=%d ms: normals %.2f compact %.2f kd_build %.2f nn %.2f accumulate+solve %.2f output %.2f\n", nt,
                1e3 * tph[0], 1e3 * tph[1], 1e3 * tph[2], 1e3 * tph[3], 1e3 * tph[4], 1e3 * tph[5]);
#undef ORC_PHASE
    if (p->nn_method == ORC_NN_KDTREE) kd_free(&kd);
    free(corr); free(d2c); free(nrm4); free(snrm4);
    clist_free(&src); clist_free(&tgt);
    return res->status;
}

int orc_nn_once(const float *src4, const float *tgt4, const orc_params *p, const double *T,
                int use_normals, int32_t *idx_out, float *d2_out)
{
    return orc_nn_once_ex(src4, tgt4, p, T, use_normals, 0, idx_out, d2_out);
}

/* the same with the source subset of a coarse iteration (spec S4c) when `coarse` is set */
int orc_nn_once_ex(const float *src4, const float *tgt4, const orc_params *p, const double *T,
                   int use_normals, int coarse, int32_t *idx_out, float *d2_out)
{
    const int N = p->width * p->height;
    const float zmax = (float)p->z_filter;
    float *nrm4 = NULL;
    if (use_normals) {      /* 1: the 7x7-window normals (S2); 2: the per-plane normals (S2p) */
        nrm4 = malloc(sizeof(float) * 4 * (size_t)N);
        if (use_normals == 2) orc_plane_normals(tgt4, p, nrm4, NULL, NULL);
        else orc_normals(tgt4, p, nrm4);
    }
    clist src, tgt;
    clist_build(&src, src4, NULL, N, zmax);
    clist_build(&tgt, tgt4, nrm4, N, zmax);
    kdtree kd; memset(&kd, 0, sizeof(kd));
    if (p->nn_method == ORC_NN_KDTREE) kd_build(&kd, &tgt, n_threads(p));
    int *corr = malloc(sizeof(int) * (size_t)(src.n + 1));
    float *d2c = malloc(sizeof(float) * (size_t)(src.n + 1));
    double Tid[16];
    for (int k = 0; k < 16; ++k) Tid[k] = (k % 5 == 0) ? 1.0 : 0.0;
    nn_pass(&src, &tgt, &kd, T ? T : Tid, gate2_of(p), p->nn_method, n_threads(p), corr, d2c, coarse ? p->width : 0);
    for (int i = 0; i < N; ++i) { if (idx_out) idx_out[i] = -1; if (d2_out) d2_out[i] = INFINITY; }
    for (int i = 0; i < src.n; ++i) {
        if (idx_out) idx_out[src.orig[i]] = corr[i] >= 0 ? tgt.orig[corr[i]] : -1;
        if (d2_out) d2_out[src.orig[i]] = d2c[i];
    }
    const int ns = src.n;
    if (p->nn_method == ORC_NN_KDTREE) kd_free(&kd);
    free(corr); free(d2c); free(nrm4);
    clist_free(&src); clist_free(&tgt);
    return ns;
}

/* --------------------------------------------------- per-plane fit (row a6)
 * PCL's LS refinement inside SACSegmentation::segment (setOptimizeCoefficients,
 * src/GraphicEnd.cpp:363): mean + covariance of the inliers -> eigenvector of the
 * smallest eigenvalue, d = -n.c; sign normalised so d >= 0 (src/GraphicEnd.cpp:383-387). */
void orc_fit_planes(const float *xyz4, const int32_t *labels, int n, int nplanes,
                    float *planes, int32_t *counts)
{
    for (int pl = 0; pl < nplanes; ++pl) {
        /* integer fixed-point (2^-16 m) moments about the sensor origin: exact, order-free sums (same arithmetic
         * as P3 of the segmentation spec) */
        planes[4 * pl] = planes[4 * pl + 1] = planes[4 * pl + 2] = planes[4 * pl + 3] = 0.0f;
        counts[pl] = 0;
        int64_t m[10] = { 0 };
        for (int i = 0; i < n; ++i) {
            if (labels[i] != pl) continue;
            const int64_t x = llrint((double)xyz4[4 * (size_t)i] * 65536.0), y = llrint((double)xyz4[4 * (size_t)i + 1] * 65536.0),
                          z = llrint((double)xyz4[4 * (size_t)i + 2] * 65536.0);
            m[0] += 1; m[1] += x; m[2] += y; m[3] += z;
            m[4] += x * x; m[5] += x * y; m[6] += x * z; m[7] += y * y; m[8] += y * z; m[9] += z * z;
        }
        counts[pl] = (int32_t)m[0];
        if (m[0] < 3) continue;
        const double inv = 1.0 / (double)m[0];
        const double mx = (double)m[1] * inv, my = (double)m[2] * inv, mz = (double)m[3] * inv;
        double C[6] = { (double)m[4] * inv - mx * mx, (double)m[5] * inv - mx * my, (double)m[6] * inv - mx * mz,
                        (double)m[7] * inv - my * my, (double)m[8] * inv - my * mz, (double)m[9] * inv - mz * mz };
        double ev[3], V[9];
        orc_eig3(C, ev, V);
        int k = 0;
        if (ev[1] < ev[k]) k = 1;
        if (ev[2] < ev[k]) k = 2;
        double nx = V[0 + k], ny = V[3 + k], nz = V[6 + k];
        const double len = sqrt(nx * nx + ny * ny + nz * nz);
        nx = nx / len; ny = ny / len; nz = nz / len;
        const double cx = mx / 65536.0, cy = my / 65536.0, cz = mz / 65536.0;
        double d = -((nx * cx + ny * cy) + nz * cz);
        if (d < 0.0) { nx = -nx; ny = -ny; nz = -nz; d = -d; }            /* src/GraphicEnd.cpp:383-387 */
        planes[4 * pl] = (float)nx; planes[4 * pl + 1] = (float)ny; planes[4 * pl + 2] = (float)nz; planes[4 * pl + 3] = (float)d;
    }
}

/* --------------------------------------------------- S2p: per-plane normals (ORC_EST_PLANE)
 * SURVEY.md App. C2, per-plane variant: "labels -> per-plane {sum p, sum pp^T, n} -> eigensolve -> (n, d), sign so d >= 0
 * (src/GraphicEnd.cpp:383-387); points take their plane's normal".  The labels and the fits are the segmentation's (P1-P5,
 * seg_oracle.c: the planes the reference extracts per frame, src/GraphicEnd.cpp:353-430).  d >= 0 means n . x = -d <= 0 for the
 * plane's points: the normal looks toward the camera, the orientation S2 gives its window normals.  w carries 1 + plane. */
int orc_plane_normals(const float *xyz4, const orc_params *p, float *nrm4, float *planes8, int32_t *labels)
{
    const int N = p->width * p->height;
    orc_seg_params sp;
    sp.distance_threshold = p->seg_distance_threshold; sp.plane_percent = p->seg_plane_percent;
    sp.max_planes = p->seg_max_planes; sp.hypotheses = p->seg_hypotheses; sp.seed = p->seg_seed;
    float *pl = planes8 ? planes8 : malloc(sizeof(float) * 8 * (size_t)sp.max_planes);
    int32_t *lab = labels ? labels : malloc(sizeof(int32_t) * (size_t)N);
    const int np = orc_segment_planes(xyz4, N, (float)p->z_filter, &sp, pl, lab);
    /* a pixel on no plane keeps its 7x7-window normal (S2) with w = 0.75 -- "a normal, no plane" -- unless plane_only is set:
     * three planes rarely constrain all six degrees of freedom (two of them are often parallel; measured: the synthetic room's
     * target frame yields three z-facing planes and the pose runs away), the clutter between them does */
    float *win = NULL;
    if (!p->plane_only) { win = malloc(sizeof(float) * 4 * (size_t)N); orc_normals(xyz4, p, win); }
    for (int i = 0; i < N; ++i) {
        float *o = nrm4 + 4 * (size_t)i;
        const int r = lab[i];
        if (r >= 0 && r < np) { o[0] = pl[8 * r]; o[1] = pl[8 * r + 1]; o[2] = pl[8 * r + 2]; o[3] = (float)(1 + r); }
        else if (win && win[4 * (size_t)i + 3] > 0.5f) { o[0] = win[4 * (size_t)i]; o[1] = win[4 * (size_t)i + 1]; o[2] = win[4 * (size_t)i + 2]; o[3] = 0.75f; }
        else { o[0] = o[1] = o[2] = o[3] = 0.0f; }
    }
    free(win);
    if (!planes8) free(pl);
    if (!labels) free(lab);
    return np;
}

/* spec S4p, association: plane (n, d) of frame 1 carried by X_2 = R X_1 + t: n' = R n, d' = d - n'.t in double (operations in
 * the order written), sign rule d' >= 0, rounded to float; nearest plane of frame 2 by the squared L2 distance on (a, b, c, d)
 * accumulated as d2 = fmaf(e_k, e_k, d2), k = 0..3, in float; ties -> lowest index (what slam3d_plane_gate / slam3d_match_planes
 * compute; FlannBasedMatcher::match of src/GraphicEnd.cpp:459-484 computed exactly). */
void orc_plane_assoc(const float *planes1, int n1, const float *planes2, int n2, const double *T, int32_t *assoc)
{
    double Tid[16];
    for (int k = 0; k < 16; ++k) Tid[k] = (k % 5 == 0) ? 1.0 : 0.0;
    if (!T) T = Tid;
    for (int i = 0; i < n1; ++i) {
        const double a = planes1[8 * i], b = planes1[8 * i + 1], c = planes1[8 * i + 2], d = planes1[8 * i + 3];
        double n[3];
        for (int r = 0; r < 3; ++r) n[r] = (T[r * 4] * a + T[r * 4 + 1] * b) + T[r * 4 + 2] * c;
        double dd = d - ((n[0] * T[3] + n[1] * T[7]) + n[2] * T[11]);
        if (dd < 0.0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; dd = -dd; }
        const float m[4] = { (float)n[0], (float)n[1], (float)n[2], (float)dd };
        int best = -1;
        float bd = INFINITY;
        for (int j = 0; j < n2; ++j) {
            float d2 = 0.0f;
            for (int k = 0; k < 4; ++k) { const float e = m[k] - planes2[8 * j + k]; d2 = fmaf(e, e, d2); }
            if (d2 < bd) { bd = d2; best = j; }
        }
        assoc[i] = best;
    }
}

/* ------------------------------------------------------- pose error (a14)
 * src/exp1/exp1_2.cpp:167-171,285-289 ; tools/evaluate_rpe.py:138-173 */
void orc_pose_error(const double *Tref, const double *T, double *rot_err, double *trans_err)
{
    /* Tref^-1 = [R^T, -R^T t] */
    double E[16];
    double Ri[9], ti[3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Ri[r * 3 + c] = Tref[c * 4 + r];
    for (int r = 0; r < 3; ++r) ti[r] = -(Ri[r * 3] * Tref[3] + Ri[r * 3 + 1] * Tref[7] + Ri[r * 3 + 2] * Tref[11]);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
            E[r * 4 + c] = Ri[r * 3] * T[c] + Ri[r * 3 + 1] * T[4 + c] + Ri[r * 3 + 2] * T[8 + c];
        E[r * 4 + 3] = Ri[r * 3] * T[3] + Ri[r * 3 + 1] * T[7] + Ri[r * 3 + 2] * T[11] + ti[r];
    }
    *trans_err = sqrt(E[3] * E[3] + E[7] * E[7] + E[11] * E[11]);
    double ca = (E[0] + E[5] + E[10] - 1.0) / 2.0;
    if (ca > 1.0) ca = 1.0;
    if (ca < -1.0) ca = -1.0;
    *rot_err = acos(ca);
}
