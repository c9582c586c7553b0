// ubench_bf16acc.hip -- how exactly does v_mfma_f32_16x16x32_bf16 accumulate? (tools only)
// k_nn_mfma16's filter bound assumes: products exact, and the sum of the <= 24 terms off by at most 23 TRUNCATIONS (2^-23)
// of the running magnitude S = sum |terms|.  This measures the worst |D - exact| / S over random operands with the
// kernel's magnitude pattern (large terms that cancel: |q|^2 - 2 p.q - thr near 0) and reports it in units of 2^-24.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdint>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint4 *a, const uint4 *b, float *d)
{
    union U { uint4 u; bf16x8 v; };
    U A, B; A.u = a[blockIdx.x * 64 + threadIdx.x]; B.u = b[blockIdx.x * 64 + threadIdx.x];
    const f32x4 z = {0, 0, 0, 0};
    const f32x4 r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.v, B.v, z, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[(blockIdx.x * 64 + threadIdx.x) * 4 + i] = r[i];
}
static uint16_t bf16_rne(float v) { uint32_t u; memcpy(&u, &v, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static void split3(float v, uint16_t s[3]) { s[0] = bf16_rne(v); float r1 = v - bf16_f(s[0]); s[1] = bf16_rne(r1); float r2 = r1 - bf16_f(s[1]); s[2] = bf16_rne(r2); }
int main()
{
    const int T = 4096;                                    // 16x16 tiles
    std::vector<uint16_t> A((size_t)T * 64 * 8), B((size_t)T * 64 * 8);
    std::vector<double> exact((size_t)T * 256), mag((size_t)T * 256);
    srand(12345);
    auto rnd = [](float lo, float hi) { return lo + (hi - lo) * (float)rand() / (float)RAND_MAX; };
    for (int t = 0; t < T; ++t) {
        float p[16][3], q[16][3], thr[16], n2[16];
        const float scale = t % 4 == 0 ? 4.5f : (t % 4 == 1 ? 2.0f : (t % 4 == 2 ? 0.5f : 7.0f));
        for (int i = 0; i < 16; ++i) for (int c = 0; c < 3; ++c) p[i][c] = rnd(-scale, scale);
        for (int j = 0; j < 16; ++j) {          // targets near the queries: D close to 0, large terms cancel
            const int i = j;
            for (int c = 0; c < 3; ++c) q[j][c] = p[i][c] + rnd(-0.02f, 0.02f);
            n2[j] = fmaf(q[j][2], q[j][2], fmaf(q[j][1], q[j][1], q[j][0] * q[j][0]));
        }
        for (int i = 0; i < 16; ++i) thr[i] = rnd(0.0f, 1e-3f) - (p[i][0] * p[i][0] + p[i][1] * p[i][1] + p[i][2] * p[i][2]);
        for (int l = 0; l < 64; ++l) {
            const int kb = l >> 4, rc = l & 15;
            uint16_t *a = &A[((size_t)t * 64 + l) * 8], *b = &B[((size_t)t * 64 + l) * 8];
            uint16_t s[3];
            if (kb < 3) {
                split3(-2.0f * p[rc][kb], s); a[0] = s[0]; a[1] = s[0]; a[2] = s[0]; a[3] = s[2]; a[4] = s[1]; a[5] = s[1]; a[6] = a[7] = 0;
                split3(q[rc][kb], s);         b[0] = s[0]; b[1] = s[1]; b[2] = s[2]; b[3] = s[0]; b[4] = s[0]; b[5] = s[1]; b[6] = b[7] = 0;
            } else {
                split3(-thr[rc], s);          a[0] = 0x3f80; a[1] = 0x3f80; a[2] = 0x3f80; a[3] = s[0]; a[4] = s[1]; a[5] = s[2]; a[6] = a[7] = 0;
                split3(n2[rc], s);            b[0] = s[0]; b[1] = s[1]; b[2] = s[2]; b[3] = 0x3f80; b[4] = 0x3f80; b[5] = 0x3f80; b[6] = b[7] = 0;
            }
        }
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double e = 0, m = 0;
            for (int kb = 0; kb < 4; ++kb) for (int s8 = 0; s8 < 8; ++s8) {
                const double x = (double)bf16_f(A[((size_t)t * 64 + kb * 16 + i) * 8 + s8]) * (double)bf16_f(B[((size_t)t * 64 + kb * 16 + j) * 8 + s8]);
                e += x; m += fabs(x);
            }
            exact[(size_t)t * 256 + i * 16 + j] = e; mag[(size_t)t * 256 + i * 16 + j] = m;
        }
    }
    uint4 *da, *db; float *dd;
    hipMalloc(&da, A.size() * 2); hipMalloc(&db, B.size() * 2); hipMalloc(&dd, (size_t)T * 256 * 4);
    hipMemcpy(da, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(T), dim3(64), 0, 0, da, db, dd);
    std::vector<float> D((size_t)T * 256);
    hipMemcpy(D.data(), dd, D.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, worst_abs = 0;
    for (int t = 0; t < T; ++t) for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;                        // C/D layout of the 16x16 forms
        const double e = exact[(size_t)t * 256 + row * 16 + col], m = mag[(size_t)t * 256 + row * 16 + col];
        const double err = fabs((double)D[((size_t)t * 64 + l) * 4 + r] - e);
        if (err / m > worst) worst = err / m;
        if (err > worst_abs) worst_abs = err;
    }
    printf("tiles %d: worst |D - exact| / sum|terms| = %.3f x 2^-24   (the filter's bound: 24 x 2^-23 = 48 x 2^-24); worst absolute %.3e\n",
           T, worst * 16777216.0, worst_abs);
    return worst * 16777216.0 < 48.0 ? 0 : 1;
}
