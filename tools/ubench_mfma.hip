// ubench_mfma.hip -- issue rate of v_mfma_f32_16x16x4_f32 by operand form (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ITERS 2048
template <int MODE> __global__ void k(float *out, float a, float b)
{
    f32x4 c0 = { a, 1, 2, 3 }, c1 = c0 + 1.f, c2 = c0 + 2.f, c3 = c0 + 3.f;
    f32x4 acc = { 0, 0, 0, 0 };
    const f32x4 z = { 0, 0, 0, 0 };
    float av = a + threadIdx.x, bv = b;
    int mn = 0x7fffffff;
    for (int i = 0; i < ITERS; ++i) {
        f32x4 d0, d1, d2, d3;
        if (MODE == 0) {          // fresh D, C from constant VGPRs (what k_nn_mfma does)
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av + 1, bv, c1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av + 2, bv, c2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av + 3, bv, c3, 0, 0, 0);
        } else if (MODE == 1) {   // fresh D, C = 0
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, z, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av + 1, bv, z, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av + 2, bv, z, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av + 3, bv, z, 0, 0, 0);
        } else {                  // accumulate form
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av + 1, bv, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av + 2, bv, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av + 3, bv, c3, 0, 0, 0);
            d0 = c0; d1 = c1; d2 = c2; d3 = c3;
        }
        if (MODE != 2) {
            mn = min(mn, min(min(__float_as_int(d0[0]), __float_as_int(d1[1])), min(__float_as_int(d2[2]), __float_as_int(d3[3]))));
        }
        bv += 1.0f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + mn + c0[0] + c1[1] + c2[2] + c3[3];
}
template <int MODE> static void run(const char *name, float *out, int wpb)
{
    const int blocks = 256 * 4, threads = 64 * wpb;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, 0.5f); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * wpb * 4.0 * ITERS;
    printf("%-34s waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", name, wpb, ms, n * 2048 / ms / 1e9);
}
int main()
{
    float *out; hipMalloc(&out, 4 * 1024 * 1024);
    for (int wpb = 1; wpb <= 4; wpb *= 2) {
        run<0>("fresh D, C in constant VGPRs", out, wpb);
        run<1>("fresh D, C = 0", out, wpb);
        run<2>("accumulate (C = D)", out, wpb);
    }
    return 0;
}
