// ubench.hip -- instruction-rate probes for gfx950 used to shape the NN inner loop (tools only).
// hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>

#define ITERS 4096

__global__ void k_fma(float *out, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        x0 = __fmaf_rn(x0, a, b); x1 = __fmaf_rn(x1, a, b); x2 = __fmaf_rn(x2, a, b); x3 = __fmaf_rn(x3, a, b);
        x4 = __fmaf_rn(x4, a, b); x5 = __fmaf_rn(x5, a, b); x6 = __fmaf_rn(x6, a, b); x7 = __fmaf_rn(x7, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

typedef float float2v __attribute__((ext_vector_type(2)));
__global__ void k_pkfma(float *out, float a, float b)
{
    float2v x0 = { (float)threadIdx.x, 1.f }, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    float2v av = { a, a }, bv = { b, b };
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(av), "v"(bv));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(av), "v"(bv));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(av), "v"(bv));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(av), "v"(bv));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x4) : "v"(av), "v"(bv));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x5) : "v"(av), "v"(bv));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x6) : "v"(av), "v"(bv));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x7) : "v"(av), "v"(bv));
    }
    float2v s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// the canonical candidate step: 3 sub, mul, 2 fma, u64 compare-select (10 VALU)
__global__ void k_cand_u64(unsigned long long *out, const float4 *__restrict__ cand, int n)
{
    const float px = threadIdx.x * 0.01f, py = threadIdx.x * 0.02f, pz = 1.0f;
    unsigned long long best = ~0ull;
    for (int r = 0; r < ITERS / 64; ++r)
        for (int k = 0; k < n; ++k) {
            const float4 q = cand[k];
            const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
            const float d2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx));
            const unsigned long long key = ((unsigned long long)(unsigned)__float_as_int(d2) << 32) | (unsigned)__float_as_int(q.w);
            best = key < best ? key : best;
        }
    out[blockIdx.x * blockDim.x + threadIdx.x] = best;
}

// strict-less float compare + index select (9 VALU: no 64-bit compare)
__global__ void k_cand_f32(unsigned long long *out, const float4 *__restrict__ cand, int n)
{
    const float px = threadIdx.x * 0.01f, py = threadIdx.x * 0.02f, pz = 1.0f;
    float bd = 1e30f; int bj = -1;
    for (int r = 0; r < ITERS / 64; ++r)
        for (int k = 0; k < n; ++k) {
            const float4 q = cand[k];
            const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
            const float d2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx));
            const bool lt = d2 < bd;
            bd = lt ? d2 : bd; bj = lt ? __float_as_int(q.w) : bj;
        }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((unsigned long long)(unsigned)__float_as_int(bd) << 32) | (unsigned)bj;
}

// min-only (7 VALU): is there anything below the bound?  (index recovered in a rare second pass)
__global__ void k_cand_min(unsigned long long *out, const float4 *__restrict__ cand, int n)
{
    const float px = threadIdx.x * 0.01f, py = threadIdx.x * 0.02f, pz = 1.0f;
    float bd = 1e30f;
    for (int r = 0; r < ITERS / 64; ++r)
        for (int k = 0; k < n; ++k) {
            const float4 q = cand[k];
            const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
            const float d2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx));
            bd = fminf(bd, d2);
        }
    out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_int(bd);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(float *out, float a, float b)
{
    f32x4 c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;
    float av = a + threadIdx.x, bv = b;
    for (int i = 0; i < ITERS; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

template <class F> static float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main()
{
    const int blocks = 256 * 8, threads = 256;          // 8 waves/SIMD
    float *out; hipMalloc(&out, sizeof(float) * blocks * threads);
    unsigned long long *out64; hipMalloc(&out64, 8 * blocks * threads);
    std::vector<float4> hc(64); for (int i = 0; i < 64; ++i) { hc[i] = make_float4(i * 0.1f, i * 0.2f, 1.f + i * 0.01f, 0.f); memcpy(&hc[i].w, &i, 4); }
    float4 *cand; hipMalloc(&cand, sizeof(float4) * 64); hipMemcpy(cand, hc.data(), sizeof(float4) * 64, hipMemcpyHostToDevice);
    const double waves = (double)blocks * threads / 64.0;
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f); });
    printf("v_fma_f32      : %.3f ms  %.1f TFLOP/s  (%.2f cyc/wave-instr/SIMD @2.4GHz)\n", ms, waves * 64 * 8.0 * ITERS * 2 / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / (waves * 8.0 * ITERS));
    ms = timeit([&] { hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f); });
    printf("v_pk_fma_f32   : %.3f ms  %.1f TFLOP/s  (%.2f cyc/wave-instr/SIMD)\n", ms, waves * 64 * 8.0 * ITERS * 4 / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / (waves * 8.0 * ITERS));
    ms = timeit([&] { hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f); });
    printf("mfma 16x16x4f32: %.3f ms  %.1f TFLOP/s  (%.2f cyc/instr/SIMD)\n", ms, waves * 4.0 * ITERS * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / (waves * 4.0 * ITERS));
    const double pairs = waves * 64.0 * (ITERS / 64) * 64;
    ms = timeit([&] { hipLaunchKernelGGL(k_cand_u64, dim3(blocks), dim3(threads), 0, 0, out64, cand, 64); });
    printf("cand u64 key   : %.3f ms  %.2f Gpair/s  (%.2f cyc per candidate per wave)\n", ms, pairs / ms / 1e6, ms * 1e-3 * 2.4e9 * 1024 / (pairs / 64));
    ms = timeit([&] { hipLaunchKernelGGL(k_cand_f32, dim3(blocks), dim3(threads), 0, 0, out64, cand, 64); });
    printf("cand f32 lt    : %.3f ms  %.2f Gpair/s  (%.2f cyc per candidate per wave)\n", ms, pairs / ms / 1e6, ms * 1e-3 * 2.4e9 * 1024 / (pairs / 64));
    ms = timeit([&] { hipLaunchKernelGGL(k_cand_min, dim3(blocks), dim3(threads), 0, 0, out64, cand, 64); });
    printf("cand min only  : %.3f ms  %.2f Gpair/s  (%.2f cyc per candidate per wave)\n", ms, pairs / ms / 1e6, ms * 1e-3 * 2.4e9 * 1024 / (pairs / 64));
    return 0;
}
