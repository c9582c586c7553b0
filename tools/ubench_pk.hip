// microbenchmark: the NN scan inner loop with scalar fp32 (7 VALU per candidate) against packed-fp32 candidate pairs
// (5 VALU per candidate), broadcast ds_read_b128 in both.   hipcc -O3 --offload-arch=gfx950 -o ubench_pk ubench_pk.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long key_min(unsigned long long a, unsigned long long b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
    return (unsigned long long)__double_as_longlong(r);
}
constexpr int NC = 64;   // candidates staged per wave
template <int MODE, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_scan(const float4 *__restrict__ cand_g, const float4 *__restrict__ src,
                                                                                          unsigned long long *__restrict__ out, int reps)
{
    __shared__ float4 st_all[4][NC];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 *st = st_all[w];
    st[lane] = cand_g[(blockIdx.x * 4 + w) % 64 * NC + lane];
    __syncthreads();
    const float4 p = src[blockIdx.x * 256 + threadIdx.x];
    const float px = p.x, py = p.y, pz = p.z;
    unsigned long long bkey = 0x7f7fffffffffffffull;
    for (int r = 0; r < reps; ++r) {
        const int cnt = __builtin_amdgcn_readfirstlane(16);
#pragma unroll 1
        for (int qd = 0; qd < 4; ++qd) {
            const float4 *__restrict__ c = st + qd * 16;
            if (MODE == 0) {
                for (int i = 0; i < cnt; i += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float4 q = c[i + u];
                        const float dx = q.y - px, dy = q.z - py, dz = q.w - pz;
                        const float d2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx));
                        const unsigned long long key = ((unsigned long long)(unsigned int)__float_as_int(d2) << 32) | (unsigned int)__float_as_int(q.x);
                        bkey = key_min(bkey, key);
                    }
                }
            } else {
                const f32x2 px2 = {px, px}, py2 = {py, py};
                for (int i = 0; i < cnt; i += 4) {
#pragma unroll
                    for (int u = 0; u < 4; u += 2) {
                        const float4 a = c[i + u];          // x0 x1 y0 y1
                        const float4 b = c[i + u + 1];      // pix0 z0 pix1 z1
                        const f32x2 qx = {a.x, a.y}, qy = {a.z, a.w};
                        const f32x2 dx = qx - px2, dy = qy - py2;
                        const f32x2 t = __builtin_elementwise_fma(dy, dy, dx * dx);
                        const float dz0 = b.y - pz, dz1 = b.w - pz;
                        const float d0 = __fmaf_rn(dz0, dz0, t.x), d1 = __fmaf_rn(dz1, dz1, t.y);
                        const unsigned long long k0 = ((unsigned long long)(unsigned int)__float_as_int(d0) << 32) | (unsigned int)__float_as_int(b.x);
                        const unsigned long long k1 = ((unsigned long long)(unsigned int)__float_as_int(d1) << 32) | (unsigned int)__float_as_int(b.z);
                        bkey = key_min(bkey, k0);
                        bkey = key_min(bkey, k1);
                    }
                }
            }
        }
        asm volatile("" : "+v"(bkey));
    }
    out[blockIdx.x * 256 + threadIdx.x] = bkey;
}
template <int MODE, int WPE>
static void run(const float4 *cand, const float4 *src, unsigned long long *out, int blocks, int reps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        k_scan<MODE, WPE><<<blocks, 256>>>(cand, src, out, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cands = (double)blocks * 4 * reps * 64;     // candidate evaluations (per wave) summed
        if (it == 2)
            printf("mode %d wpe %d: %.3f ms  %.2f cycles@2.4GHz per candidate per SIMD\n", MODE, WPE, ms, ms * 1e-3 * 2.4e9 * 1024 / cands);
    }
}
int main()
{
    const int blocks = 256 * 8 * 4, reps = 200;
    float4 *cand, *src; unsigned long long *out;
    hipMalloc(&cand, 64 * NC * sizeof(float4)); hipMalloc(&src, blocks * 256 * sizeof(float4)); hipMalloc(&out, blocks * 256 * 8);
    std::vector<float4> hc(64 * NC), hs(blocks * 256);
    for (auto &v : hc) v = make_float4(rand() % 1000, rand() / (float)RAND_MAX, rand() / (float)RAND_MAX, rand() / (float)RAND_MAX);
    for (auto &v : hs) v = make_float4(rand() / (float)RAND_MAX, rand() / (float)RAND_MAX, rand() / (float)RAND_MAX, 0);
    hipMemcpy(cand, hc.data(), hc.size() * 16, hipMemcpyHostToDevice); hipMemcpy(src, hs.data(), hs.size() * 16, hipMemcpyHostToDevice);
    run<0, 8>(cand, src, out, blocks, reps); run<1, 8>(cand, src, out, blocks, reps);
    run<0, 4>(cand, src, out, blocks, reps); run<1, 4>(cand, src, out, blocks, reps);
    run<0, 1>(cand, src, out, blocks, reps); run<1, 1>(cand, src, out, blocks, reps);
    return 0;
}
