// ubench_gram.hip -- prototype for DESIGN.md section 9 item (0): the wave's Gram matrix G = sum over its 64 lanes of v v^T
// (v = 8 integer-valued components per lane) on the fp64 matrix cores, against the shipped formulation (products, rounding,
// packed permlane / DPP reductions).  Checks exactness against a host sum and times both.  Tools only.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_gram.hip -o /tmp/ubench_gram && /tmp/ubench_gram
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NC = 8;            // components per lane (a0..a5, b_hi, b_lo in the real thing)
constexpr int REPS = 256;

// v_mfma_f64_16x16x4_f64: A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k (one f64 each), D col = lane & 15, row = (lane >> 4) + 4 r.
// Step s covers the lanes (points) 4 s .. 4 s + 3: lane (c, k) supplies component c of point 4 s + k for BOTH operands.
__global__ __launch_bounds__(256) void k_gram_mfma(const double *__restrict__ V /* [waves][64][NC] */, long long *__restrict__ G /* [waves][64] */, int reps)
{
    __shared__ double slab[4][64 * NC];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * 4 + w;
    double v[NC];
    for (int c = 0; c < NC; ++c) v[c] = V[(wave * 64 + lane) * NC + c];
    d4 acc = { 0, 0, 0, 0 };
    for (int r = 0; r < reps; ++r) {
        double *S = slab[w];
        for (int c = 0; c < NC; ++c) S[lane * NC + c] = v[c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int c = lane & 7, k = lane >> 4;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const double x = S[(4 * s + k) * NC + c];          // lanes 8..15 of a row repeat 0..7: rows / columns 8..15 of D are ignored
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // D[i][j], i = (lane >> 4) + 4 r, j = lane & 15: keep i, j < 8 -> G[i * 8 + j]
    const int j = lane & 15;
    if (j < 8) {
        G[wave * 64 + ((lane >> 4) + 0) * 8 + j] = (long long)acc[0];
        G[wave * 64 + ((lane >> 4) + 4) * 8 + j] = (long long)acc[1];
    }
}

// the shipped formulation, reduced to its shape: per term a product, a rounding, and a wave reduction (plain shuffles here)
__global__ __launch_bounds__(256) void k_gram_valu(const double *__restrict__ V, long long *__restrict__ G, int reps)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * 4 + w;
    double v[NC];
    for (int c = 0; c < NC; ++c) v[c] = V[(wave * 64 + lane) * NC + c];
    double tot[36];
    for (int t = 0; t < 36; ++t) tot[t] = 0;
    for (int r = 0; r < reps; ++r) {
        int t = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int j = i; j < NC; ++j) {
                double x = rint(v[i] * v[j]);
                for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
                tot[t++] += x;
            }
    }
    if (lane == 0) {
        int t = 0;
        for (int i = 0; i < NC; ++i)
            for (int j = i; j < NC; ++j) { G[wave * 64 + i * 8 + j] = (long long)tot[t]; G[wave * 64 + j * 8 + i] = (long long)tot[t]; ++t; }
    }
}

int main()
{
    const int blocks = 1200, waves = blocks * 4;
    std::vector<double> hV((size_t)waves * 64 * NC);
    srand(7);
    for (auto &x : hV) x = (double)((rand() % (1 << 18)) - (1 << 17));        // 18-bit integers here: products exact and even REPS x the 64-lane sums stay below 2^53
                                                                               // (the real thing: 22 bits, one pass -> 2^50)
    double *dV; long long *dG;
    (void)hipMalloc(&dV, hV.size() * sizeof(double)); (void)hipMalloc(&dG, (size_t)waves * 64 * sizeof(long long));
    (void)hipMemcpy(dV, hV.data(), hV.size() * sizeof(double), hipMemcpyHostToDevice);
    std::vector<long long> ref((size_t)waves * 64), got((size_t)waves * 64);
    for (int wv = 0; wv < waves; ++wv)
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 8; ++j) {
                long long s = 0;
                for (int l = 0; l < 64; ++l) s += (long long)hV[((size_t)wv * 64 + l) * NC + i] * (long long)hV[((size_t)wv * 64 + l) * NC + j];
                ref[(size_t)wv * 64 + i * 8 + j] = s;
            }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        for (int reps : { 1, REPS }) {
            (void)hipMemset(dG, 0, (size_t)waves * 64 * sizeof(long long));
            (void)hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_gram_mfma, dim3(blocks), dim3(256), 0, 0, dV, dG, reps);
            else hipLaunchKernelGGL(k_gram_valu, dim3(blocks), dim3(256), 0, 0, dV, dG, reps);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(got.data(), dG, got.size() * sizeof(long long), hipMemcpyDeviceToHost);
            long long bad = 0;
            for (size_t k = 0; k < got.size(); ++k) bad += got[k] != ref[k] * reps;
            printf("%s reps %3d: %.1f us  (%.3f us per Gram of 4800 waves)  mismatches %lld\n", which == 0 ? "mfma f64" : "valu    ", reps, ms * 1e3, ms * 1e3 / reps, bad);
        }
    }
    return 0;
}
