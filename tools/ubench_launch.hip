// developer tool: what an (almost) empty launch costs as a function of its grid -- hipcc --offload-arch=gfx950 -O3 tools/ubench_launch.hip -o tools/ubench_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_empty(int *p) { if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 1; }
__global__ __launch_bounds__(256) void k_load(const float4 *__restrict__ a, float *out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    float4 v = a[i % n];
    if (v.x == 12345.678f) out[i] = v.y;
}
int main()
{
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    float4 *a; hipMalloc(&a, 16 << 20); hipMemset(a, 0, 16 << 20);
    float *o; hipMalloc(&o, 4 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = { 1, 64, 150, 300, 600, 1200, 2400, 4800 };
    for (int mode = 0; mode < 2; ++mode)
        for (int g : grids) {
            hipGraph_t graph; hipGraphExec_t ex;
            hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
            for (int k = 0; k < 200; ++k) {
                if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, s, (int *)nullptr);
                else hipLaunchKernelGGL(k_load, dim3(g), dim3(256), 0, s, a, o, 1 << 20);
            }
            hipStreamEndCapture(s, &graph); hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0);
            hipGraphLaunch(ex, s); hipStreamSynchronize(s);
            hipEventRecord(e0, s);
            for (int r = 0; r < 5; ++r) hipGraphLaunch(ex, s);
            hipEventRecord(e1, s); hipStreamSynchronize(s);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s grid %5d x 256: %.2f us per launch (graph of 200 dependent launches)\n", mode ? "one 16-B load" : "empty       ", g, 1e3 * ms / 1000);
            hipGraphExecDestroy(ex); hipGraphDestroy(graph);
        }
    return 0;
}
