// Store-pattern micro-benchmark (measurement helper, not part of the product): N regions of R bytes each.
//  A: one wave per region, 1 KiB per store instruction, regions written concurrently by all resident waves
//  B: one 256-thread block per 4 regions, the 4 waves write ONE region together (4 KiB per step), region after region
//  C: plain grid-stride fill of the whole buffer
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void kA(float4 *out, int n, int f4)
{
    const int lane = threadIdx.x & 63, env = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (env >= n) return;
    float4 *dst = out + (size_t)env * f4;
    const float4 v = make_float4(1.f, 0.f, 1.f, 0.f);
    for (int j = lane; j < f4; j += 64) dst[j] = v;
}
__global__ __launch_bounds__(256) void kB(float4 *out, int n, int f4)
{
    const float4 v = make_float4(1.f, 0.f, 1.f, 0.f);
    for (int w = 0; w < 4; ++w) {
        const int env = blockIdx.x * 4 + w;
        if (env >= n) return;
        float4 *dst = out + (size_t)env * f4;
        for (int j = threadIdx.x; j < f4; j += 256) dst[j] = v;
    }
}
__global__ __launch_bounds__(256) void kC(float4 *out, size_t total)
{
    const float4 v = make_float4(1.f, 0.f, 1.f, 0.f);
    for (size_t j = blockIdx.x * 256ull + threadIdx.x; j < total; j += (size_t)gridDim.x * 256) out[j] = v;
}
int main(int argc, char **argv)
{
    const int n = 16384;
    for (int wh : {150, 1024}) {
        const int f4 = 3 * wh;
        const size_t total = (size_t)n * f4;
        float4 *buf; hipMalloc(&buf, total * sizeof(float4));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int k = 0; k < 3; ++k) {
            float best = 1e9f;
            for (int rep = 0; rep < 8; ++rep) {
                hipEventRecord(e0);
                if (k == 0) hipLaunchKernelGGL(kA, dim3(n / 4), dim3(256), 0, 0, buf, n, f4);
                else if (k == 1) hipLaunchKernelGGL(kB, dim3(n / 4), dim3(256), 0, 0, buf, n, f4);
                else hipLaunchKernelGGL(kC, dim3(2048), dim3(256), 0, 0, buf, total);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (rep > 1 && ms < best) best = ms;
            }
            printf("WH=%d pattern %c: %.2f us  %.2f TB/s\n", wh, "ABC"[k], best * 1e3, total * 16.0 / (best * 1e-3) / 1e12);
        }
        hipFree(buf);
    }
    return 0;
}
