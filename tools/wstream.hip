// Store-stream micro-benchmark (measurement helper, not part of the product): what a pure store kernel reaches when the
// written footprint does / does not fit the 256 MiB Infinity Cache.  16 384 regions of R bytes per launch (pattern A of
// wpattern.hip: one wave per region, 1 KiB per store instruction), launches cycle through `nbuf` distinct buffers.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/wstream tools/wstream.hip && tools/exp/wstream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void kA(float4 *out, int n, int f4, int remap)
{
    int b = blockIdx.x;
    if (remap) { const int per = gridDim.x >> 3; b = b < (per << 3) ? (b & 7) * per + (b >> 3) : b; }
    const int lane = threadIdx.x & 63, env = b * 4 + (threadIdx.x >> 6);
    if (env >= n) return;
    float4 *dst = out + (size_t)env * f4;
    const float4 v = make_float4(1.f, 0.f, 1.f, 0.f);
    for (int j = lane; j < f4; j += 64) dst[j] = v;
}
int main()
{
    const int n = 16384;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wh : {150, 1024}) {
        const int f4 = 3 * wh;
        const size_t bytes = (size_t)n * f4 * 16;
        for (int nbuf : {1, 2, 4, 8, 32}) {
            if (bytes * nbuf > (size_t)30 << 30) continue;
            std::vector<float4 *> bufs(nbuf);
            for (auto &p : bufs) { hipMalloc(&p, bytes); hipMemset(p, 0, bytes); }
            for (int remap = 0; remap < 2; ++remap) {
                const int reps = 8 * nbuf > 64 ? 8 * nbuf : 64;
                for (int k = 0; k < nbuf; ++k) hipLaunchKernelGGL(kA, dim3(n / 4), dim3(256), 0, 0, bufs[k], n, f4, remap);
                hipEventRecord(e0);
                for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(kA, dim3(n / 4), dim3(256), 0, 0, bufs[k % nbuf], n, f4, remap);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("region %6d B x 16384, %2d buffer(s) = %7.1f MB footprint, xcd_remap=%d: %7.2f us per launch, %.2f TB/s\n", f4 * 16, nbuf,
                       bytes * nbuf / 1e6, remap, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
            }
            for (auto p : bufs) hipFree(p);
        }
    }
    return 0;
}
