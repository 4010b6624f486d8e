// Sustained bf16 MFMA rate of the two dense shapes under the board's power cap: 16x16x32 (this library's kernels) vs 32x32x16.
// Register operands only (random bit patterns: the power depends on the data), 8 waves per CU x 2 workgroups, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_power tools/micro/mfma_power.hip ; ./mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const uint32_t *seed, float *out, int iters)
{
    uint32_t s = seed[threadIdx.x & 255] ^ (blockIdx.x * 2654435761u);
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        uint32_t w[4];
        for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; w[j] = (s & 0x3fff3fffu) | 0x30003000u; }   // finite bf16 pairs of mixed magnitude
        __builtin_memcpy(&a[i], w, 16);
        for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; w[j] = (s & 0x3fff3fffu) | 0x30003000u; }
        __builtin_memcpy(&b[i], w, 16);
    }
    if (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{ 0, 0, 0, 0 };
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        }
        float t = 0;
        for (int i = 0; i < 16; ++i) t += acc[i][0] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = t;
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)                                     // 8 MFMAs of 32x32x16 = the flops of 16 of 16x16x32
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + r) & 3], acc[i], 0, 0, 0);
        }
        float t = 0;
        for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][15];
        out[blockIdx.x * 256 + threadIdx.x] = t;
    }
}

int main(int argc, char **argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 4.0;
    uint32_t h[256];
    for (int i = 0; i < 256; ++i) h[i] = 0x9e3779b9u * (i + 1);
    uint32_t *seed; float *out;
    hipMalloc(&seed, sizeof h); hipMalloc(&out, 4 * 256 * 2048);
    hipMemcpy(seed, h, sizeof h, hipMemcpyHostToDevice);
    const int blocks = 2048, iters = 20000;                                  // 8 blocks per CU: two waves per SIMD
    const double flop = (double)blocks * 4 * iters * 16 * 16384.0;           // per launch
    for (int shape : { 16, 32, 16, 32 }) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        double best = 0, total_ms = 0; int n = 0;
        while (total_ms < secs * 1e3) {
            hipEventRecord(e0);
            if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, seed, out, iters);
            else hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, seed, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            total_ms += ms; ++n;
            if (total_ms > secs * 500) best = best > flop / ms / 1e12 ? best : flop / ms / 1e12;     // PFLOP/s (flop / ms / 1e12); second half only
        }
        printf("{\"shape\": \"%s\", \"launches\": %d, \"ms_per_launch\": %.3f, \"mean_PFLOPs\": %.3f, \"best_launch_PFLOPs_second_half\": %.3f}\n",
               shape == 16 ? "16x16x32" : "32x32x16", n, total_ms / n, flop * n / total_ms / 1e12, best);
        fflush(stdout);
    }
    return 0;
}
