// Store-stream micro-benchmark, round 3 (measurement helper, not part of the product): what decides the time of a kernel that
// writes 16 384 observation regions of 7 200 B when the launches cycle through 32 DISTINCT buffers (3.8 GB: a true HBM stream,
// what a rollout collector does) -- the store cache policy (plain / sc1 = write-through / nt / combinations), the shape of a
// wave's stream (one 7 200-B region per wave: 32-B aligned, two partial 128-B lines per region; or four regions per wave as
// ONE 28 800-B line-aligned stream), and a dead time in front of the first store (the step logic of k_step4).
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/wstream2 tools/wstream2.hip && tools/exp/wstream2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// POL = the aux cache bits of the buffer store: 0 plain, 1 sc0, 2 nt, 16 sc1 (write-through), combinations by OR
template <int POL>
__device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t rs, int f4_index, v4f v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rs, f4_index * 16, 0, POL);
}

__device__ __forceinline__ int xcd_block(int b, int nblocks)
{
    const int per = nblocks >> 3;
    return b < (per << 3) ? (b & 7) * per + (b >> 3) : b;
}

__device__ __forceinline__ void dead_time(int ticks)              // 100 MHz constant clock
{
    if (ticks <= 0) return;
    const unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
}

// SHAPE 0: one wave per region (f4 float4s), 4 regions per block.   SHAPE 1: one wave per 4 regions as one stream.
// SHAPE 2: one wave per 4 regions, region after region (7 full store instructions + a 2-lane tail each: what k_step4 did in round 2).
template <int POL, int SHAPE>
__global__ __launch_bounds__(256) void kst(v4f *out, int n, int f4, int delay, int mode)
{
    const int b = xcd_block(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // mode bit 0: all-zero data; bit 1: lane-dependent data; bit 2: the dead time is staggered over the waves (0 .. delay)
    v4f v = { 1.f, 0.f, 1.f, 0.f };
    if (mode & 1) v = v4f{ 0.f, 0.f, 0.f, 0.f };
    if (mode & 2) v = v4f{ (float)threadIdx.x, (float)blockIdx.x, 1.f / (1 + threadIdx.x), 3.f };
    if (mode & 4) delay = (int)(((blockIdx.x * 4u + (threadIdx.x >> 6)) * 2654435761u >> 16) % (unsigned)(delay + 1));
    dead_time(delay);
    if (SHAPE == 0) {
        const int env = b * 4 + wv;
        if (env >= n) return;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)__builtin_amdgcn_readfirstlane(env) * f4, 0, f4 * 16, 0x00020000);
        for (int j = lane; j < f4; j += 64) st<POL>(rs, j, v);
    } else if (SHAPE == 1) {
        const int env0 = (b * 4 + wv) * 4;
        if (env0 >= n) return;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)__builtin_amdgcn_readfirstlane(env0) * f4, 0, 4 * f4 * 16, 0x00020000);
        for (int j = lane; j < 4 * f4; j += 64) st<POL>(rs, j, v);
    } else {
        const int env0 = (b * 4 + wv) * 4;
        if (env0 >= n) return;
        for (int r = 0; r < 4; ++r) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)(__builtin_amdgcn_readfirstlane(env0) + r) * f4, 0, f4 * 16, 0x00020000);
            for (int j = lane; j < f4; j += 64) st<POL>(rs, j, v);
        }
    }
}

template <int POL, int SHAPE>
static float run(std::vector<v4f *> &bufs, int n, int f4, int delay, int reps, int mode = 0)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nbuf = (int)bufs.size();
    const int blocks = SHAPE == 0 ? n / 4 : n / 16;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        for (int k = 0; k < nbuf; ++k) hipLaunchKernelGGL((kst<POL, SHAPE>), dim3(blocks), dim3(256), 0, 0, bufs[k], n, f4, delay, mode);
        hipEventRecord(e0);
        for (int k = 0; k < reps; ++k) hipLaunchKernelGGL((kst<POL, SHAPE>), dim3(blocks), dim3(256), 0, 0, bufs[k % nbuf], n, f4, delay, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best / reps * 1e3f;
}

int main()
{
    const int n = 16384, f4 = 450;
    const size_t bytes = (size_t)n * f4 * 16;
    auto pol = [](int p) { return p == 0 ? "plain" : p == 16 ? "sc1" : p == 2 ? "nt" : p == 17 ? "sc0 sc1" : p == 18 ? "sc1 nt" : p == 1 ? "sc0" : "sc0 sc1 nt"; };
    const char *shape[] = { "wave per region", "wave per 4 regions, one aligned stream", "wave per 4 regions, region by region" };
    for (int nbuf : { 1, 32 }) {
        std::vector<v4f *> bufs(nbuf);
        for (auto &p : bufs) { hipMalloc(&p, bytes); hipMemset(p, 0, bytes); }
        const int reps = nbuf == 1 ? 128 : 8 * nbuf;
        printf("== %d buffer(s), %.1f MB footprint\n", nbuf, bytes * nbuf / 1e6);
        for (int delay : { 0, 300 }) {
#define ROW(P, S) { const float us = run<P, S>(bufs, n, f4, delay, reps); \
            printf("dead time %d us, %-10s, %-42s: %6.2f us per launch, %.2f TB/s\n", delay / 100, pol(P), shape[S], us, bytes / (us * 1e-6) / 1e12); fflush(stdout); }
            ROW(0, 0) ROW(16, 0) ROW(2, 0) ROW(17, 0) ROW(18, 0) ROW(1, 0) ROW(19, 0)
            ROW(0, 1) ROW(16, 1) ROW(2, 1) ROW(17, 1)
            ROW(0, 2) ROW(16, 2)
        }
        for (int mode : { 1, 2, 4 }) {
            const float us = run<16, 1>(bufs, n, f4, mode == 4 ? 300 : 0, reps, mode);
            printf("sc1, one aligned stream, mode %d (1 zeros, 2 lane-dependent data, 4 dead time staggered 0..3 us): %6.2f us per launch\n", mode, us);
        }
        for (auto p : bufs) hipFree(p);
    }
    return 0;
}
