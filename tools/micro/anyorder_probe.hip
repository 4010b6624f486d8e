// Does hipExtAnyOrderLaunch clear the AQL barrier bit on gfx950?  Two kernels of 8 small workgroups that each hold their CU for
// ~100 us (bounded by the 100 MHz wall clock: every wave leaves after `ticks`), launched back to back on one stream:
//   in order:  kernel B starts after kernel A has ended (stamps: B.start >= A.end)
//   any order: B's workgroups start while A's still run (B.start ~ A.start)
// Build: hipcc -O2 --offload-arch=gfx950 tools/micro/anyorder_probe.hip -o tools/micro/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdint.h>

__global__ void k_hold(uint64_t *stamps, int slot, uint64_t ticks)
{
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
    if (threadIdx.x == 0) {
        atomicMin((unsigned long long *)&stamps[2 * slot], (unsigned long long)t0);
        atomicMax((unsigned long long *)&stamps[2 * slot + 1], (unsigned long long)wall_clock64());
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main()
{
    uint64_t *d, h[8];
    CK(hipMalloc(&d, sizeof h));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 4; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
            CK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
            CK(hipDeviceSynchronize());
            for (int k = 0; k < 4; ++k) {
                if (mode == 0) hipLaunchKernelGGL(k_hold, dim3(8), dim3(64), 0, st, d, k, (uint64_t)10000);
                else hipExtLaunchKernelGGL(k_hold, dim3(8), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, d, k, (uint64_t)10000);
            }
            CK(hipGetLastError());
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
            printf("%s rep %d:", mode ? "any-order" : "in-order ", rep);
            for (int k = 0; k < 4; ++k) printf("  k%d [%6.1f, %6.1f] us", k, (double)(h[2 * k] - h[0]) / 100.0, (double)(h[2 * k + 1] - h[0]) / 100.0);
            printf("\n");
        }
    }
    return 0;
}
