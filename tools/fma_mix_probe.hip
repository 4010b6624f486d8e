// Probe (gfx950): are v_pk_fma_f32 (both lanes, with and without the op_sel_hi broadcast hipcc emits) and v_fma_mix_f32 the same single-rounding
// fma as v_fma_f32?  Operands: full-precision f32 (a 48-bit product that an unfused multiply-add would round).  Host check: glibc fmaf (exact).
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/exp/fma_mix_probe tools/fma_mix_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
typedef _Float16 f16;
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ void k(const float *x, const float *w, const float *b, float *r_fma, float *r_pk0, float *r_pk1, float *r_bc1, float *r_mix, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float xv = x[i], wf = w[i], bf = b[i];
    float a;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(xv), "v"(wf), "v"(bf));
    r_fma[i] = a;
    float2v X = {xv, xv}, W = {wf, wf}, B = {bf, bf}, P, Q;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(P) : "v"(X), "v"(W), "v"(B));
    r_pk0[i] = P[0]; r_pk1[i] = P[1];
    float2v W0 = {wf, 0.0f}, B0 = {bf, 0.0f};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(Q) : "v"(X), "v"(W0), "v"(B0));
    r_bc1[i] = Q[1];
    // f16 operands for the mix form: w, b rounded to f16 (host does the same)
    uint32_t wb = (uint32_t)__builtin_bit_cast(uint16_t, (f16)wf), bb = (uint32_t)__builtin_bit_cast(uint16_t, (f16)bf);
    float m;
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(m) : "v"(xv), "v"(wb), "v"(bb));
    r_mix[i] = m;
}
int main()
{
    const int n = 1 << 24;
    std::vector<float> x(n), w(n), b(n);
    srand(1);
    auto rnd = [] { return (float)((double)rand() / RAND_MAX * 2.0 - 1.0); };
    for (int i = 0; i < n; ++i) {
        x[i] = rnd() * 40.0f;                    // accumulator-like
        w[i] = 0.05f + 0.3f * fabsf(rnd());      // rstd-like
        b[i] = rnd() * 0.8f;                     // shift-like: cancellation with x * w happens
    }
    float *d[8];
    for (int j = 0; j < 8; ++j) (void)hipMalloc(&d[j], n * 4);
    (void)hipMemcpy(d[0], x.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d[1], w.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d[2], b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], n);
    std::vector<float> r[5];
    for (int j = 0; j < 5; ++j) { r[j].resize(n); (void)hipMemcpy(r[j].data(), d[3 + j], n * 4, hipMemcpyDeviceToHost); }
    long bad[5] = { 0, 0, 0, 0, 0 }; int shown = 0;
    for (int i = 0; i < n; ++i) {
        const float want = fmaf(x[i], w[i], b[i]);
        for (int j = 0; j < 4; ++j) if (r[j][i] != want) { ++bad[j]; if (shown++ < 6) printf("form %d: x=%a w=%a b=%a want=%a got=%a unfused=%a\n", j, x[i], w[i], b[i], want, r[j][i], x[i] * w[i] + b[i]); }
        if (r[4][i] != fmaf(x[i], (float)(f16)w[i], (float)(f16)b[i])) ++bad[4];
    }
    printf("n=%d wrong: v_fma_f32 %ld  v_pk_fma_f32.lo %ld  .hi %ld  .hi(op_sel_hi broadcast) %ld  v_fma_mix_f32 %ld\n", n, bad[0], bad[1], bad[2], bad[3], bad[4]);
    return 0;
}
