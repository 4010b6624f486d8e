// inv_gae.hip -- GAE / return scan, PPOAgent.compute_advantages (inversus_rl/ppo_agent.py:127-157).
//
// Arithmetic contract (NumPy >= 2 scalar rules, verified against the reference): every
// operation is a separately rounded f32 op, in this order, with no FMA contraction
// (this file is compiled with -ffp-contract=off):
//     done:      A = r - V[k]
//     otherwise: A = ((r + gamma*V[k+1]) - V[k]) + gamma_lam*A_next
//     ret = A + V[k]
// The recurrence is sequential by nature; parallelism comes from independent chains.
//
//  * per_env: [T][N] buffers, one lane per env column -> loads/stores are coalesced across
//    the env batch; the T loop is chunked so that a chunk's loads are all in flight before
//    the dependent chain starts.  HBM-bound in principle (17 B per transition), latency-bound
//    at N = 16 384 (256 waves on 256 CUs).
//  * reference_flat: the reference scans ONE chain over the env-interleaved buffer
//    (training.py:128-137 fill order), so V[k+1] is the next env at the same timestep.  A
//    `done` cuts the chain, so the buffer splits into independent segments, each ending at a
//    done (or at the buffer end); the lane that owns a segment end walks its segment
//    backwards.  Bit-identical to the sequential scan, parallel across segments.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/inversus_hip.h"

namespace {

constexpr int kChunk = 32;

__device__ __forceinline__ float gae_step(float r, float v, float vnext, bool done, float gamma, float gl, float &last)
{
    if (done) {
        last = r - v;
    } else {
        const float gv = gamma * vnext;
        const float s = r + gv;
        const float delta = s - v;
        const float g2 = gl * last;
        last = delta + g2;
    }
    return last;
}

struct GaeChunk {
    float r[kChunk], v[kChunk];
    uint8_t d[kChunk];
};

// loads of time steps t, t-1, ..., t-kChunk+1 of column i (rows below 0 are clamped, never consumed)
__device__ __forceinline__ void gae_load(GaeChunk &c, const float *__restrict__ r, const float *__restrict__ v,
                                         const uint8_t *__restrict__ done, int t, int N, int i)
{
#pragma unroll
    for (int k = 0; k < kChunk; ++k) {
        const size_t idx = (size_t)max(t - k, 0) * N + i;
        c.r[k] = r[idx]; c.v[k] = v[idx]; c.d[k] = done[idx];
    }
}

// Only N lanes exist (one per env), i.e. 256 waves at N = 16 384 -- one per CU -- so the scan is bound by
// memory latency, not bandwidth: the T loop is software-pipelined, the 3 x 32 loads of chunk c+1 are
// in flight while the dependent chain of chunk c runs.
__global__ __launch_bounds__(64) void k_gae_per_env(const float *__restrict__ r, const float *__restrict__ v,
                                                     const uint8_t *__restrict__ done, int T, int N, float gamma, float gl,
                                                     const float *__restrict__ last_value, float *__restrict__ adv,
                                                     float *__restrict__ ret)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float vnext = last_value ? last_value[i] : 0.0f;
    float last = 0.0f;
    GaeChunk cur, nxt;
    gae_load(cur, r, v, done, T - 1, N, i);
    for (int t = T - 1; t >= 0; t -= kChunk) {
        if (t - kChunk >= 0) gae_load(nxt, r, v, done, t - kChunk, N, i);
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
            if (t - k >= 0) {
                const size_t idx = (size_t)(t - k) * N + i;
                const float a = gae_step(cur.r[k], cur.v[k], vnext, cur.d[k] != 0, gamma, gl, last);
                adv[idx] = a;
                ret[idx] = a + cur.v[k];
                vnext = cur.v[k];
            }
        }
        cur = nxt;
    }
}

__global__ __launch_bounds__(256) void k_gae_flat(const float *__restrict__ r, const float *__restrict__ v,
                                                   const uint8_t *__restrict__ done, long long n, float gamma, float gl,
                                                   const float *__restrict__ last_value, float *__restrict__ adv,
                                                   float *__restrict__ ret)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const bool is_end = done[e] != 0 || e == n - 1;
    if (!is_end) return;
    float vnext = (e == n - 1) ? (last_value ? last_value[0] : 0.0f) : 0.0f;   // only read when !done[e]
    float last = 0.0f;
    for (long long k = e; k >= 0; --k) {
        const bool dn = done[k] != 0;
        if (dn && k != e) break;                    // previous segment's end
        const float rv = r[k], vv = v[k];
        const float a = gae_step(rv, vv, vnext, dn, gamma, gl, last);
        adv[k] = a;
        ret[k] = a + vv;
        vnext = vv;
    }
}

}  // namespace

extern "C" int inv_gae(const float *r, const float *v, const uint8_t *done, int32_t T, int32_t N, float gamma,
                       float gamma_lam, const float *last_value, int32_t layout, float *adv, float *ret, void *stream)
{
    if (!r || !v || !done || !adv || !ret || T < 1 || N < 1) return INV_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (layout == INV_GAE_PER_ENV) {
        hipLaunchKernelGGL(k_gae_per_env, dim3((N + 63) / 64), dim3(64), 0, st, r, v, done, T, N, gamma, gamma_lam, last_value, adv, ret);
    } else if (layout == INV_GAE_REFERENCE_FLAT) {
        const long long n = (long long)T * N;
        hipLaunchKernelGGL(k_gae_flat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, r, v, done, n, gamma, gamma_lam, last_value, adv, ret);
    } else {
        return INV_ERR_BAD_ARG;
    }
    return hipGetLastError() == hipSuccess ? INV_OK : INV_ERR_HIP;
}
