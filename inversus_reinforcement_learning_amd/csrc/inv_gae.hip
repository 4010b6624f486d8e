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
//  * per_env: [T][N] buffers, one chain per env column; tiles of 32 columns x 128 steps are
//    staged through LDS by a whole 256-thread block so that enough memory operations are in
//    flight (see k_gae_per_env).  HBM-bound in principle: 17 B per transition.
//  * reference_flat: the reference scans ONE chain over the env-interleaved buffer
//    (training.py:128-137 fill order), so V[k+1] is the next env at the same timestep.  A
//    `done` cuts the chain, so the buffer splits into independent segments, each ending at a
//    done (or at the buffer end); the lane that owns a segment end walks its segment
//    backwards.  Bit-identical to the sequential scan, parallel across segments.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/inversus_hip.h"

namespace {

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

// per_env scan, tiled through LDS.  Only N independent chains exist (one per env): with one lane per
// chain the kernel runs 256 waves at N = 16 384, one per CU, bound by that wave's memory latency and by
// the per-step dependent code (measured 1.5-1.7 TB/s).  So a 256-thread block owns a tile of
// kGaeE = 32 env columns x kGaeT = 128 time steps and splits the work by what is truly sequential:
//   1. all four waves stream r, V (plus the V row of step t+1) and done into LDS (coalesced 128-B rows);
//   2. all threads compute, in parallel, the part of every step that does not depend on the chain:
//        e = done ? r - V : (r + gamma*V') - V            (same roundings as the reference, in order)
//   3. the first 32 lanes run the chains: A = done ? e : e + gamma_lam*A'  -- one multiply, one add and
//      one select per step, LDS reads unrolled ahead of the dependent ops;
//   4. all threads form ret = A + V and stream A, ret back.
// ~69 KB of LDS per block -> two blocks per CU; T > 128 loops over tiles from the end carrying A'.
constexpr int kGaeT = 128, kGaeE = 32, kGaeThreads = 256, kGaeRowsPerPass = kGaeThreads / kGaeE;

__global__ __launch_bounds__(kGaeThreads) void k_gae_per_env(const float *__restrict__ r, const float *__restrict__ v,
                                                              const uint8_t *__restrict__ done, int T, int N, float gamma,
                                                              float gl, const float *__restrict__ last_value,
                                                              float *__restrict__ adv, float *__restrict__ ret)
{
    __shared__ float se[kGaeT][kGaeE];           // r, then e (step 2), then A (step 3)
    __shared__ float sv[kGaeT + 1][kGaeE];       // sv[k + 1] = V at tile row k; sv[0] = V of the step after the tile
    __shared__ uint8_t sd[kGaeT][kGaeE];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * kGaeE;
    const int c = tid & (kGaeE - 1), k0 = tid / kGaeE;           // this thread's column and first tile row
    const bool col_ok = i0 + c < N;
    const bool chain = tid < kGaeE && col_ok;
    float last = 0.0f;

    for (int thi = T - 1; thi >= 0; thi -= kGaeT) {
        const int rows = min(kGaeT, thi + 1);                 // tile row k <-> time step thi - k
        if (col_ok) {
            if (k0 == 0) sv[0][c] = (thi + 1 < T) ? v[(size_t)(thi + 1) * N + i0 + c] : (last_value ? last_value[i0 + c] : 0.0f);
#pragma unroll 8
            for (int k = k0; k < rows; k += kGaeRowsPerPass) {
                const size_t idx = (size_t)(thi - k) * N + i0 + c;
                se[k][c] = r[idx]; sv[k + 1][c] = v[idx]; sd[k][c] = done[idx];
            }
        }
        __syncthreads();
        if (col_ok) {
#pragma unroll 8
            for (int k = k0; k < rows; k += kGaeRowsPerPass) {
                const float rr = se[k][c], vv = sv[k + 1][c], vn = sv[k][c];
                const float gv = gamma * vn;
                const float sum = rr + gv;
                const float d_nd = sum - vv;                  // (r + gamma*V') - V
                const float d_d = rr - vv;                    // r - V
                se[k][c] = sd[k][c] ? d_d : d_nd;
            }
        }
        __syncthreads();
        if (chain) {
#pragma unroll 16
            for (int k = 0; k < rows; ++k) {
                const float e = se[k][tid];
                const float g2 = gl * last;
                const float cand = e + g2;
                last = sd[k][tid] ? e : cand;
                se[k][tid] = last;
            }
        }
        __syncthreads();
        if (col_ok) {
#pragma unroll 8
            for (int k = k0; k < rows; k += kGaeRowsPerPass) {
                const size_t idx = (size_t)(thi - k) * N + i0 + c;
                const float a = se[k][c];
                adv[idx] = a; ret[idx] = a + sv[k + 1][c];
            }
        }
        __syncthreads();                                      // se / sv / sd are rewritten by the next tile
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
        hipLaunchKernelGGL(k_gae_per_env, dim3((N + kGaeE - 1) / kGaeE), dim3(kGaeThreads), 0, st, r, v, done, T, N, gamma, gamma_lam, last_value, adv, ret);
    } else if (layout == INV_GAE_REFERENCE_FLAT) {
        const long long n = (long long)T * N;
        hipLaunchKernelGGL(k_gae_flat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, r, v, done, n, gamma, gamma_lam, last_value, adv, ret);
    } else {
        return INV_ERR_BAD_ARG;
    }
    return hipGetLastError() == hipSuccess ? INV_OK : INV_ERR_HIP;
}
