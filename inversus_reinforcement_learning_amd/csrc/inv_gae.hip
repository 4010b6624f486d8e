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
//  * per_env: [T][N] buffers, one chain per env column; a 512-thread block owns 64 columns and splits
//    128-step tiles over its 8 waves by time, everything in registers (see k_gae_per_env).
//    HBM-bound in principle: 17 B per transition.
//  * reference_flat: the reference scans ONE chain over the env-interleaved buffer
//    (training.py:128-137 fill order), so V[k+1] is the next env at the same timestep.  A
//    `done` cuts the chain, so the buffer splits into independent segments, each ending at a
//    done (or at the buffer end); the lane that owns a segment end walks its segment
//    backwards.  Bit-identical to the sequential scan, parallel across segments.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

#include "../../include/inversus_hip.h"
#include "inv_internal.h"

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

// per_env scan.  Only N independent chains exist (one per env column of the [T][N] buffers) and a chain is sequential by
// contract (every step's roundings are the reference's), so one lane per chain alone means 256 waves at N = 16 384, one per CU,
// each bound by its own memory latency (measured 1.5-1.7 TB/s).  Round 1 staged 32-column x 128-step tiles through LDS (12.9 us
// for 35.7 MB: load / precompute / chain / write-back phases of all blocks in lockstep, 69 KB of LDS traffic per tile each way).
// Round 3 keeps everything in REGISTERS and splits a tile over the waves of a block by TIME:
//   * a block owns 64 env columns (lane = column: every global access is a 256-B row piece) and tiles of kGaeT = 128 steps;
//     wave w owns the tile's rows [16 w, 16 w + 16) (row k <-> time step thi - k, row 0 = the latest step);
//   * every wave first requests ALL its rows (r, V, done: 48 loads + the V row of the step after its slice) -- 8 waves x 49
//     loads = 100 KB in flight per CU -- and computes the chain-independent part of every step in parallel:
//         e = done ? r - V : (r + gamma*V') - V                        (same roundings as the reference, in order)
//   * the chain A = done ? e : e + gamma_lam*A' then runs through the waves in turn (wave 0 holds the latest steps): 16
//     dependent steps per wave, the carry handed on through one LDS word per column and a workgroup barrier;
//   * a wave streams its A and ret = A + V rows out AFTER handing the carry on, i.e. while the next wave runs its steps.
// T > 128 loops over tiles from the end; the carry stays in LDS.
// Round 4: COLS columns per block, 64 (lane = column) or 32 (small N).  With 32 the wave's 64 lanes are 32 columns x 2 SUB-SLICES of the
// wave's 16 rows (8 rows each; every global access a 128-B row piece): the chain passes through the sub-slices in turn, handed from one
// to the next by a lane shuffle -- still 16 dependent steps per wave -- and N envs make N / 32 blocks instead of N / 64: 128 at N = 4 096,
// where 64-column blocks left three CUs in four idle.  (16 columns x 4 sub-slices, 64-B pieces: never better than 32; tools builds only.)
constexpr int kGaeT = 128, kGaeWaves = 8, kGaeRows = kGaeT / kGaeWaves, kGaeThreads = kGaeWaves * 64;

template <int COLS>
__global__ __launch_bounds__(kGaeThreads) void k_gae_per_env(const float *__restrict__ r, const float *__restrict__ v,
                                                              const uint8_t *__restrict__ done, int T, int N, float gamma,
                                                              float gl, const float *__restrict__ last_value,
                                                              float *__restrict__ adv, float *__restrict__ ret)
{
    constexpr int SUB = 64 / COLS, RS = kGaeRows / SUB;         // sub-slices per wave, rows per sub-slice
    static_assert(COLS * SUB == 64 && RS * SUB == kGaeRows, "column / sub-slice split");
    __shared__ float carry[COLS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane % COLS, sub = lane / COLS;
    const int col = blockIdx.x * COLS + c;
    const bool col_ok = col < N;
    const int colc = col_ok ? col : N - 1;                      // clamped: out-of-range lanes load a valid column and never store
    if (w == 0 && sub == 0) carry[c] = 0.0f;                    // last_gae = 0 (ppo_agent.py:141)

    for (int thi = T - 1; thi >= 0; thi -= kGaeT) {
        const int rows = min(kGaeT, thi + 1);
        const int k0 = w * kGaeRows + sub * RS;                 // my first tile row (row k <-> time step thi - k)
        const int nr = max(0, min(RS, rows - k0));              // my rows in this tile
        float rr[RS], vv[RS], e[RS];
        uint32_t dd[RS];
        const int tn = thi - k0 + 1;                            // the step after my latest row (<= T; may be < 1 for an idle lane)
        const float *vnp = (tn < T || !last_value) ? v + (size_t)min(max(tn, 0), T - 1) * N + colc : last_value + colc;
        float vnext = *vnp;
        if (tn >= T && !last_value) vnext = 0.0f;
        // branch-free: rows beyond the tile's end (only in a last, partial tile) load step 0 again and are never used -- a load
        // inside a conditional block makes hipcc drain vmcnt at the block's end, i.e. one memory round trip per row
#pragma unroll
        for (int j = 0; j < RS; ++j) {
            const size_t idx = (size_t)max(thi - k0 - j, 0) * N + colc;
            rr[j] = r[idx]; vv[j] = v[idx]; dd[j] = done[idx];
        }
#pragma unroll
        for (int j = 0; j < RS; ++j) {
            const float vn = j == 0 ? vnext : vv[j - 1];
            const float gv = gamma * vn;
            const float sum = rr[j] + gv;
            const float d_nd = sum - vv[j];                     // (r + gamma*V') - V
            const float d_d = rr[j] - vv[j];                    // r - V
            e[j] = dd[j] ? d_d : d_nd;
        }
        __syncthreads();                                        // carry of the previous tile (or its initial 0) is visible
#pragma unroll 1
        for (int p = 0; p < kGaeWaves; ++p) {
            if (p * kGaeRows >= rows) break;                    // block-uniform
            if (w == p) {
                float last = carry[c];
#pragma unroll
                for (int q = 0; q < SUB; ++q) {                 // the chain runs through the sub-slices in turn (every lane computes,
                    float mine = last;                          // sub-slice q's result counts)
#pragma unroll
                    for (int j = 0; j < RS; ++j) {
                        const float g2 = gl * mine;
                        const float cand = e[j] + g2;
                        const float nl = dd[j] ? e[j] : cand;
                        mine = j < nr ? nl : mine;
                        if (SUB == 1 || sub == q) e[j] = mine;  // e becomes A
                    }
                    last = SUB == 1 ? mine : __shfl(mine, c + COLS * q);
                }
                if (sub == 0) carry[c] = last;
            }
            __syncthreads();
            if (w == p && col_ok) {
#pragma unroll
                for (int j = 0; j < RS; ++j) {
                    if (j < nr) {
                        const size_t idx = (size_t)(thi - k0 - j) * N + col;
                        adv[idx] = e[j]; ret[idx] = e[j] + vv[j];
                    }
                }
            }
        }
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
        // 64-column blocks (256-B row pieces) once they give at least every second CU a block; below that 32-column blocks (128-B
        // pieces, twice the blocks).  Measured, us per launch for 64 / 32 / 16 columns (profiles/r04_gae_cols_ab.txt):
        //   T = 128, N = 16 384: 8.24 / 8.53 / 13.9      T = 128, N = 131 072: 55.7 / 62.9 / 89.0
        //   T = 128, N =  4 096: 6.25 / 5.16 / 5.17      T = 512, N =   4 096: 20.0 / 16.3 / 19.3
        int cols = N >= 32 * inv_cu_count() ? 64 : 32;
#ifdef INV_TOOLS_BUILD
        if (const char *e = getenv("INV_GAE_COLS")) cols = atoi(e);                      // tools/gae_kernel_time.py A/B
#endif
        if (cols == 64)
            hipLaunchKernelGGL(k_gae_per_env<64>, dim3((N + 63) / 64), dim3(kGaeThreads), 0, st, r, v, done, T, N, gamma, gamma_lam, last_value, adv, ret);
#ifdef INV_TOOLS_BUILD
        else if (cols == 16)
            hipLaunchKernelGGL(k_gae_per_env<16>, dim3((N + 15) / 16), dim3(kGaeThreads), 0, st, r, v, done, T, N, gamma, gamma_lam, last_value, adv, ret);
#endif
        else
            hipLaunchKernelGGL(k_gae_per_env<32>, dim3((N + 31) / 32), dim3(kGaeThreads), 0, st, r, v, done, T, N, gamma, gamma_lam, last_value, adv, ret);
    } else if (layout == INV_GAE_REFERENCE_FLAT) {
        const long long n = (long long)T * N;
        hipLaunchKernelGGL(k_gae_flat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, r, v, done, n, gamma, gamma_lam, last_value, adv, ret);
    } else {
        return INV_ERR_BAD_ARG;
    }
    return hipGetLastError() == hipSuccess ? INV_OK : INV_ERR_HIP;
}
