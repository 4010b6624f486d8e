// inv_optim.hip -- the scalar tail of the PPO update on the device:
//   inv_ppo_loss        ratio / clipped surrogate / value MSE / entropy of one minibatch AND their gradients with respect
//                       to the logits and values, one kernel (inversus_rl/ppo_agent.py:211-224 run under autograd there:
//                       ~70 small torch kernels per minibatch)
//   inv_clip_adam_step  global gradient norm + clip_grad_norm_(0.5) + Adam over ONE flat f32 parameter buffer
//                       (ppo_agent.py:227-232; torch: ~30 multi-tensor launches over 28 tensors)
// Plain f32 VALU work, HBM-bound: the optimizer step reads p, g, m, v and writes p, m, v once (7 x 41 MB at 15 x 10).
// -ffp-contract stays off for this file: the formulas are compared with torch's to 1e-6.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/inversus_hip.h"
#include "inv_internal.h"
#include "inv_device.h"                                   // inv::philox4x32

namespace {

constexpr int kActions = 13;
constexpr int kLossThreads = 256;
constexpr int kNormBlocks = 1024, kNormThreads = 256;

__device__ __forceinline__ float block_sum(float x, float *sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = x;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    __syncthreads();
    return t;
}

// One thread per sample.  Forward (ppo_agent.py:213-224):
//   logp = log_softmax(logits), new_logp = logp[a], entropy = -sum p logp
//   ratio = exp(new_logp - old_logp), policy = -min(ratio A, clamp(ratio, 1-c, 1+c) A), value = (v - R)^2
//   loss = mean(policy) + value_coef mean(value) - entropy_coef mean(entropy)
// Backward of `loss` (what autograd computes for the same expression):
//   d policy / d new_logp = -A ratio  if the unclipped term is the minimum (or the ratio is inside the clip range), else 0
//   d new_logp / d logit_j = [j == a] - p_j        d entropy / d logit_j = -p_j (logp_j + entropy)
//   d value / d v = 2 (v - R)
__global__ __launch_bounds__(kLossThreads) void k_ppo_loss(const float *__restrict__ logits, const float *__restrict__ values,
                                                            const int64_t *__restrict__ actions, const float *__restrict__ old_logp,
                                                            const float *__restrict__ adv, const float *__restrict__ ret, int B,
                                                            float clip, float value_coef, float entropy_coef,
                                                            float *__restrict__ dlogits, float *__restrict__ dvalues,
                                                            float *__restrict__ partial /* [gridDim.x][3] */)
{
    __shared__ float sh[kLossThreads / 64];
    const int i = blockIdx.x * kLossThreads + threadIdx.x;
    float pol = 0.0f, val = 0.0f, ent = 0.0f;
    if (i < B) {
        float z[kActions];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < kActions; ++j) { z[j] = logits[(size_t)i * kActions + j]; mx = fmaxf(mx, z[j]); }
        float se = 0.0f;
#pragma unroll
        for (int j = 0; j < kActions; ++j) se += expf(z[j] - mx);
        const float lse = mx + logf(se);
        const int a = (int)actions[i];
        float p[kActions], lp[kActions];
        float new_logp = 0.0f;
#pragma unroll
        for (int j = 0; j < kActions; ++j) {
            lp[j] = z[j] - lse;
            p[j] = expf(lp[j]);
            ent -= p[j] * lp[j];
            if (j == a) new_logp = lp[j];
        }
        const float A = adv[i];
        const float ratio = expf(new_logp - old_logp[i]);
        const float clamped = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
        const float s1 = ratio * A, s2 = clamped * A;
        pol = -fminf(s1, s2);
        const float dv = values[i] - ret[i];
        val = dv * dv;
        const float inv_b = 1.0f / (float)B;
        // gradient of -min(s1, s2): through s1 when it is the minimum; when the ratio is unclipped s1 == s2 and both carry A ratio
        const bool inside = ratio >= 1.0f - clip && ratio <= 1.0f + clip;
        const float g_logp = (inside || s1 < s2) ? -A * ratio : 0.0f;
#pragma unroll
        for (int j = 0; j < kActions; ++j) {
            const float d_new = (j == a ? 1.0f : 0.0f) - p[j];
            const float d_ent = -p[j] * (lp[j] + ent);
            dlogits[(size_t)i * kActions + j] = inv_b * (g_logp * d_new - entropy_coef * d_ent);
        }
        dvalues[i] = inv_b * value_coef * 2.0f * dv;
    }
    const float sp = block_sum(pol, sh), svv = block_sum(val, sh), sen = block_sum(ent, sh);
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 3 + 0] = sp;
        partial[blockIdx.x * 3 + 1] = svv;
        partial[blockIdx.x * 3 + 2] = sen;
    }
}

// Categorical(logits).sample() and .log_prob(sample) of PPOAgent.act (ppo_agent.py:96-100) as one kernel, one thread per sample:
// log-softmax with the SAME expressions as k_ppo_loss above (so exp(new_logp - old_logp) is exactly 1 when the update's first
// forward reproduces the rollout's logits), then the inverse CDF at one Philox4x32-10 uniform keyed by (seed words in device
// memory, sample index).  torch.distributions + multinomial were ~25 launches and 0.15 ms per policy and vector step.
__global__ __launch_bounds__(256) void k_sample_actions(const float *__restrict__ logits, int B, const int64_t *__restrict__ seed,
                                                         int64_t *__restrict__ actions, uint8_t *__restrict__ actions_u8, float *__restrict__ log_prob)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    float z[kActions];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < kActions; ++j) { z[j] = logits[(size_t)i * kActions + j]; mx = fmaxf(mx, z[j]); }
    float se = 0.0f;
#pragma unroll
    for (int j = 0; j < kActions; ++j) se += expf(z[j] - mx);
    const float lse = mx + logf(se);
    const uint64_t s0 = (uint64_t)seed[0], s1 = (uint64_t)seed[1];
    const inv::U4 r = inv::philox4x32((uint32_t)i, (uint32_t)(s1 >> 32), (uint32_t)s1, 0x53414d50u /* "SAMP" */, (uint32_t)s0, (uint32_t)(s0 >> 32));
    const float u = (float)(r.x >> 8) * (1.0f / 16777216.0f);            // [0, 1) on a 2^-24 grid
    float total = 0.0f, p[kActions];
#pragma unroll
    for (int j = 0; j < kActions; ++j) { p[j] = expf(z[j] - lse); total += p[j]; }
    const float target = u * total;                                       // total = 1 up to rounding: scale instead of trusting it
    int a = kActions - 1;
    float cum = 0.0f, lp = z[kActions - 1] - lse;
    bool found = false;
#pragma unroll
    for (int j = 0; j < kActions; ++j) {
        cum += p[j];
        if (!found && cum > target) { a = j; lp = z[j] - lse; found = true; }
    }
    if (actions) actions[i] = a;
    if (actions_u8) actions_u8[i] = (uint8_t)a;
    if (log_prob) log_prob[i] = lp;
}

// sum of squares of the flat gradient: kNormBlocks partial sums (f32 per thread over a grid-stride of float4, f64 across threads)
__global__ __launch_bounds__(kNormThreads) void k_sumsq_partial(const float *__restrict__ g, long long n, double *__restrict__ partial)
{
    __shared__ double sh[kNormThreads / 64];
    const long long n4 = n >> 2;
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    float acc = 0.0f;
    for (long long q = (long long)blockIdx.x * kNormThreads + threadIdx.x; q < n4; q += (long long)gridDim.x * kNormThreads) {
        const float4 v = g4[q];
        acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; acc += v * v; }
    double d = (double)acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kNormThreads / 64; ++w) t += sh[w];
        partial[blockIdx.x] = t;
    }
}

// clip_grad_norm_ + torch.optim.Adam (no weight decay, no amsgrad) in one pass over the flat buffers.
//   total = sqrt(sum g^2);  coef = min(1, max_norm / (total + 1e-6));  g' = coef g
//   m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// Every block re-adds the kNormBlocks partial sums in the same order (deterministic, no extra launch, no host round trip).
__global__ __launch_bounds__(256) void k_clip_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, long long n, const double *__restrict__ partial,
                                                   float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float max_norm, float grad_scale, float *__restrict__ norm_out)
{
    __shared__ float s_coef;
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int b = 0; b < kNormBlocks; ++b) t += partial[b];
        const float total = (float)sqrt(t) * fabsf(grad_scale);
        float coef = max_norm > 0.0f ? max_norm / (total + 1e-6f) : 1.0f;
        coef = coef < 1.0f ? coef : 1.0f;
        s_coef = coef * grad_scale;
        if (blockIdx.x == 0 && norm_out) *norm_out = total;
    }
    __syncthreads();
    const float coef = s_coef;
    const float step = lr / bc1;
    const long long n4 = n >> 2;
    float4 *p4 = reinterpret_cast<float4 *>(p), *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
        float4 pp = p4[q], mm = m4[q], vv = v4[q];
        const float4 gg = g4[q];
        float *pe = &pp.x, *me = &mm.x, *ve = &vv.x;
        const float *ge = &gg.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gc = ge[e] * coef;
            me[e] = b1 * me[e] + (1.0f - b1) * gc;
            ve[e] = b2 * ve[e] + (1.0f - b2) * gc * gc;
            pe[e] -= step * (me[e] / (sqrtf(ve[e]) / bc2_sqrt + eps));
        }
        p4[q] = pp; m4[q] = mm; v4[q] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
        const long long e = (n4 << 2) + threadIdx.x;
        const float gc = g[e] * coef;
        const float mn = b1 * m[e] + (1.0f - b1) * gc, vn = b2 * v[e] + (1.0f - b2) * gc * gc;
        m[e] = mn; v[e] = vn;
        p[e] -= step * (mn / (sqrtf(vn) / bc2_sqrt + eps));
    }
}

}  // namespace

extern "C" int32_t inv_ppo_loss_blocks(int32_t B) { return B > 0 ? (B + kLossThreads - 1) / kLossThreads : 0; }

extern "C" int inv_ppo_loss(const float *logits, const float *values, const int64_t *actions, const float *old_log_probs,
                            const float *advantages, const float *returns, int32_t B, float clip_ratio, float value_coef,
                            float entropy_coef, float *dlogits, float *dvalues, float *partial_sums, void *stream)
{
    if (!logits || !values || !actions || !old_log_probs || !advantages || !returns || !dlogits || !dvalues || !partial_sums || B < 1)
        return inv_fail(INV_ERR_BAD_ARG, "inv_ppo_loss: NULL tensor or B < 1");
    hipLaunchKernelGGL(k_ppo_loss, dim3(inv_ppo_loss_blocks(B)), dim3(kLossThreads), 0, (hipStream_t)stream, logits, values, actions,
                       old_log_probs, advantages, returns, (int)B, clip_ratio, value_coef, entropy_coef, dlogits, dvalues, partial_sums);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? INV_OK : inv_fail(INV_ERR_HIP, "inv_ppo_loss: %s", hipGetErrorString(e));
}

extern "C" int64_t inv_clip_adam_workspace_bytes(void) { return (int64_t)kNormBlocks * 8 + 8; }

extern "C" int inv_clip_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                                  float beta1, float beta2, float eps, int32_t step, float max_grad_norm, float grad_scale,
                                  void *workspace, void *stream)
{
    if (!params || !grads || !exp_avg || !exp_avg_sq || !workspace || n < 1 || step < 1)
        return inv_fail(INV_ERR_BAD_ARG, "inv_clip_adam_step: NULL buffer, n < 1 or step < 1");
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15u)
        return inv_fail(INV_ERR_BAD_ARG, "inv_clip_adam_step: buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    double *partial = reinterpret_cast<double *>(workspace);
    float *norm_out = reinterpret_cast<float *>(partial + kNormBlocks);
    hipLaunchKernelGGL(k_sumsq_partial, dim3(kNormBlocks), dim3(kNormThreads), 0, st, grads, (long long)n, partial);
    // bias corrections in f64 on the host, like torch (1 - beta ** step)
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const int blocks = (int)((n / 4 + 255) / 256 < 4096 ? ((n / 4 + 255) / 256 > 0 ? (n / 4 + 255) / 256 : 1) : 4096);
    hipLaunchKernelGGL(k_clip_adam, dim3(blocks), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, (long long)n, partial, lr, beta1,
                       beta2, eps, (float)bc1, (float)sqrt(bc2), max_grad_norm, grad_scale, norm_out);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? INV_OK : inv_fail(INV_ERR_HIP, "inv_clip_adam_step: %s", hipGetErrorString(e));
}

extern "C" int inv_sample_actions(const float *logits, int32_t B, const int64_t *seed, int64_t *actions, uint8_t *actions_u8, float *log_prob,
                                  void *stream)
{
    if (!logits || !seed || B < 0 || (!actions && !actions_u8))
        return inv_fail(INV_ERR_BAD_ARG, "inv_sample_actions: NULL logits / seed, B < 0, or neither action output");
    if (B == 0) return INV_OK;
    hipLaunchKernelGGL(k_sample_actions, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, (int)B, seed, actions, actions_u8, log_prob);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return inv_fail(INV_ERR_HIP, "inv_sample_actions: %s", hipGetErrorString(e));
    return INV_OK;
}
