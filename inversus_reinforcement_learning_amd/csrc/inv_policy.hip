// inv_policy.hip -- fused trunk of the reference's CNN policy (inversus_rl/policies.py:29-42, 91-99):
//   conv3x3(12->32) LN ReLU, conv3x3(32->64) LN ReLU, conv3x3(64->128) LN ReLU,
//   conv3x3(128->128) + residual, LN, ReLU
// for the 10 x 15 board, ONE workgroup per sample, everything between the f32 observation and the
// 19 200-feature vector kept in LDS.  SURVEY.md section 8f row N1.
//
// Why: with PyTorch-ROCm the trunk is 4 convs + 4 LayerNorm + 4 ReLU passes over up to 1.26 GB f32
// activations per 16 384-sample batch (19 ms, 80 TFLOP/s, activation-traffic bound); here the
// activations never leave the CU.  Convolutions are implicit GEMMs on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, f32 accumulate):  D[cout][pixel] = W[cout][k] . X[k][pixel],
// k = tap * Cin + c.  Activations live channels-last in a zero-haloed 12 x 17 LDS tile
// ([position][channel], 272-B position stride so that the 16-B fragment reads of 32 consecutive pixels
// spread over the banks); the B fragment of lane (pixel r, half h) for k-step (tap, 16-channel group)
// is one ds_read_b128 at position(r) + tap offset.  Weights are pre-packed on the host so that the A
// fragment of a lane is one coalesced 16-B global load (L2-resident: 486 KB bf16 for all four layers);
// a wave owns one 32-channel row block and all 5 pixel tiles, so each weight fragment is loaded once per
// sample and reused by 5 MFMAs.  The accumulator layout (column = pixel on the lane, rows = channels
// in registers, 4 consecutive channels per register quad) makes the LayerNorm epilogue write 8-B packed
// bf16 quads straight into the next layer's tile.  LayerNorm([C,H,W]) statistics are f32 sums over the
// whole sample (one cross-wave reduction per layer); affine parameters are pre-transposed to
// [pixel][channel].
//
// Numerics: bf16 operands, f32 accumulation, f32 LayerNorm, bf16 activations between layers -- NOT
// bit-comparable with the f32 reference policy; tests bound the difference (tests/test_gpu_policy.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/inversus_hip.h"

namespace pol {

constexpr int H = 10, W = 15, P = H * W;                 // 150 pixels
constexpr int PW = W + 2, PH = H + 2, NPOS = PW * PH;    // 12 x 17 haloed tile, 204 positions
constexpr int CSTRIDE = 136;                             // bf16 per position: 128 channels + 8 pad = 272 B
constexpr int NT = 5;                                    // pixel tiles of 32 (160 >= 150)
constexpr int THREADS = 256;
constexpr float LN_EPS = 1e-5f;

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ int pos_of(int p) { return (p / W + 1) * PW + (p % W) + 1; }

__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// One conv3x3 (pad 1) + bias (+ residual) + LayerNorm over the whole sample + ReLU.
// in / out: haloed channels-last LDS tiles; CIN is the padded input channel count (multiple of 16).
template <int CIN, int COUT, bool RESIDUAL>
__device__ __forceinline__ void conv_ln_relu(const inv_policy_layer &lp, const bf16 *__restrict__ in, bf16 *__restrict__ out,
                                             float *red, int lane, int wave)
{
    constexpr int MT = COUT / 32, KPT = CIN / 16, KS = 9 * KPT;
    const int r = lane & 31, h = lane >> 5;
    // tile ownership: one 32-channel row block per wave, as many pixel tiles as it takes
    int m, n0, nstride, ncount;
    if (MT == 4) { m = wave; n0 = 0; nstride = 1; ncount = 5; }
    else if (MT == 2) { m = wave & 1; n0 = (wave >> 1) * 3; nstride = 1; ncount = (wave >> 1) ? 2 : 3; }
    else { m = 0; n0 = wave; nstride = 4; ncount = wave == 0 ? 2 : 1; }

    int pix[NT], base[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        pix[i] = (n0 + i * nstride) * 32 + r;
        base[i] = pos_of(min(pix[i], P - 1)) * CSTRIDE + 8 * h;     // columns >= 150 read pixel 149; never stored
    }

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.0f;

    const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(lp.w) + (size_t)m * KS * 64 + lane;
    bf16x8 a_next = wp[0];
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const int toff = ((tap / 3 - 1) * PW + (tap % 3 - 1)) * CSTRIDE;
#pragma unroll 2
        for (int kc = 0; kc < KPT; ++kc) {
            const int ks = tap * KPT + kc;
            const bf16x8 a = a_next;
            if (ks + 1 < KS) a_next = wp[(size_t)(ks + 1) * 64];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                if (i < ncount) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8 *>(in + base[i] + toff + kc * 16);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                }
            }
        }
    }

    // bias (+ residual), LayerNorm statistics.  Register v of a lane: channel m*32 + 8*(v>>2) + 4*h + (v&3), pixel pix[i].
    float s = 0.0f, ss = 0.0f;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (i < ncount) {
            const bool valid = pix[i] < P;
            const int ipos = pos_of(min(pix[i], P - 1)) * CSTRIDE;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch0 = m * 32 + 8 * g + 4 * h;
                const float4 bq = *reinterpret_cast<const float4 *>(lp.bias + ch0);
                float x0 = acc[i][4 * g + 0] + bq.x, x1 = acc[i][4 * g + 1] + bq.y;
                float x2 = acc[i][4 * g + 2] + bq.z, x3 = acc[i][4 * g + 3] + bq.w;
                if (RESIDUAL) {
                    const bf16x4 rq = *reinterpret_cast<const bf16x4 *>(in + ipos + ch0);
                    x0 += (float)rq[0]; x1 += (float)rq[1]; x2 += (float)rq[2]; x3 += (float)rq[3];
                }
                acc[i][4 * g + 0] = x0; acc[i][4 * g + 1] = x1; acc[i][4 * g + 2] = x2; acc[i][4 * g + 3] = x3;
                if (valid) {
                    s += (x0 + x1) + (x2 + x3);
                    ss += (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
                }
            }
        }
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (lane == 0) { red[wave] = s; red[4 + wave] = ss; }
    __syncthreads();
    const float cnt = (float)(COUT * P);
    const float S = (red[0] + red[1]) + (red[2] + red[3]), SS = (red[4] + red[5]) + (red[6] + red[7]);
    const float mean = S / cnt;
    const float var = fmaxf(SS / cnt - mean * mean, 0.0f);
    const float rstd = 1.0f / sqrtf(var + LN_EPS);

#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (i < ncount && pix[i] < P) {
            const int opos = pos_of(pix[i]) * CSTRIDE;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch0 = m * 32 + 8 * g + 4 * h;
                const float4 wq = *reinterpret_cast<const float4 *>(lp.lnw + (size_t)pix[i] * COUT + ch0);
                const float4 bq = *reinterpret_cast<const float4 *>(lp.lnb + (size_t)pix[i] * COUT + ch0);
                bf16x4 o;
                o[0] = (bf16)fmaxf((acc[i][4 * g + 0] - mean) * rstd * wq.x + bq.x, 0.0f);
                o[1] = (bf16)fmaxf((acc[i][4 * g + 1] - mean) * rstd * wq.y + bq.y, 0.0f);
                o[2] = (bf16)fmaxf((acc[i][4 * g + 2] - mean) * rstd * wq.z + bq.z, 0.0f);
                o[3] = (bf16)fmaxf((acc[i][4 * g + 3] - mean) * rstd * wq.w + bq.w, 0.0f);
                *reinterpret_cast<bf16x4 *>(out + opos + ch0) = o;
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(THREADS) void k_policy_trunk(inv_policy_params prm, const float *__restrict__ obs,
                                                          bf16 *__restrict__ feat, int N)
{
    __shared__ __attribute__((aligned(16))) bf16 tile[2][NPOS * CSTRIDE];      // 2 x 55 488 B
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // zero both tiles once: the halo and the pad channels stay zero for the life of the block
    for (int i = tid; i < 2 * NPOS * CSTRIDE / 8; i += THREADS) {
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.0f;
        reinterpret_cast<bf16x8 *>(&tile[0][0])[i] = z;
    }
    __syncthreads();

    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        // observation f32 [12][150] -> tile 0, channels-last, 16 channels (12..15 zero)
        const float *o = obs + (size_t)n * 12 * P;
        for (int e = tid; e < 16 * P; e += THREADS) {
            const int c = e / P, p = e - c * P;
            const float val = c < 12 ? o[c * P + p] : 0.0f;
            tile[0][pos_of(p) * CSTRIDE + c] = (bf16)val;
        }
        __syncthreads();
        conv_ln_relu<16, 32, false>(prm.layer[0], tile[0], tile[1], red, lane, wave);
        conv_ln_relu<32, 64, false>(prm.layer[1], tile[1], tile[0], red, lane, wave);
        conv_ln_relu<64, 128, false>(prm.layer[2], tile[0], tile[1], red, lane, wave);
        conv_ln_relu<128, 128, true>(prm.layer[3], tile[1], tile[0], red, lane, wave);
        // features, channels-last [150][128] bf16 (the FC weights are permuted to this order on the host)
        bf16x8 *dst = reinterpret_cast<bf16x8 *>(feat + (size_t)n * P * 128);
        for (int q = tid; q < P * 16; q += THREADS) {
            const int p = q >> 4, part = q & 15;
            dst[q] = *reinterpret_cast<const bf16x8 *>(&tile[0][pos_of(p) * CSTRIDE + part * 8]);
        }
        __syncthreads();
    }
}

}  // namespace pol

extern "C" int inv_policy_trunk(const inv_policy_params *params, const float *obs, void *feat_bf16, int32_t N, void *stream)
{
    if (!params || !obs || !feat_bf16 || N < 0) return INV_ERR_BAD_ARG;
    if (N == 0) return INV_OK;
    for (int l = 0; l < 4; ++l)
        if (!params->layer[l].w || !params->layer[l].bias || !params->layer[l].lnw || !params->layer[l].lnb) return INV_ERR_BAD_ARG;
    static int cus = 0;                              // CU count, queried once (not in a captured region)
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else cus = 256;
    }
    const int grid = N < cus ? N : cus;             // one resident workgroup per CU (111 KB of LDS each), persistent over samples
    hipLaunchKernelGGL(pol::k_policy_trunk, dim3(grid), dim3(pol::THREADS), 0, (hipStream_t)stream, *params, obs,
                       reinterpret_cast<pol::bf16 *>(feat_bf16), (int)N);
    return hipGetLastError() == hipSuccess ? INV_OK : INV_ERR_HIP;
}
