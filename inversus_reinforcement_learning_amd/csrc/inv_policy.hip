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

// bf16 pipeline, not bit-comparable with the f32 module anyway: let the epilogue use FMAs even though the
// library is built with -ffp-contract=off for the env / GAE kernels
#pragma clang fp contract(fast)

namespace pol {

constexpr int H = 10, W = 15, P = H * W;                 // 150 pixels
constexpr int PW = W + 2, PH = H + 2, NPOS = PW * PH;    // 12 x 17 haloed tile, 204 positions
constexpr int OBS_WORDS = (12 * W * H + 31) / 32;          // 57 u32 words of one packed observation
constexpr int CSTRIDE = 136;                             // bf16 per position: 128 channels + 8 pad = 272 B
constexpr int THREADS = 256, WAVES = THREADS / 64;       // 4 waves per block, 2 blocks per CU (55 KB LDS each)
constexpr float LN_EPS = 1e-5f;

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

__device__ __forceinline__ int pos_of(int p) { return (p / W + 1) * PW + (p % W) + 1; }

// Diagnostic build only (-DINV_POLICY_STAMPS, never the shipped library): per-phase cycle shares of wave 0,
// accumulated into a buffer of its own that no kernel code reads (cdna_hip_programming.md, "In-kernel stamps").
#ifdef INV_POLICY_STAMPS
__device__ unsigned long long g_stamps[16];
__device__ __forceinline__ unsigned long long stamp_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define STAMP(slot)                                                                          \
    do {                                                                                     \
        const unsigned long long now_ = stamp_now();                                         \
        if (threadIdx.x == 0) atomicAdd(&g_stamps[slot], now_ - stamp_prev);                 \
        stamp_prev = now_;                                                                   \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// One conv3x3 (pad 1) + bias (+ residual) + LayerNorm over the whole sample + ReLU, as an implicit GEMM on
// v_mfma_f32_16x16x32_bf16:  D[cout][pixel] += W[cout][k] . X[k][pixel],  k-step = (tap, 32 channels).
//   A fragment (weights, global, pre-packed):  lane l holds W[16*mt + (l&15)][32*ks + 8*(l>>4) + j]
//   B fragment (activations, LDS):             lane l holds X[channel 32*kc + 8*(l>>4) + j][pixel 16*nt + (l&15)]
//                                              = 16 contiguous bytes of the channels-last tile
//   accumulator:                               reg q of lane l = D[16*mt + 4*(l>>4) + q][16*nt + (l&15)]
// Work split over the 8 waves: MW row blocks x NW pixel tiles per wave (150 pixels = 10 tiles of 16, so the
// two halves of the workgroup take 5 tiles each -- no ragged tile, no predication in the main loop).
// The k loop is software-pipelined by hand: the weight fragments of tap t+1 and the activation fragments of
// k-step s+1 are in flight while the MFMAs of k-step s issue (hipcc does not do this on its own: the first
// version waited on every ds_read and on the L2 load of the next weights inside each k-step).
template <int CIN, int COUT, int MW, int NW, bool RAGGED, bool RESIDUAL>
__device__ __forceinline__ void conv_ln_relu(const inv_policy_layer &lp, bf16 *tile, float *red, int lane, int m0, int n0, int ncount,
                                             unsigned long long &stamp_prev, int layer)
{
    constexpr int KPT = CIN / 32, KS = 9 * KPT;
    const int c16 = lane & 15, kg = lane >> 4;

    int pix[NW], base[NW], tap0[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        pix[i] = (n0 + i) * 16 + c16;
        base[i] = pos_of(min(pix[i], P - 1)) * CSTRIDE;             // columns >= 150 read pixel 149; never stored
        tap0[i] = base[i] - (PW + 1) * CSTRIDE + 8 * kg;            // fragment address of tap (0,0), channel group 0
    }

    f32x4 acc[MW][NW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
        for (int i = 0; i < NW; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[mi][i][q] = 0.0f;

    const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(lp.w) + (size_t)m0 * KS * 64 + lane;
#ifndef POL_AD
#define POL_AD 3
#endif
    constexpr int AD = POL_AD;                                            // weight fragments are requested AD k-steps ahead (L2 latency)
    bf16x8 a[AD + 1][MW], b[NW];
#pragma unroll
    for (int d = 0; d < AD; ++d)
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) a[d][mi] = wp[((size_t)mi * KS + (d < KS ? d : KS - 1)) * 64];
#pragma unroll
    for (int i = 0; i < NW; ++i) b[i] = *reinterpret_cast<const bf16x8 *>(tile + tap0[i]);
    // fully unrolled: every tap / channel-group offset is an immediate of the ds_read / global_load.
    // Software pipeline in place: the weight fragments of k-step s+AD are requested at the top of k-step s
    // (ring of AD+1 register sets), and each activation fragment is re-loaded for k-step s+1 right after its
    // last MFMA of k-step s (one k-step = MW*NW MFMAs of latency cover, no second fragment buffer).
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int ksn = ks + 1 < KS ? ks + 1 : ks;                 // redundant prefetches at the end stay in bounds
        const int ksa = ks + AD < KS ? ks + AD : KS - 1;
        const int tapn = ksn / KPT, kcn = ksn - tapn * KPT;
        const int off_n = ((tapn / 3) * PW + (tapn % 3)) * CSTRIDE + kcn * 32;
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) a[(ks + AD) % (AD + 1)][mi] = wp[((size_t)mi * KS + ksa) * 64];
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (!RAGGED || i < ncount) {
#pragma unroll
                for (int mi = 0; mi < MW; ++mi)
                    acc[mi][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks % (AD + 1)][mi], b[i], acc[mi][i], 0, 0, 0);
            }
            b[i] = *reinterpret_cast<const bf16x8 *>(tile + tap0[i] + off_n);
        }
    }

    STAMP(1 + layer);                                                // GEMM
    // bias (+ residual) and the LayerNorm statistics
    float s = 0.0f, ss = 0.0f;
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        const int ch0 = (m0 + mi) * 16 + 4 * kg;
        const float4 bq = *reinterpret_cast<const float4 *>(lp.bias + ch0);
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            float x0 = acc[mi][i][0] + bq.x, x1 = acc[mi][i][1] + bq.y, x2 = acc[mi][i][2] + bq.z, x3 = acc[mi][i][3] + bq.w;
            if (RESIDUAL) {
                const bf16x4 rq = *reinterpret_cast<const bf16x4 *>(tile + base[i] + ch0);
                x0 += (float)rq[0]; x1 += (float)rq[1]; x2 += (float)rq[2]; x3 += (float)rq[3];
            }
            acc[mi][i][0] = x0; acc[mi][i][1] = x1; acc[mi][i][2] = x2; acc[mi][i][3] = x3;
            if (i < ncount && pix[i] < P) {
                s += (x0 + x1) + (x2 + x3);
                ss += (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
            }
        }
    }
    // LayerNorm parameters travel in groups of G tiles (one L2 round trip per group, 2 x G float4 registers per
    // lane).  POL_LNPIPE: group g+1 is requested before group g is consumed, and group 0 before the statistics
    // reduction, so a wave sees one exposed round trip per layer instead of one per group.
    constexpr int G = NW < 5 ? NW : 5, GPM = NW / G, NG = MW * GPM;
    static_assert(NW % G == 0, "tiles per wave must be a multiple of the parameter group");
#ifdef POL_LNPIPE
    f16x8 ln[2][G];
#else
    f16x8 ln[1][G];
#endif
    // lp.ln: fp16 [pixel][COUT/4][8] = {weight x4, bias x4} of 4 consecutive channels: ONE 16-byte load per tile
    auto load_group = [&](int g, int buf) {
        const int mi = g / GPM, i0 = (g - mi * GPM) * G, ch0 = (m0 + mi) * 16 + 4 * kg;
#pragma unroll
        for (int j = 0; j < G; ++j)
            ln[buf][j] = *reinterpret_cast<const f16x8 *>(reinterpret_cast<const _Float16 *>(lp.ln) + ((size_t)min(pix[i0 + j], P - 1) * COUT + ch0) * 2);
    };
#if defined(POL_LNPIPE) || defined(POL_LNPRE)
    load_group(0, 0);
#endif
    s = wave_sum(s);
    ss = wave_sum(ss);
    const int wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave] = s; red[WAVES + wave] = ss; }
    __syncthreads();
    STAMP(5 + layer);                                                // statistics + barrier
    const float cnt = (float)(COUT * P);
    float S = 0.0f, SS = 0.0f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { S += red[w]; SS += red[WAVES + w]; }
    const float mean = S / cnt;
    const float var = fmaxf(SS / cnt - mean * mean, 0.0f);
    const float rstd = 1.0f / sqrtf(var + LN_EPS);
    const float shift = -mean * rstd;

#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int mi = g / GPM, i0 = (g - mi * GPM) * G, ch0 = (m0 + mi) * 16 + 4 * kg;
#ifdef POL_LNPIPE
        const int buf = g & 1;
        if (g + 1 < NG) load_group(g + 1, buf ^ 1);
#elif defined(POL_LNPRE)
        const int buf = 0;
        if (g > 0) load_group(g, 0);
#else
        const int buf = 0;
        load_group(g, 0);
#endif
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int i = i0 + j;
            if (i < ncount && pix[i] < P) {
                bf16x4 o;
                const f16x8 q = ln[buf][j];
                o[0] = (bf16)fmaxf(fmaf(fmaf(acc[mi][i][0], rstd, shift), (float)q[0], (float)q[4]), 0.0f);
                o[1] = (bf16)fmaxf(fmaf(fmaf(acc[mi][i][1], rstd, shift), (float)q[1], (float)q[5]), 0.0f);
                o[2] = (bf16)fmaxf(fmaf(fmaf(acc[mi][i][2], rstd, shift), (float)q[2], (float)q[6]), 0.0f);
                o[3] = (bf16)fmaxf(fmaf(fmaf(acc[mi][i][3], rstd, shift), (float)q[3], (float)q[7]), 0.0f);
                // IN PLACE: every read of the input tile by the GEMM happened before the statistics barrier
                // above, and the residual read of this lane's own elements precedes this write
                *reinterpret_cast<bf16x4 *>(tile + base[i] + ch0) = o;
            }
        }
        __builtin_amdgcn_sched_barrier(0);                         // at most two groups' parameter sets live at a time
    }
    __syncthreads();
    STAMP(9 + layer);                                                // epilogue + barrier
}

// BITS: the observation arrives as the packed bitstring inv_step writes under INV_STEP_OBS_BITS (57 words per sample).
template <bool BITS>
__global__ __launch_bounds__(THREADS, 2) void k_policy_trunk(inv_policy_params prm, const void *__restrict__ obs_in,
                                                          bf16 *__restrict__ feat, int N)
{
    __shared__ __attribute__((aligned(16))) bf16 tile[NPOS * CSTRIDE];         // 55 488 B: layers run in place -> 2 blocks / CU
    __shared__ float red[2 * WAVES];
    __shared__ uint32_t obw[BITS ? OBS_WORDS : 1];                             // BITS: the current sample's bitstring
    const float *obs = reinterpret_cast<const float *>(obs_in);
    const uint32_t *obits = reinterpret_cast<const uint32_t *>(obs_in);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)__builtin_amdgcn_readfirstlane((uint32_t)(tid >> 6));   // uniform -> tile ownership and weight pointers in SGPRs

    // zero both tiles once: the halo and the pad channels stay zero for the life of the block
    for (int i = tid; i < NPOS * CSTRIDE / 8; i += THREADS) {
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.0f;
        reinterpret_cast<bf16x8 *>(&tile[0])[i] = z;
    }
    if (BITS && tid < OBS_WORDS && (int)blockIdx.x < N) obw[tid] = obits[(size_t)blockIdx.x * OBS_WORDS + tid];
    __syncthreads();

    unsigned long long stamp_prev = 0;
#ifdef INV_POLICY_STAMPS
    stamp_prev = stamp_now();
#endif
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        // Per-lane addresses (LDS fragment bases, LayerNorm parameter pointers of all four layers: ~160
        // registers) are invariant over this loop; hipcc hoists them out and then spills them to scratch
        // around the MFMA loops.  Laundering the lane id makes them cheap per-sample recomputations instead.
        int lane_i = lane;
        asm volatile("" : "+v"(lane_i));
        // observation -> tile, channels-last, 32 channels (12..31 zero: the k-step is 32 wide).
        uint32_t next_word = 0u;
        if (BITS) {
            // the NEXT sample's 57 words are requested now and parked in a register until the end of this sample
            const int nn = n + (int)gridDim.x;
            if (tid < OBS_WORDS && nn < N) next_word = obits[(size_t)nn * OBS_WORDS + tid];
            // one 8-byte quad (4 channels) of a pixel per work item: quads 0..2 = channels 0..11 from the bitstring
            // (bit c*150 + p; bf16 1.0 = 0x3F80), quads 3..7 = the zero pad
            for (int z = tid; z < P * 8; z += THREADS) {
                const int p = z >> 3, part = z & 7;
                uint32_t lo = 0u, hi = 0u;
                if (part < 3) {
                    const int b0 = (4 * part) * P + p, b1 = b0 + P, b2 = b1 + P, b3 = b2 + P;
                    lo = (((obw[b0 >> 5] >> (b0 & 31)) & 1u) * 0x3F80u) | (((obw[b1 >> 5] >> (b1 & 31)) & 1u) * 0x3F800000u);
                    hi = (((obw[b2 >> 5] >> (b2 & 31)) & 1u) * 0x3F80u) | (((obw[b3 >> 5] >> (b3 & 31)) & 1u) * 0x3F800000u);
                }
                *reinterpret_cast<uint2 *>(&tile[pos_of(p) * CSTRIDE + part * 4]) = make_uint2(lo, hi);
            }
        } else {
            // f32 [12][150]: both float4 loads of a thread are issued before anything is converted (one HBM round trip, not 19).
            const float4 *o4 = reinterpret_cast<const float4 *>(obs + (size_t)n * 12 * P);     // 450 float4 per sample
            const int q0 = tid, q1 = tid + THREADS;
            const float4 v0 = o4[q0];
            float4 v1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q1 < 12 * P / 4) v1 = o4[q1];
            // pad channels 12..31 of every pixel: 40 B at byte offset 24 of the position = 5 x 8 B
            for (int z = tid; z < P * 5; z += THREADS) {
                const int p = z / 5, part = z - p * 5;
                bf16x4 zz;
                zz[0] = zz[1] = zz[2] = zz[3] = (bf16)0.0f;
                *reinterpret_cast<bf16x4 *>(&tile[pos_of(p) * CSTRIDE + 12 + part * 4]) = zz;
            }
            const float e0[4] = { v0.x, v0.y, v0.z, v0.w }, e1[4] = { v1.x, v1.y, v1.z, v1.w };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * q0 + j, c = e / P, p = e - c * P;
                tile[pos_of(p) * CSTRIDE + c] = (bf16)e0[j];
            }
            if (q1 < 12 * P / 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * q1 + j, c = e / P, p = e - c * P;
                    tile[pos_of(p) * CSTRIDE + c] = (bf16)e1[j];
                }
            }
        }
        __syncthreads();
        STAMP(0);                                                    // observation load + conversion
        // tile ownership (4 waves).  A weight fragment comes from L2 through the vector-memory path and is reused by
        // NW MFMAs, an activation fragment comes from LDS and is reused by MW; MW x NW = 20 accumulator tiles is what
        // the register file allows.  At 2 x 10 the LDS reads alone need the CU's full 128 B/clk while the MFMAs
        // issue; conv3/4 use 4 row blocks x 5 pixel tiles (half the LDS traffic, each weight fragment fetched by two
        // waves), which also leaves the register allocator enough room to keep the epilogue out of scratch.
        conv_ln_relu<32, 32, 1, 5, false, false>(prm.layer[0], tile, red, lane_i, wave & 1, (wave >> 1) * 5, 5, stamp_prev, 0);
#if defined(POL_C2_MW2)
        conv_ln_relu<32, 64, 2, 5, false, false>(prm.layer[1], tile, red, lane_i, (wave & 1) * 2, (wave >> 1) * 5, 5, stamp_prev, 1);
#else
        conv_ln_relu<32, 64, 1, 10, false, false>(prm.layer[1], tile, red, lane_i, wave, 0, 10, stamp_prev, 1);
#endif
        conv_ln_relu<64, 128, 4, 5, false, false>(prm.layer[2], tile, red, lane_i, (wave & 1) * 4, (wave >> 1) * 5, 5, stamp_prev, 2);
        conv_ln_relu<128, 128, 4, 5, false, true>(prm.layer[3], tile, red, lane_i, (wave & 1) * 4, (wave >> 1) * 5, 5, stamp_prev, 3);
        // features, channels-last [150][128] bf16 (the FC weights are permuted to this order on the host)
        bf16x8 *dst = reinterpret_cast<bf16x8 *>(feat + (size_t)n * P * 128);
        for (int q = tid; q < P * 16; q += THREADS) {
            const int p = q >> 4, part = q & 15;
            dst[q] = *reinterpret_cast<const bf16x8 *>(&tile[pos_of(p) * CSTRIDE + part * 8]);
        }
        if (BITS && tid < OBS_WORDS) obw[tid] = next_word;        // every read of obw happened before the first barrier of this sample
        __syncthreads();
        STAMP(13);                                                   // feature write-out
    }
}

}  // namespace pol

#ifdef INV_POLICY_STAMPS
extern "C" int inv_policy_debug_stamps(unsigned long long *out16_host, int reset)
{
    if (hipMemcpyFromSymbol(out16_host, HIP_SYMBOL(pol::g_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return INV_ERR_HIP;
    if (reset) {
        unsigned long long z[16] = { 0 };
        if (hipMemcpyToSymbol(HIP_SYMBOL(pol::g_stamps), z, sizeof z) != hipSuccess) return INV_ERR_HIP;
    }
    return INV_OK;
}
#endif

static int launch_trunk(const inv_policy_params *params, const void *obs, bool bits, void *feat_bf16, int32_t N, void *stream)
{
    if (!params || !obs || !feat_bf16 || N < 0) return INV_ERR_BAD_ARG;
    if (N == 0) return INV_OK;
    for (int l = 0; l < 4; ++l)
        if (!params->layer[l].w || !params->layer[l].bias || !params->layer[l].ln) return INV_ERR_BAD_ARG;
    static int cus = 0;                              // CU count, queried once (not in a captured region)
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else cus = 256;
    }
    const int grid = N < 2 * cus ? N : 2 * cus;     // two resident 256-thread workgroups per CU (55 KB of LDS each), persistent over samples
    if (bits)
        hipLaunchKernelGGL(pol::k_policy_trunk<true>, dim3(grid), dim3(pol::THREADS), 0, (hipStream_t)stream, *params, obs,
                           reinterpret_cast<pol::bf16 *>(feat_bf16), (int)N);
    else
        hipLaunchKernelGGL(pol::k_policy_trunk<false>, dim3(grid), dim3(pol::THREADS), 0, (hipStream_t)stream, *params, obs,
                           reinterpret_cast<pol::bf16 *>(feat_bf16), (int)N);
    return hipGetLastError() == hipSuccess ? INV_OK : INV_ERR_HIP;
}

extern "C" int inv_policy_trunk(const inv_policy_params *params, const float *obs, void *feat_bf16, int32_t N, void *stream)
{
    return launch_trunk(params, obs, false, feat_bf16, N, stream);
}

extern "C" int inv_policy_trunk_bits(const inv_policy_params *params, const uint32_t *bits, void *feat_bf16, int32_t N, void *stream)
{
    return launch_trunk(params, bits, true, feat_bf16, N, stream);
}
