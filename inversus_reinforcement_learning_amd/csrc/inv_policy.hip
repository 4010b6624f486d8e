// inv_policy.hip -- fused trunk of the reference's CNN policy (inversus_rl/policies.py:29-42, 91-99):
//   conv3x3(12->32) LN ReLU, conv3x3(32->64) LN ReLU, conv3x3(64->128) LN ReLU,
//   conv3x3(128->128) + residual, LN, ReLU
// for the 10 x 15 board, ONE workgroup per sample, everything between the f32 observation and the
// 19 200-feature vector kept in LDS.  SURVEY.md section 8f row N1.
//
// Why: with PyTorch-ROCm the trunk is 4 convs + 4 LayerNorm + 4 ReLU passes over up to 1.26 GB f32
// activations per 16 384-sample batch (19 ms, 80 TFLOP/s, activation-traffic bound); here the
// activations never leave the CU.  Convolutions are implicit GEMMs on the bf16 matrix cores
// (v_mfma_f32_16x16x32_bf16, f32 accumulate):  D[cout][pixel] = W[cout][k] . X[k][pixel],
// k = tap * Cin + c.  Activations live channels-last in a zero-haloed 12 x 17 LDS tile
// ([position][channel], 272-B position stride so that the 16-B fragment reads of 16 consecutive pixels
// spread over the banks); the B fragment of lane (pixel r, k-group g) for k-step (tap, 32-channel group)
// is one ds_read_b128 at position(r) + tap offset.  Weights are pre-packed on the host so that the A
// fragment of a lane is one coalesced 16-B global load (L2-resident: 486 KB bf16 for all four layers).
// The accumulator layout (column = pixel on the lane, 4 consecutive channels per register quad) makes the
// LayerNorm epilogue write 8-B packed bf16 quads straight into the next layer's tile.  LayerNorm([C,H,W])
// statistics are f32 sums over the whole sample (one cross-wave reduction per layer); affine parameters are
// packed fp16 [pixel][channel/4][weight x4 | bias x4], one 16-B load per accumulator tile.
//
// The same file holds the training side: the SAVE variant of the forward, the backward
// kernel (LayerNorm / ReLU backward + transposed implicit GEMMs), the weight-gradient kernel (MFMA over pixels
// through ds_read_b64_tr_b16) and the parameter-gradient reduction kernel.
//
// Numerics: bf16 operands, f32 accumulation, f32 LayerNorm, bf16 activations between layers -- NOT
// bit-comparable with the f32 reference policy; tests bound the difference (tests/test_gpu_policy.py).
#if defined(INV_POLICY_STAMPS) && !defined(INV_TOOLS_BUILD)
#error "INV_POLICY_STAMPS is a diagnostic build (in-kernel stamps): tools/build_variant.py only (it defines INV_TOOLS_BUILD)"
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/inversus_hip.h"
#include "inv_internal.h"

// bf16 pipeline, not bit-comparable with the f32 module anyway: let the epilogue use FMAs even though the
// library is built with -ffp-contract=off for the env / GAE kernels
#pragma clang fp contract(fast)

#if (defined(POL_BWD_NO_GEMM) || defined(POL_BWD_NO_DY) || defined(POL_BWD_NO_DZ) || defined(POL_BWD_NO_XLOAD)) && !defined(INV_TOOLS_BUILD)
#error "POL_BWD_NO_* builds compute wrong gradients on purpose (timing probes): tools/build_variant.py only (it defines INV_TOOLS_BUILD)"
#endif

namespace pol {

constexpr int H = 10, W = 15, P = H * W;                 // 150 pixels
constexpr int PW = W + 2, PH = H + 2, NPOS = PW * PH;    // 12 x 17 haloed tile, 204 positions
constexpr int OBS_WORDS = (12 * W * H + 31) / 32;          // 57 u32 words of one packed observation
// GEMM pixel tiles are BOARD ROWS (15 pixels + one dummy column that reads the right halo) on a
// 288-B position stride.  ds_read_b128 is served in groups of 16 lanes that mix two channel groups (lanes 0-3, 12-15 of one,
// 4-11 of the next): with consecutive positions 72 dwords (8 banks mod 64) apart and channel groups 4 dwords apart all sixteen
// 16-B pieces of a group fall into different banks; a tile of 16 LINEAR pixels wraps to the next board row (a +2 position
// jump) and the 272-B stride puts lane (g, 12) on the banks of lane (g + 1, 11): 2-way conflicts on every fragment read
// (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE in round 1).  Conflict-free reads are what lets a wave own 2 row blocks x
// ALL 10 pixel tiles: every weight fragment is then loaded by ONE wave per workgroup instead of two (0.47 instead of 0.94 MB
// per sample through the 64 B/clk vector memory path) at twice the (now half as expensive) activation fragment reads.
constexpr int CSF = 144;                                 // forward: 288 B per position
constexpr int NPOSF = NPOS + 1;                          // the dummy column of the last board row reads one position past the halo
constexpr int THREADS = 256, WAVES = THREADS / 64;       // 4 waves per block, 2 blocks per CU (55 KB LDS each)
constexpr float LN_EPS = 1e-5f;

// The GEMM loops below are software pipelines written out in source order (weight fragments requested AD k-steps ahead,
// every activation fragment re-loaded right after its last MFMA of the k-step).  hipcc's scheduler undoes that: it sinks
// the loads to just before their first use (ds_read_b128 ... s_waitcnt lgkmcnt ... v_mfma on the same registers; weight
// loads waited on two instructions after their issue), which leaves every wave latency-bound and the matrix pipe busy only
// when a co-resident wave happens to fill the gaps.  A scheduling barrier after each group pins the written order.
#define POL_PIN() __builtin_amdgcn_sched_barrier(0)
// Timing probes of the banded forward (tools/band_knock_ab.sh; the results are WRONG on purpose): 1 no MFMAs, 2 no weight-fragment
// reloads in the k loop, 4 no global loads in the band input pass, 8 no z write-back stores, 16 no B-fragment LDS reads in the k
// loop, 32 no feature pass
#ifndef POL_TRUNK_KNOCK
#define POL_TRUNK_KNOCK 0                  // the 15 x 10 GEMM core's timing probes (tools/trunk_knock_ab.sh): 1 no MFMAs (operands still loaded), 2 no weight-fragment reloads, 16 no B-fragment LDS reads inside the loop, 32 no LayerNorm-parameter loads in the inference epilogue
#endif
#if POL_TRUNK_KNOCK && !defined(INV_TOOLS_BUILD)
#error "POL_TRUNK_KNOCK builds compute wrong features on purpose (timing probes): tools/build_variant.py only (it defines INV_TOOLS_BUILD)"
#endif
#ifndef POL_BAND_KNOCK
#define POL_BAND_KNOCK 0
#elif !defined(INV_TOOLS_BUILD)
#error "POL_BAND_KNOCK builds compute wrong features on purpose (timing probes): tools/build_variant.py only (it defines INV_TOOLS_BUILD)"
#endif

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

__device__ __forceinline__ int pos_of(int p) { return (p / W + 1) * PW + (p % W) + 1; }

// Diagnostic build only (-DINV_POLICY_STAMPS, never the shipped library): per-phase cycle shares of wave 0,
// accumulated into a buffer of its own that no kernel code reads (cdna_hip_programming.md, "In-kernel stamps").
#ifdef INV_POLICY_STAMPS
__device__ unsigned long long g_stamps[4 * 32];          // [wave][slot]
// Round 5: the deltas are accumulated in LDS (ds_add_u64, no return: counted in lgkmcnt, a few cycles) and flushed to g_stamps once
// per workgroup at the end of the kernel.  The first version did a GLOBAL atomic per stamp: every later `s_waitcnt vmcnt` of the
// kernel (LayerNorm parameters, weight fragments) then also waited for that atomic (vmcnt retires in order), which inflated the
// epilogue segments by the atomic's round trip and doubled the stamped build's run time.
__shared__ unsigned long long s_stamps[4 * 32];
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
        if ((threadIdx.x & 63) == 0) atomicAdd(&s_stamps[(threadIdx.x >> 6) * 32 + (slot)], now_ - stamp_prev); \
        stamp_prev = now_;                                                                   \
    } while (0)
#define STAMP_INIT() do { if (threadIdx.x < 128) s_stamps[threadIdx.x] = 0ull; __syncthreads(); } while (0)
#define STAMP_FLUSH() do { __syncthreads(); if (threadIdx.x < 128) atomicAdd(&g_stamps[threadIdx.x], s_stamps[threadIdx.x]); } while (0)
#else
#define STAMP(slot) do { } while (0)
#define STAMP_INIT() do { } while (0)
#define STAMP_FLUSH() do { } while (0)
#endif

// Sum over the 64 lanes, result in every lane.  Rows of 16 lanes reduce with four DPP adds (quad swaps, half-row and row
// mirror: VALU-only, no LDS crossbar); the four row sums are combined through readlane.  (__shfl_xor lowers to six
// dependent ds_bpermute round trips per value.)
__device__ __forceinline__ float wave_sum(float x)
{
#define POL_DPP_ADD(ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true))
    POL_DPP_ADD(0xB1);                               // quad_perm [1,0,3,2]
    POL_DPP_ADD(0x4E);                               // quad_perm [2,3,0,1]
    POL_DPP_ADD(0x141);                              // row_half_mirror
    POL_DPP_ADD(0x140);                              // row_mirror
#undef POL_DPP_ADD
    const int xi = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
    return (r0 + r1) + (r2 + r3);
}


// ---- GEMM core of the 10 x 15 kernels (round 5): input rows SHARED across the three vertical taps ---------------------------
// The B fragment of output row i for tap (dy, dx) is input row i + dy (haloed coordinates) shifted by dx: the same 16 positions x
// 32 channels that output row i + 1 needs for tap (dy - 1, dx).  A wave owns whole board rows, so it reads every input row
// fragment ONCE per (channel group, dx) and feeds it to the (up to) three output rows it belongs to, each with that tap row's own
// weight fragment:
//     for (kc, dx):  A[dy][mi] = W fragments of taps (0..2, dx), channel group kc            (3 * MW loads, one group ahead)
//                    for input row jq:  B = tile[row jq][.. + dx][kc]                         (ONE ds_read_b128)
//                                       acc[mi][jq - dy] += A[dy][mi] * B   for dy = 0, 1, 2   (up to 3 * MW MFMAs)
// instead of one read per (output row, tap, kc): 10 instead of 30 fragment reads per (kc, dx) for a wave that owns all ten rows
// (1 440 -> 480 KB of LDS reads per sample in conv4), one address register instead of NW.  FULL (the wave owns rows 0 .. H-1):
// the top and bottom halo rows are all zeros and are skipped -- 28 instead of 30 MFMAs per (kc, dx, mi), -6.7 % matrix work.
// k-order of every accumulator: (kc, dx, dy) -- the training forward, the rollout forward with its numerics and the backward's
// transposed convolutions all run THIS loop (bit-identical features between the first two is a tested property).
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// `early`: called once, at the start of the LAST (kc, dx) group (after every weight fragment has been requested): the caller's loads that
// should be in flight while the last 28 x MW MFMAs run (the LayerNorm parameters of the forward's epilogue).
// A layer with ONE channel group (CIN = 32: three groups of 15-28 MFMAs, each far shorter than an L2 round trip) requests all its nine
// fragments up front; the wide layers keep a ring of two groups.  (Measured and not kept, profiles/r05_trunk_experiments.md: the first
// group's fragments and the bias requested by the CALLER across the previous layer's statistics / barriers / epilogue -- 24 more live
// registers, no gain, as in round 2.)
template <int CIN, int MW, int NW, bool FULL, class HOOK = NoHook>
__device__ __forceinline__ void conv_gemm_rows(const void *w, const bf16 *tile, int lane, int m0, int n0, f32x4 (&acc)[MW][NW], const HOOK &early = HOOK())
{
    constexpr int KPT = CIN / 32, KS = 9 * KPT, NG = 3 * KPT;
    constexpr int JQ0 = FULL ? 1 : 0, NR = FULL ? NW : NW + 2;       // input rows jq = JQ0 .. JQ0 + NR - 1 (relative to haloed row n0)
    constexpr int NS = NG * NR;                                      // row steps of the whole GEMM
    constexpr int BD = 3;                                            // B fragments are requested BD - 1 row steps ahead
    constexpr int NSLOT = NG <= 3 ? NG : 2, PG = NG <= 3 ? NG : 1;   // one channel group: every fragment resident; else a ring of two groups
    const int c16 = lane & 15, kg = lane >> 4;
    const bf16 *rp = tile + (n0 * PW + c16) * CSF + 8 * kg;          // every fragment address = rp + an immediate
    const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(w) + (size_t)m0 * KS * 64 + lane;
    bf16x8 a[NSLOT][3][MW], b[BD];
#define POL_ROWS_A(g, slot)                                                                                        \
    _Pragma("unroll") for (int dy_ = 0; dy_ < 3; ++dy_)                                                            \
        _Pragma("unroll") for (int mi_ = 0; mi_ < MW; ++mi_)                                                       \
            a[slot][dy_][mi_] = wp[((size_t)mi_ * KS + (dy_ * 3 + (g) % 3) * KPT + (g) / 3) * 64]
#define POL_ROWS_B(t) (*reinterpret_cast<const bf16x8 *>(rp + ((JQ0 + (t) % NR) * PW + ((t) / NR) % 3) * CSF + (((t) / NR) / 3) * 32))
#pragma unroll
    for (int g = 0; g < ((POL_TRUNK_KNOCK & 2) ? 1 : PG); ++g) { POL_ROWS_A(g, g); }
#pragma unroll
    for (int t = 0; t < BD - 1; ++t) b[t] = POL_ROWS_B(t < NS ? t : NS - 1);
    // two nested fully unrolled loops (one flat loop of NS steps exceeds the unroller's size budget for pragma loops and is left
    // rolled, with the fragment arrays in scratch): every index below is a constant after unrolling
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (!(POL_TRUNK_KNOCK & 2) && NG > 3 && g + 1 < NG) { POL_ROWS_A(g + 1, (g + 1) & 1); }
        if (g == NG - 1) { early(); POL_PIN(); }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int t = g * NR + r, jq = JQ0 + r;
            if (!(POL_TRUNK_KNOCK & 16) && t + BD - 1 < NS) b[(t + BD - 1) % BD] = POL_ROWS_B(t + BD - 1);
            POL_PIN();
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int i = jq - dy;                               // output row (relative to n0) this input row serves through tap row dy
                if (i >= 0 && i < NW) {
#pragma unroll
                    for (int mi = 0; mi < MW; ++mi) {
                        if (POL_TRUNK_KNOCK & 1) {
                            typedef unsigned int knock_u4 __attribute__((ext_vector_type(4)));
                            asm volatile("" :: "v"(__builtin_bit_cast(knock_u4, a[(POL_TRUNK_KNOCK & 2) ? 0 : g % NSLOT][dy][mi])), "v"(__builtin_bit_cast(knock_u4, b[t % BD])));
                        } else {
                            acc[mi][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(POL_TRUNK_KNOCK & 2) ? 0 : g % NSLOT][dy][mi], b[t % BD], acc[mi][i], 0, 0, 0);
                        }
                    }
                }
            }
            POL_PIN();
        }
    }
#undef POL_ROWS_A
#undef POL_ROWS_B
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
// SAVE (training forward): additionally stores what the backward pass needs -- the normalised pre-activation
// x^ = (z - mean) * rstd and the activation, both bf16 channels-last [150][COUT] of this sample, and rstd.
struct LayerSave { bf16 *xhat; bf16 *act; float *rstd; };

template <int CIN, int COUT, int MW, int NW, bool RAGGED, bool RESIDUAL, bool SAVE = false, bool TN = false>
__device__ __forceinline__ void conv_ln_relu(const inv_policy_layer &lp, bf16 *tile, float *red, int lane, int m0, int n0, int ncount,
                                             unsigned long long &stamp_prev, int layer, const LayerSave sv = LayerSave{ nullptr, nullptr, nullptr })
{
    // Every fused multiply-add below is written as fmaf: with contraction left to the backend the SAVE and inference
    // instantiations fused the sum of squares differently (v_mul + v_fmac vs v_pk_mul + v_add), rstd differed in the last
    // bit for some samples and ~4e-7 of the features by one bf16 ulp -- the training forward and the rollout forward with
    // its numerics (TN) must agree bit for bit.
#pragma clang fp contract(off)
    const int c16 = lane & 15, kg = lane >> 4;

    int pix[NW], base[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        pix[i] = (n0 + i) * W + min(c16, W - 1);                    // tile = board row n0 + i; column 15 is a dummy (never stored)
        base[i] = ((n0 + i + 1) * PW + c16 + 1) * CSF;              // the dummy column sits on the right halo: reads zeros
    }

    // the accumulators start at the conv bias (4 consecutive channels per lane: one 16-B load per row block, requested
    // together with the first weight fragments instead of after the GEMM)
    f32x4 acc[MW][NW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        const float4 bq = *reinterpret_cast<const float4 *>(lp.bias + (m0 + mi) * 16 + 4 * kg);
#pragma unroll
        for (int i = 0; i < NW; ++i) { acc[mi][i][0] = bq.x; acc[mi][i][1] = bq.y; acc[mi][i][2] = bq.z; acc[mi][i][3] = bq.w; }
    }

    // residual (layer 4): this lane's own input elements are added to the accumulators' START values (bias + residual) BEFORE the GEMM
    // -- its inputs do not change, the LDS round trips hide behind the first MFMAs, and nothing but the statistics is left between the
    // GEMM and the statistics barrier (read after the GEMM they were ten dependent read -> wait -> add groups on the wave's critical
    // path: "bias+stats L4" = 9.3 % of a sample in profiles/r05_v2_policy_stamps_lds.json)
    if constexpr (RESIDUAL) {
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const bf16x4 r4 = *reinterpret_cast<const bf16x4 *>(tile + base[i] + (m0 + mi) * 16 + 4 * kg);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mi][i][e] += (float)r4[e];
            }
    }
    // LayerNorm affine parameters of ALL this wave's tiles -- fp16 [pixel][COUT/4][8] = {weight x4, bias x4} of 4 consecutive channels,
    // ONE 16-byte load per tile -- are requested early enough for their L2 round trip to overlap the statistics and the barrier (the
    // weight / activation fragment registers of the GEMM are dead there).  Requested group by group inside the epilogue they cost 15 %
    // of the kernel (a build without them: 1.08 vs 1.27 ms; profiles/r02_trunk_knockouts.md).  Round 5: requested at the start of the
    // GEMM's LAST (kc, dx) group (conv_gemm_rows' `early` hook) instead of after it: their round trip also hides behind the last
    // 28 x MW MFMAs, -2.5 % per forward (profiles/r05_trunk_experiments.md).
    f16x8 ln[SAVE ? 1 : MW][SAVE ? 1 : NW];
    auto load_ln = [&]() {
        if constexpr (!SAVE) {
#pragma unroll
            for (int mi = 0; mi < MW; ++mi)
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    if (POL_TRUNK_KNOCK & 32) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ln[mi][i][e] = (_Float16)(e < 4 ? 1.0f : 0.25f);
                        continue;
                    }
                    ln[mi][i] = *reinterpret_cast<const f16x8 *>(reinterpret_cast<const _Float16 *>(lp.ln) +
                                                                 ((size_t)min(pix[i], P - 1) * COUT + (m0 + mi) * 16 + 4 * kg) * 2);
                }
        }
    };
    conv_gemm_rows<CIN, MW, NW, NW == H>(lp.w, tile, lane, m0, n0, acc, load_ln);
    STAMP(1 + layer);                                                // GEMM
    // (+ residual) and the LayerNorm statistics
    static_assert(!RAGGED, "tile ownership is 5 or 10 full-width tiles per wave");
    (void)ncount;
    const bool vcol = c16 < W;                                       // the same for every tile of the lane
    float s, ss;
    // the sums run on accumulator PAIRS with v_pk_add_f32 / v_pk_fma_f32 (4 instructions per quad instead of 10; VALU issue,
    // not LDS or latency, is what the knock-out probes show to be scarce).  Every instantiation sums in the same order.
    // Four INDEPENDENT chains (sum / sum of squares x the two register pairs of a quad): a dependent vector instruction beside the
    // co-resident workgroup's MFMA stream issues once per MFMA slot, so one chain of 4 x MW x NW dependent packed adds was ~1 000 cycles
    // of this segment by itself.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 sA = { 0.0f, 0.0f }, sB = { 0.0f, 0.0f }, qA = { 0.0f, 0.0f }, qB = { 0.0f, 0.0f };
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const float x0 = acc[mi][i][0], x1 = acc[mi][i][1], x2 = acc[mi][i][2], x3 = acc[mi][i][3];
            {
                const f32x2 pa = { x0, x1 }, pb = { x2, x3 };
                sA += pa;
                sB += pb;
                qA = __builtin_elementwise_fma(pa, pa, qA);
                qB = __builtin_elementwise_fma(pb, pb, qB);
            }
        }
    }
    s = (sA[0] + sA[1]) + (sB[0] + sB[1]);
    ss = (qA[0] + qA[1]) + (qB[0] + qB[1]);
    if (!vcol) { s = 0.0f; ss = 0.0f; }                              // row tiles: the dummy column's sums
    s = wave_sum(s);
    ss = wave_sum(ss);
    const int wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave] = s; red[WAVES + wave] = ss; }
    STAMP(14 + layer);                                               // bias / residual / statistics of this wave
    __syncthreads();
    STAMP(5 + layer);                                                // wait at the statistics barrier
    // (multiplications by 1 / count and v_rsq_f32 instead of two f32 divisions and a square root + division: ~12 instead of ~50
    // DEPENDENT vector instructions between the statistics barrier and the first epilogue instruction of every wave)
    constexpr float inv_cnt = 1.0f / (float)(COUT * P);
    const float S = (red[0] + red[1]) + (red[2] + red[3]), SS = (red[WAVES] + red[WAVES + 1]) + (red[WAVES + 2] + red[WAVES + 3]);
    static_assert(WAVES == 4, "the cross-wave sums are written out for four waves");
    const float mean = S * inv_cnt;
    const float var = fmaxf(fmaf(-mean, mean, SS * inv_cnt), 0.0f);
    const float rstd = __builtin_amdgcn_rsqf(var + LN_EPS);
    const float shift = -mean * rstd;
    if (SAVE && sv.rstd && threadIdx.x == 0) *sv.rstd = rstd;

    if constexpr (SAVE) {
        // Training forward: x^ (rounded to bf16) goes into the tile in place of the activation, then ONE row pass by all
        // threads writes x^ to HBM as whole 2*COUT-byte rows, applies the affine + ReLU to what it just read (so the
        // backward, which recomputes y = x^ * w + b from the stored x^, sees the same ReLU mask) and leaves the activation
        // in the tile and in HBM -- coalesced 16-B stores instead of 8-B quads scattered from the accumulator layout.
        // Round 5: the row pass is a two-stage pipeline over batches of CH items (like the backward's): the LayerNorm parameters of
        // batch b + 1 are requested before batch b is consumed, batch 0 already before the x^ writes and the barrier.  As one rolled loop
        // (hipcc does not pipeline it) every item exposed an L2 round trip: ds_read -> two 16-byte loads -> wait -> compute -> stores.
        constexpr int C8 = COUT / 8, ITEMS = P * C8, NIT = (ITEMS + THREADS - 1) / THREADS, CH = NIT < 5 ? NIT : 5, NB = (NIT + CH - 1) / CH;
        int tid_l = threadIdx.x;
        asm volatile("" : "+v"(tid_l));                                // (keeps the item addresses out of the sample loop's hoisted set)
        f16x8 l0[2][CH], l1[2][CH];
        auto request = [&](int bt) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int k = bt * CH + j;
                if (k < NIT) {
                    const int q = min(tid_l + k * THREADS, ITEMS - 1);   // clamped: the tail threads re-load the last item and discard it
                    const _Float16 *lq = reinterpret_cast<const _Float16 *>(lp.ln) + (size_t)q * 16;   // (p * COUT + part * 8) * 2
                    l0[bt & 1][j] = *reinterpret_cast<const f16x8 *>(lq);
                    l1[bt & 1][j] = *reinterpret_cast<const f16x8 *>(lq + 8);
                }
            }
        };
        request(0);
        POL_PIN();
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
            const int ch0 = (m0 + mi) * 16 + 4 * kg;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                if (vcol) {
                    bf16x4 xq;
#pragma unroll
                    for (int e = 0; e < 4; ++e) xq[e] = (bf16)fmaf(acc[mi][i][e], rstd, shift);
                    *reinterpret_cast<bf16x4 *>(tile + base[i] + ch0) = xq;
                }
            }
        }
        __syncthreads();
        bf16x8 *xdst = reinterpret_cast<bf16x8 *>(sv.xhat), *adst = reinterpret_cast<bf16x8 *>(sv.act);
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            if (bt + 1 < NB) request(bt + 1);
            POL_PIN();
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int k = bt * CH + j, q = tid_l + k * THREADS;
                if (k < NIT && q < ITEMS) {
                    const int p = q / C8, part = q - p * C8;
                    bf16 *tp = tile + pos_of(p) * CSF + part * 8;
                    const bf16x8 xv = *reinterpret_cast<const bf16x8 *>(tp);
                    if (xdst) xdst[q] = xv;
                    const f16x8 w0 = l0[bt & 1][j], w1 = l1[bt & 1][j];
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = (bf16)fmaxf(fmaf((float)xv[e], (float)w0[e], (float)w0[4 + e]), 0.0f);
                        o[4 + e] = (bf16)fmaxf(fmaf((float)xv[4 + e], (float)w1[e], (float)w1[4 + e]), 0.0f);
                    }
                    *reinterpret_cast<bf16x8 *>(tp) = o;
                    if (adst) adst[q] = o;
                }
            }
            POL_PIN();
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
            const int ch0 = (m0 + mi) * 16 + 4 * kg;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                if (vcol) {
                    bf16x4 o;
                    const f16x8 q = ln[mi][i];
                    // pairs: x^ by one v_pk_fma_f32, ReLU AFTER the bf16 rounding as a signed 16-bit maximum with 0 on the packed pair
                    // (v_pk_max_i16: a negative bf16 is a negative int16; relu(round(x)) == round(relu(x))): 5 instead of 7 per pair
                    typedef short s16x2 __attribute__((ext_vector_type(2)));
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    const f32x2 rs2 = { rstd, rstd }, sh2 = { shift, shift };
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 a2 = { acc[mi][i][2 * h], acc[mi][i][2 * h + 1] };
                        f32x2 xh = __builtin_elementwise_fma(a2, rs2, sh2);
                        if (TN) { xh[0] = (float)(bf16)xh[0]; xh[1] = (float)(bf16)xh[1]; }     // train numerics: see below
                        // x^ * w + b with the fp16 parameters read in place (v_fma_mix_f32: f32 x f16 + f16, one rounding -- measured equal
                        // to v_fma_f32 on converted operands, tools/fma_mix_probe.hip); written as asm because hipcc otherwise pairs the two
                        // fmaf into a v_pk_fma_f32 fed by four v_cvt_f32_f16 (5 instructions instead of 2)
                        const uint32_t wq = __builtin_bit_cast(u32x4, q)[h], bq = __builtin_bit_cast(u32x4, q)[2 + h];
                        float y0, y1;
                        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(y0) : "v"(xh[0]), "v"(wq), "v"(bq));
                        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[0,1,1]" : "=v"(y1) : "v"(xh[1]), "v"(wq), "v"(bq));
                        uint32_t yp;                                       // both roundings in one instruction (hipcc: two + a v_perm_b32)
                        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(yp) : "v"(y0), "v"(y1));
                        const s16x2 z2 = { 0, 0 };
                        const bf16x2 r = __builtin_bit_cast(bf16x2, __builtin_elementwise_max(__builtin_bit_cast(s16x2, yp), z2));
                        o[2 * h] = r[0]; o[2 * h + 1] = r[1];
                    }
                    // IN PLACE: every read of the input tile by the GEMM happened before the statistics barrier
                    // above, and the residual read of this lane's own elements precedes this write
                    *reinterpret_cast<bf16x4 *>(tile + base[i] + ch0) = o;
                }
            }
        }
    STAMP(18 + layer);                                               // epilogue of this wave
    __syncthreads();
    }
    STAMP(9 + layer);                                                // wait at the end-of-layer barrier (SAVE: whole epilogue)
}

// (A start-up stagger of the two workgroups that share a CU -- the second one 30 k / 55 k cycles late -- was measured in round 2: no
// effect in either build, profiles/r02_trunk_ab_pin_stagger.log; removed.)

// BITS: the observation arrives as the packed bitstring inv_step writes under INV_STEP_OBS_BITS (57 words per sample).
// SAVE: training forward -- every layer also stores x^, its activation and rstd for k_policy_trunk_bwd.
template <bool BITS, bool SAVE, bool TN = false>
__global__ __launch_bounds__(THREADS, 2) void k_policy_trunk(inv_policy_params prm, const void *__restrict__ obs_in,
                                                          bf16 *__restrict__ feat, int N, inv_policy_saved sv)
{
    __shared__ __attribute__((aligned(16))) bf16 tile[NPOSF * CSF];         // 55 488 B: layers run in place -> 2 blocks / CU
    __shared__ float red[2 * WAVES];
    __shared__ uint32_t obw[BITS ? OBS_WORDS : 1];                             // BITS: the current sample's bitstring
    const float *obs = reinterpret_cast<const float *>(obs_in);
    const uint32_t *obits = reinterpret_cast<const uint32_t *>(obs_in);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)__builtin_amdgcn_readfirstlane((uint32_t)(tid >> 6));   // uniform -> tile ownership and weight pointers in SGPRs

    // zero both tiles once: the halo and the pad channels stay zero for the life of the block
    for (int i = tid; i < NPOSF * CSF / 8; i += THREADS) {
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.0f;
        reinterpret_cast<bf16x8 *>(&tile[0])[i] = z;
    }
    if (BITS && tid < OBS_WORDS && (int)blockIdx.x < N) obw[tid] = obits[(size_t)blockIdx.x * OBS_WORDS + tid];
    __syncthreads();

    unsigned long long stamp_prev = 0;
    STAMP_INIT();
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
                *reinterpret_cast<uint2 *>(&tile[pos_of(p) * CSF + part * 4]) = make_uint2(lo, hi);
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
                *reinterpret_cast<bf16x4 *>(&tile[pos_of(p) * CSF + 12 + part * 4]) = zz;
            }
            const float e0[4] = { v0.x, v0.y, v0.z, v0.w }, e1[4] = { v1.x, v1.y, v1.z, v1.w };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * q0 + j, c = e / P, p = e - c * P;
                tile[pos_of(p) * CSF + c] = (bf16)e0[j];
            }
            if (q1 < 12 * P / 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * q1 + j, c = e / P, p = e - c * P;
                    tile[pos_of(p) * CSF + c] = (bf16)e1[j];
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
        if (SAVE) {
            const size_t sn = (size_t)n * P;
            bf16 *const xh0 = reinterpret_cast<bf16 *>(sv.xhat[0]) + sn * 32, *const xh1 = reinterpret_cast<bf16 *>(sv.xhat[1]) + sn * 64;
            bf16 *const xh2 = reinterpret_cast<bf16 *>(sv.xhat[2]) + sn * 128, *const xh3 = reinterpret_cast<bf16 *>(sv.xhat[3]) + sn * 128;
            bf16 *const ac0 = reinterpret_cast<bf16 *>(sv.act[0]) + sn * 32, *const ac1 = reinterpret_cast<bf16 *>(sv.act[1]) + sn * 64;
            bf16 *const ac2 = reinterpret_cast<bf16 *>(sv.act[2]) + sn * 128;
            float *const rs = sv.rstd + (size_t)n * 4;
            conv_ln_relu<32, 32, 1, 5, false, false, true>(prm.layer[0], tile, red, lane_i, wave & 1, (wave >> 1) * 5, 5, stamp_prev, 0, LayerSave{ xh0, ac0, rs });
            conv_ln_relu<32, 64, 1, 10, false, false, true>(prm.layer[1], tile, red, lane_i, wave, 0, 10, stamp_prev, 1, LayerSave{ xh1, ac1, rs + 1 });
            conv_ln_relu<64, 128, 2, 10, false, false, true>(prm.layer[2], tile, red, lane_i, wave * 2, 0, 10, stamp_prev, 2, LayerSave{ xh2, ac2, rs + 2 });
            conv_ln_relu<128, 128, 2, 10, false, true, true>(prm.layer[3], tile, red, lane_i, wave * 2, 0, 10, stamp_prev, 3, LayerSave{ xh3, nullptr, rs + 3 });
        } else {
            conv_ln_relu<32, 32, 1, 5, false, false, false, TN>(prm.layer[0], tile, red, lane_i, wave & 1, (wave >> 1) * 5, 5, stamp_prev, 0);
            conv_ln_relu<32, 64, 1, 10, false, false, false, TN>(prm.layer[1], tile, red, lane_i, wave, 0, 10, stamp_prev, 1);
            conv_ln_relu<64, 128, 2, 10, false, false, false, TN>(prm.layer[2], tile, red, lane_i, wave * 2, 0, 10, stamp_prev, 2);
            conv_ln_relu<128, 128, 2, 10, false, true, false, TN>(prm.layer[3], tile, red, lane_i, wave * 2, 0, 10, stamp_prev, 3);
        }
        // features, channels-last [150][128] bf16 (the FC weights are permuted to this order on the host)
        bf16x8 *dst = reinterpret_cast<bf16x8 *>(feat + (size_t)n * P * 128);
        for (int q = tid; q < P * 16; q += THREADS) {
            const int p = q >> 4, part = q & 15;
            dst[q] = *reinterpret_cast<const bf16x8 *>(&tile[pos_of(p) * CSF + part * 8]);
        }
        if (BITS && tid < OBS_WORDS) obw[tid] = next_word;        // every read of obw happened before the first barrier of this sample
        __syncthreads();
        STAMP(13);                                                   // feature write-out
    }
    STAMP_FLUSH();
}


// ------------------------------------------------------------------------------------------------------------------
// Board-generic trunk for boards whose activations do not fit LDS ("banded"; SURVEY.md 8f row N1 for BASELINE config C5:
// 32 x 32 is 262 KB of bf16 activations per layer and sample).  Per sample and layer the board is processed in BANDS of R
// rows: the band's input (R + 2 rows with halo) is built in the LDS tile -- layer 1 from the observation, later layers from the
// previous layer's PRE-norm output z (bf16 channels-last in a per-workgroup scratch: 262 KB per workgroup, 134 MB for the
// 2 x 256 resident workgroups -- half the Infinity Cache; until round 5 two ping-pong buffers, 268 MB;
// the host side keeps ONE scratch per device and stream for all policies, fused_policy.py) with
// LayerNorm + affine + ReLU applied on the way in, which is possible because a layer's statistics are complete before the
// next layer starts -- then the same implicit GEMM as above runs on the band (tiles of 16 pixels = pieces of board rows: no
// dummy columns, conflict-free on the 288-B stride), the statistics are accumulated from the f32 accumulators, and z goes back
// to the scratch through the tile (whole 2*COUT-byte rows).  A last pass normalises z4 into the feature vector.
// z is rounded to bf16 BEFORE its LayerNorm (the 15 x 10 kernel normalises the f32 accumulator): one more rounding of the same
// size as the bf16 activations' own; tests bound the features against the f32 module.
template <int HB, int WB, int R>
struct BandGeo {
    static constexpr int P = HB * WB, PWB = WB + 2, TR = R + 2, NPOSB = TR * PWB + 1, NB = HB / R;
    static constexpr int TPR = WB / 16, NT = R * TPR;                 // 16-pixel tiles per board row / per band
    static constexpr int OWB = (12 * P + 31) / 32;
    static_assert(WB % 16 == 0 && HB % R == 0 && NT % 2 == 0 && NT <= 10, "band geometry");
};

// x^ w + b, ReLU, for 8 channels; l0 / l1: {weight x4 | bias x4} of channels 0..3 / 4..7.  (36 vector instructions per call.  A 28-
// instruction form -- packed f32 FMAs for x^, v_fma_mix_f32 written out, the ReLU as a packed int16 max on the rounded pairs -- measured
// the same 10.8 ms per 32 x 32 forward: the instruction count of this pass is not what bounds the kernel, profiles/r04_banded_probes.md.)
__device__ __forceinline__ bf16x8 ln_relu8(const bf16x8 z, const f16x8 l0, const f16x8 l1, float rstd, float shift)
{
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o[e] = (bf16)fmaxf(fmaf(fmaf((float)z[e], rstd, shift), (float)l0[e], (float)l0[4 + e]), 0.0f);
        o[4 + e] = (bf16)fmaxf(fmaf(fmaf((float)z[4 + e], rstd, shift), (float)l1[e], (float)l1[4 + e]), 0.0f);
    }
    return o;
}

// layers 2..4: tile <- relu(LN(z_prev)) for the band's R + 2 rows, zeros outside the board.
// band_rows_norm builds tile rows [ROW0, ROW0 + NROWS) from global memory -- board columns only: the halo columns and the guard
// position are zero from the kernel's start and nothing below writes them -- and the NROWS board rows are ONE contiguous piece of the
// channels-last tensor, so item q of the pass is the q-th 16-byte chunk of that piece.  `carry`: the band's rows 0, 1 are the
// previous band's rows R, R + 1, already normalised in the tile (band_conv stages its output in rows 0 .. R - 1 and leaves them
// alone); each thread moves the old contents of the position it is about to overwrite up by R rows first, so no barrier is needed
// and a band fetches R rows instead of R + 2 (8 bands of a 32-row board: 34 instead of 48 rows of z and of LayerNorm parameters).
template <class G, int R, int CPREV, int ROW0, int NROWS>
__device__ __forceinline__ void band_rows_norm(bf16 *tile, const bf16 *zprev, const void *__restrict__ ln_prev, float rstd, float shift, int band, bool carry)
{
    constexpr int WBc = G::PWB - 2, HBc = G::P / WBc, C8 = CPREV / 8, ROWI = WBc * C8, ITEMS = NROWS * ROWI, NIT = ITEMS / THREADS;
    static_assert(ITEMS % THREADS == 0 && NIT <= 8 && ((R - ROW0) * ROWI) % THREADS == 0, "whole passes; an item is carried or not by its index alone");
    // The item coordinates depend on the thread id only: hipcc hoists them out of the band loop and keeps ~4 registers per
    // item live across the GEMM, which then spills its weight pointers (scratch_load + s_waitcnt vmcnt(0) in every k-step).
    // Laundering the id makes them cheap per-band recomputations.
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int y0 = band * R + ROW0 - 1;
    bf16x8 z[NIT], old[NIT];
    f16x8 l0[NIT], l1[NIT];
    int dsto[NIT];
    bool in[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int q = tid + j * THREADS;
        const int rr = q / ROWI, rem = q - rr * ROWI, x = rem / C8, part = rem - x * C8;
        in[j] = (unsigned)(y0 + rr) < (unsigned)HBc;
        dsto[j] = ((ROW0 + rr) * G::PWB + x + 1) * CSF + part * 8;
        if (ROW0 + (j * THREADS) / ROWI >= R && carry) old[j] = *reinterpret_cast<const bf16x8 *>(tile + dsto[j]);
        const size_t o = in[j] ? ((size_t)(y0 + rr) * ROWI + rem) * 8 : 0;
        if (POL_BAND_KNOCK & 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { z[j][e] = (bf16)(float)(o & 3); l0[j][e] = (_Float16)1.0f; l1[j][e] = (_Float16)0.5f; }
            continue;
        }
        z[j] = *reinterpret_cast<const bf16x8 *>(zprev + o);
        const _Float16 *lq = reinterpret_cast<const _Float16 *>(ln_prev) + o * 2;
        l0[j] = *reinterpret_cast<const f16x8 *>(lq);
        l1[j] = *reinterpret_cast<const f16x8 *>(lq + 8);
    }
    POL_PIN();
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        if (ROW0 + (j * THREADS) / ROWI >= R && carry) *reinterpret_cast<bf16x8 *>(tile + dsto[j] - R * G::PWB * CSF) = old[j];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)0.0f;
        if (in[j]) o = ln_relu8(z[j], l0[j], l1[j], rstd, shift);
        *reinterpret_cast<bf16x8 *>(tile + dsto[j]) = o;
    }
}

template <class G, int R, int CPREV>
__device__ __forceinline__ void band_load_norm(bf16 *tile, const bf16 *zprev, const void *__restrict__ ln_prev, float rstd, float shift, int band)
{
    if (band == 0) band_rows_norm<G, R, CPREV, 0, 2>(tile, zprev, ln_prev, rstd, shift, band, false);
    band_rows_norm<G, R, CPREV, 2, R>(tile, zprev, ln_prev, rstd, shift, band, band > 0);
}

// layer 1: tile <- the observation's 12 planes (+ 20 zero channels) for the band's R + 2 rows, built and carried over like
// band_rows_norm's rows.  Item = (16-byte chunk of 8 channels, position): chunks 0 and 1 hold the planes, 2 and 3 are zeros, and a
// wave's 64 items share their chunk (NP is a multiple of 64), so the branch on it is uniform.
template <class G, int R, bool BITS, int ROW0, int NROWS>
__device__ __forceinline__ void band_rows_obs(bf16 *tile, const uint32_t *obw, const float *__restrict__ planes, int band, bool carry)
{
    constexpr int WBc = G::PWB - 2, HBc = G::P / WBc, NP = NROWS * WBc, ITEMS = 4 * NP, NIT = ITEMS / THREADS;
    static_assert(ITEMS % THREADS == 0 && NP % 64 == 0, "whole passes, one chunk per wave");
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int y0 = band * R + ROW0 - 1;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int q = tid + j * THREADS, part = q / NP, pos = q - part * NP, rr = pos / WBc, x = pos - rr * WBc, y = y0 + rr;
        bf16 *dst = tile + ((ROW0 + rr) * G::PWB + x + 1) * CSF + part * 8;
        if (carry && ROW0 + rr >= R) *reinterpret_cast<bf16x8 *>(dst - R * G::PWB * CSF) = *reinterpret_cast<const bf16x8 *>(dst);
        uint32_t w[4] = { 0u, 0u, 0u, 0u };
        if (part < 2 && (unsigned)y < (unsigned)HBc) {
            const int p = y * WBc + x;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = part * 8 + e;
                uint32_t h = 0u;
                if (c < 12) {
                    if (BITS) {
                        const int bi = c * G::P + p;
                        h = ((obw[bi >> 5] >> (bi & 31)) & 1u) * 0x3F80u;
                    } else {
                        h = __builtin_bit_cast(unsigned short, (bf16)planes[(size_t)c * G::P + p]);
                    }
                }
                w[e >> 1] |= h << (16 * (e & 1));
            }
        }
        *reinterpret_cast<uint4 *>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

template <class G, int R, bool BITS>
__device__ __forceinline__ void band_load_obs(bf16 *tile, const uint32_t *obw, const float *__restrict__ planes, int band)
{
    if (band == 0) band_rows_obs<G, R, BITS, 0, 2>(tile, obw, planes, band, false);
    band_rows_obs<G, R, BITS, 2, R>(tile, obw, planes, band, band > 0);
}


// The banded kernels' GEMM core with the input rows shared across the vertical taps (see conv_gemm_rows): a wave owns RW = NW / TPR
// whole rows of the band (TPR 16-pixel tiles per row), reads each of the RW + 2 input rows once per (channel group, dx, row half)
// and feeds it to the up to three output rows it belongs to.  Every row of a band tile can hold data (the halo rows are the
// neighbouring bands' rows), so no row is skipped.
template <class G, int CIN, int MW, int NW>
__device__ __forceinline__ void band_gemm_rows(const void *w, const bf16 *tile, int lane, int m0, int n0, f32x4 (&acc)[MW][NW])
{
    constexpr int KPT = CIN / 32, KS = 9 * KPT, NG = 3 * KPT, TPR = G::TPR, RW = NW / TPR;
    static_assert(NW % TPR == 0, "a wave owns whole rows of the band");
    constexpr int NR = (RW + 2) * TPR, NS = NG * NR, BD = 3;
    const int c16 = lane & 15, kg = lane >> 4;
    const bf16 *rp = tile + ((n0 / TPR) * G::PWB + c16) * CSF + 8 * kg;
    const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(w) + (size_t)m0 * KS * 64 + lane;
    bf16x8 a[2][3][MW], b[BD];
#define POL_ROWS_A(g, slot)                                                                                        \
    _Pragma("unroll") for (int dy_ = 0; dy_ < 3; ++dy_)                                                            \
        _Pragma("unroll") for (int mi_ = 0; mi_ < MW; ++mi_)                                                       \
            a[slot][dy_][mi_] = wp[((size_t)mi_ * KS + (dy_ * 3 + (g) % 3) * KPT + (g) / 3) * 64]
#define POL_ROWS_B(t) (*reinterpret_cast<const bf16x8 *>(rp + ((((t) % NR) / TPR) * G::PWB + (((t) % NR) % TPR) * 16 + ((t) / NR) % 3) * CSF + (((t) / NR) / 3) * 32))
    POL_ROWS_A(0, 0);
#pragma unroll
    for (int t = 0; t < BD - 1; ++t) b[t] = POL_ROWS_B(t < NS ? t : NS - 1);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if ((!(POL_BAND_KNOCK & 2)) && g + 1 < NG) { POL_ROWS_A(g + 1, (g + 1) & 1); }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int t = g * NR + r, jq = r / TPR, h = r % TPR;
            if (!(POL_BAND_KNOCK & 16) && t + BD - 1 < NS) b[(t + BD - 1) % BD] = POL_ROWS_B(t + BD - 1);
            POL_PIN();
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int ro = jq - dy;
                if (ro >= 0 && ro < RW) {
#pragma unroll
                    for (int mi = 0; mi < MW; ++mi) {
                        if (POL_BAND_KNOCK & 1) {
                            typedef unsigned int knock_u4 __attribute__((ext_vector_type(4)));
                            asm volatile("" :: "v"(__builtin_bit_cast(knock_u4, a[(POL_BAND_KNOCK & 2) ? 0 : (g & 1)][dy][mi])), "v"(__builtin_bit_cast(knock_u4, b[t % BD])));
                        } else {
                            acc[mi][ro * TPR + h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(POL_BAND_KNOCK & 2) ? 0 : (g & 1)][dy][mi], b[t % BD],
                                                                                           acc[mi][ro * TPR + h], 0, 0, 0);
                        }
                    }
                }
            }
            POL_PIN();
        }
    }
#undef POL_ROWS_A
#undef POL_ROWS_B
}

// conv3x3 on the band + bias (+ residual), statistics, z (bf16) back to the scratch through the tile
// WB: a hook on the write-back pass (k_band_conv_t's LayerNorm sums); items() sees the NOUT chunks a thread writes, as written.
struct NoWriteBackHook {
    template <int NOUT> __device__ __forceinline__ void items(int, int, const bf16x8 (&)[NOUT]) {}
};

template <class G, int R, int CIN, int COUT, int MW, int NW, bool RESIDUAL, class WB = NoWriteBackHook>
__device__ __forceinline__ void band_conv(const inv_policy_layer &lp, bf16 *tile, bf16 *zout, int band, int lane, int m0, int n0, float &s, float &ss,
                                          unsigned long long &stamp_prev, int slot0, WB &wb)
{
    asm volatile("" : "+v"(lane));
    const int c16 = lane & 15, kg = lane >> 4;
    int base[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int t = n0 + i, rr = t / G::TPR, col0 = (t - rr * G::TPR) * 16;
        base[i] = ((rr + 1) * G::PWB + col0 + c16 + 1) * CSF;
    }
    f32x4 acc[MW][NW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        const float4 bq = *reinterpret_cast<const float4 *>(lp.bias + (m0 + mi) * 16 + 4 * kg);
#pragma unroll
        for (int i = 0; i < NW; ++i) { acc[mi][i][0] = bq.x; acc[mi][i][1] = bq.y; acc[mi][i][2] = bq.z; acc[mi][i][3] = bq.w; }
    }
    band_gemm_rows<G, CIN, MW, NW>(lp.w, tile, lane, m0, n0, acc);
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        const int ch0 = (m0 + mi) * 16 + 4 * kg;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            float x0 = acc[mi][i][0], x1 = acc[mi][i][1], x2 = acc[mi][i][2], x3 = acc[mi][i][3];
            if (RESIDUAL) {
                const bf16x4 rq = *reinterpret_cast<const bf16x4 *>(tile + base[i] + ch0);
                x0 += (float)rq[0]; x1 += (float)rq[1]; x2 += (float)rq[2]; x3 += (float)rq[3];
                acc[mi][i][0] = x0; acc[mi][i][1] = x1; acc[mi][i][2] = x2; acc[mi][i][3] = x3;
            }
            s += (x0 + x1) + (x2 + x3);
            ss += (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
        }
    }
    STAMP(slot0 + 1);                                                // band GEMM + statistics
    __syncthreads();                                                 // every wave's GEMM reads of the band are done
    STAMP(slot0 + 2);                                                // wait for the other waves' GEMMs
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        const int ch0 = (m0 + mi) * 16 + 4 * kg;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)acc[mi][i][e];
            *reinterpret_cast<bf16x4 *>(tile + base[i] - G::PWB * CSF + ch0) = o;   // staged one row up: rows R, R + 1 stay the next band's rows 0, 1
        }
    }
    __syncthreads();
    constexpr int WBc = G::PWB - 2, C8 = COUT / 8;
    bf16x8 *dst = reinterpret_cast<bf16x8 *>(zout + (size_t)band * R * WBc * COUT);
    int tid1 = threadIdx.x;
    asm volatile("" : "+v"(tid1));
    constexpr int NOUT = R * WBc * C8 / THREADS;
    static_assert((R * WBc * C8) % THREADS == 0, "whole passes");
    bf16x8 zo[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {                                 // whole rows: consecutive threads, consecutive 16-B chunks
        const int q = tid1 + j * THREADS, pb = q / C8, part = q - pb * C8, rr = pb / WBc, x = pb - rr * WBc;
        zo[j] = *reinterpret_cast<const bf16x8 *>(tile + (rr * G::PWB + x + 1) * CSF + part * 8);
    }
#pragma unroll
    for (int j = 0; j < NOUT; ++j)
        if (!(POL_BAND_KNOCK & 8)) dst[tid1 + j * THREADS] = zo[j];
    wb.template items<NOUT>(band * R * WBc * C8, tid1, zo);
    __syncthreads();                                                 // the tile is rebuilt by the next band
    STAMP(slot0 + 3);                                                // z through the tile to global memory
}

__device__ __forceinline__ void band_stats(float s, float ss, float cnt, float *red, float &rstd, float &shift)
{
    s = wave_sum(s);
    ss = wave_sum(ss);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave] = s; red[WAVES + wave] = ss; }
    __syncthreads();
    float S = 0.0f, SS = 0.0f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { S += red[w]; SS += red[WAVES + w]; }
    __syncthreads();                                                 // red[] is reused by the next layer
    const float mean = S / cnt;
    const float var = fmaxf(SS / cnt - mean * mean, 0.0f);
    rstd = 1.0f / sqrtf(var + LN_EPS);
    shift = -mean * rstd;
}

// SAVE (training forward, BASELINE config C5 with the fused update): the layers' pre-norm outputs z go to per-sample tensors
// sv.z[l] ([N][P][C_l] bf16 channels-last) instead of the per-workgroup scratch, and the LayerNorm statistics to sv.stats
// ([N][4][2] f32: rstd, shift = -mean * rstd) -- everything the backward kernels below need; same arithmetic, same features.
template <int HB, int WB, int R, bool BITS, bool SAVE = false>
__global__ __launch_bounds__(THREADS, 2) void k_policy_trunk_banded(inv_policy_params prm, const void *__restrict__ obs_in, bf16 *__restrict__ feat,
                                                                     bf16 *__restrict__ scratch, int N, inv_policy_band_saved sv)
{
    using G = BandGeo<HB, WB, R>;
    __shared__ __attribute__((aligned(16))) bf16 tile[G::NPOSB * CSF];
    __shared__ float red[2 * WAVES];
    __shared__ uint32_t obw[BITS ? G::OWB : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)__builtin_amdgcn_readfirstlane((uint32_t)(tid >> 6));
    bf16 *const zW = SAVE ? nullptr : scratch + (size_t)blockIdx.x * G::P * 128;
    for (int i = tid; i < G::NPOSB * CSF / 8; i += THREADS) {
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.0f;
        reinterpret_cast<bf16x8 *>(&tile[0])[i] = z;
    }
    __syncthreads();
    STAMP_INIT();
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        int lane_i = lane;
        asm volatile("" : "+v"(lane_i));
        const float *planes = reinterpret_cast<const float *>(obs_in) + (size_t)n * 12 * G::P;
        if (BITS) {
            for (int w = tid; w < G::OWB; w += THREADS) obw[BITS ? w : 0] = reinterpret_cast<const uint32_t *>(obs_in)[(size_t)n * G::OWB + w];
            __syncthreads();
        }
        // where the four layers' z live: (SAVE) this sample's slices of the saved tensors, or ONE [P][128] scratch per workgroup in
        // which every layer's z ENDS at the end of the buffer (z_l at element P * (128 - C_l)).  A layer may overwrite its own input
        // there: band b has fetched the input's board rows up to R b + R before it stores output rows R b .. R b + R - 1 (the rows
        // after that come from the tile, band_rows_norm), and an output row of C' channels that starts P (C' - C) elements before
        // the input's ends at or before the first unread input row: HB (C' - C) + (R b + R + 1) C >= (R b + R) C' since R b + R <= HB.
        // Half the footprint of two ping-pong buffers (134 MB for 512 workgroups: inside the 256-MB Infinity Cache).
        bf16 *const z1 = SAVE ? reinterpret_cast<bf16 *>(sv.z[0]) + (size_t)n * G::P * 32 : zW + (size_t)G::P * 96;
        bf16 *const z2 = SAVE ? reinterpret_cast<bf16 *>(sv.z[1]) + (size_t)n * G::P * 64 : zW + (size_t)G::P * 64;
        bf16 *const z3 = SAVE ? reinterpret_cast<bf16 *>(sv.z[2]) + (size_t)n * G::P * 128 : zW;
        bf16 *const z4 = SAVE ? reinterpret_cast<bf16 *>(sv.z[3]) + (size_t)n * G::P * 128 : zW;
        float s, ss, rstd, shift;
        NoWriteBackHook nowb;
#ifdef INV_POLICY_STAMPS
        unsigned long long stamp_prev = stamp_now();                  // phase shares: tools/banded_stamps.py
#else
        unsigned long long stamp_prev = 0;
#endif
        // layer 1: 12 (-> 32) -> 32 channels
        s = ss = 0.0f;
#pragma unroll 1
        for (int band = 0; band < G::NB; ++band) {
            band_load_obs<G, R, BITS>(tile, obw, planes, band);
            __syncthreads();
            STAMP(0);
            band_conv<G, R, 32, 32, 1, G::NT / 2, false>(prm.layer[0], tile, z1, band, lane_i, wave & 1, (wave >> 1) * (G::NT / 2), s, ss, stamp_prev, 0, nowb);
        }
        band_stats(s, ss, (float)(32 * G::P), red, rstd, shift);
        if (SAVE && tid == 0) { sv.stats[(size_t)n * 8 + 0] = rstd; sv.stats[(size_t)n * 8 + 1] = shift; }
        // layer 2: 32 -> 64
        s = ss = 0.0f;
#pragma unroll 1
        for (int band = 0; band < G::NB; ++band) {
            band_load_norm<G, R, 32>(tile, z1, prm.layer[0].ln, rstd, shift, band);
            __syncthreads();
            STAMP(4);
            band_conv<G, R, 32, 64, 1, G::NT, false>(prm.layer[1], tile, z2, band, lane_i, wave, 0, s, ss, stamp_prev, 4, nowb);
        }
        band_stats(s, ss, (float)(64 * G::P), red, rstd, shift);
        if (SAVE && tid == 0) { sv.stats[(size_t)n * 8 + 2] = rstd; sv.stats[(size_t)n * 8 + 3] = shift; }
        // layer 3: 64 -> 128
        s = ss = 0.0f;
#pragma unroll 1
        for (int band = 0; band < G::NB; ++band) {
            band_load_norm<G, R, 64>(tile, z2, prm.layer[1].ln, rstd, shift, band);
            __syncthreads();
            STAMP(8);
            band_conv<G, R, 64, 128, 2, G::NT, false>(prm.layer[2], tile, z3, band, lane_i, wave * 2, 0, s, ss, stamp_prev, 8, nowb);
        }
        band_stats(s, ss, (float)(128 * G::P), red, rstd, shift);
        if (SAVE && tid == 0) { sv.stats[(size_t)n * 8 + 4] = rstd; sv.stats[(size_t)n * 8 + 5] = shift; }
        // layer 4: 128 -> 128, + residual
        s = ss = 0.0f;
#pragma unroll 1
        for (int band = 0; band < G::NB; ++band) {
            band_load_norm<G, R, 128>(tile, z3, prm.layer[2].ln, rstd, shift, band);
            __syncthreads();
            STAMP(12);
            band_conv<G, R, 128, 128, 2, G::NT, true>(prm.layer[3], tile, z4, band, lane_i, wave * 2, 0, s, ss, stamp_prev, 12, nowb);
        }
        band_stats(s, ss, (float)(128 * G::P), red, rstd, shift);
        if (SAVE && tid == 0) { sv.stats[(size_t)n * 8 + 6] = rstd; sv.stats[(size_t)n * 8 + 7] = shift; }
        // features: relu(LN4(z4)), channels-last [P][128]
        if (!(POL_BAND_KNOCK & 32)) {
            constexpr int ITEMS = G::P * 16, CH = 8;                 // a pass per memory round trip: 8 items in flight halve the passes of 4
            bf16x8 *dst = reinterpret_cast<bf16x8 *>(feat + (size_t)n * G::P * 128);
#pragma unroll 1
            for (int q0 = tid; q0 < ITEMS; q0 += CH * THREADS) {
                bf16x8 z[CH];
                f16x8 l0[CH], l1[CH];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int q = min(q0 + j * THREADS, ITEMS - 1);
                    z[j] = reinterpret_cast<const bf16x8 *>(z4)[q];
                    const _Float16 *lq = reinterpret_cast<const _Float16 *>(prm.layer[3].ln) + (size_t)q * 16;
                    l0[j] = *reinterpret_cast<const f16x8 *>(lq);
                    l1[j] = *reinterpret_cast<const f16x8 *>(lq + 8);
                }
                POL_PIN();
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int q = q0 + j * THREADS;
                    if (q < ITEMS) dst[q] = ln_relu8(z[j], l0[j], l1[j], rstd, shift);
                }
            }
        }
        __syncthreads();                                             // zB / obw are rewritten by the next sample
        STAMP(16);                                                   // statistics + feature pass
    }
    STAMP_FLUSH();
}


// ------------------------------------------------------------------------------------------------------------------
// Backward of the trunk for the PPO update: per sample, layer 4 -> 1,
//   dy  = da * [x^ w + b > 0]                       (ReLU mask recomputed from the saved x^)
//   dx^ = dy * w
//   dz  = rstd * (dx^ - mean(dx^) - x^ * mean(dx^ x^))         LayerNorm over the whole sample
//   da' = convT(dz)  (+ dz for the residual of layer 4)         implicit GEMM with the transposed weights
// dy and dz go to HBM (the parameter gradients are batch reductions / GEMMs over them), dz also into the LDS tile
// as the B operand of the transposed convolution, whose result da' returns to the tile as bf16 for the next layer.

// The LayerNorm / ReLU backward does not need the accumulator layout: da goes through the LDS tile as bf16 and ALL 256
// threads walk the sample row by row (8 channels = 16 B per item), so every HBM access is a coalesced 16-byte piece of a
// whole row, the work is balanced over the block, and almost nothing is live across the reductions -- the kernel fits two
// blocks per CU without spilling, like the forward (2.4 ms per 16 384 samples; a first version that ran the LayerNorm
// backward on the GEMM's accumulator registers spilled 460 B/lane at two blocks per CU (4.3 ms) and needed one block per
// CU with the whole register file to reach 3.0 ms).  Per layer: [A] dy = da * mask, sums of dx^ and dx^ x^ -> [B] dz into
// the tile and HBM -> transposed implicit GEMM on the tile -> da' (bf16) back into the tile.

// pass A + B of one layer.  FROM_GLOBAL: da comes from HBM (layer 4: dfeat) instead of the tile.
// items (pixel, 8 channels) per thread of a layer with COUT channels; CH of them are in flight at once (12-16 registers each)
template <int COUT> struct BwdItems {
    static constexpr int ITEMS = P * (COUT / 8), NIT = (ITEMS + THREADS - 1) / THREADS, CH = NIT < 5 ? NIT : 5;
};

// Registers of pass A that can be requested before the pass starts: x^ of all the thread's items (the first CH requested) and the
// two LayerNorm parameter buffers of the batch pipeline.
template <int COUT> struct BwdRegs {
    bf16x8 xc[BwdItems<COUT>::NIT];
    f16x8 l0[2][BwdItems<COUT>::CH], l1[2][BwdItems<COUT>::CH];
};

// The global loads of batch `bt` of a layer's pass A (x^ from HBM, LayerNorm parameters from L2).
template <int COUT>
__device__ __forceinline__ void bwd_request(BwdRegs<COUT> &rg, int bt, const void *ln_packed, const bf16 *xhat_n, int tid_l)
{
    constexpr int ITEMS = BwdItems<COUT>::ITEMS, NIT = BwdItems<COUT>::NIT, CH = BwdItems<COUT>::CH;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int k = bt * CH + j;
        if (k < NIT) {
            const int q = min(tid_l + k * THREADS, ITEMS - 1);     // clamped: the tail threads re-load the last item and discard it
#ifdef POL_BWD_NO_XLOAD
            for (int e = 0; e < 8; ++e) rg.xc[k][e] = (bf16)(0.01f * (float)(q & 63));
#else
            rg.xc[k] = reinterpret_cast<const bf16x8 *>(xhat_n)[q];
#endif
            const _Float16 *lq = reinterpret_cast<const _Float16 *>(ln_packed) + (size_t)q * 16;   // (p * COUT + part * 8) * 2
            rg.l0[bt & 1][j] = *reinterpret_cast<const f16x8 *>(lq);
            rg.l1[bt & 1][j] = *reinterpret_cast<const f16x8 *>(lq + 8);
        }
    }
}

// (Requesting batch 0 from the epilogue of the transposed convolution above this pass measured 1.89 vs 1.85-1.87 ms without: the other
// workgroup of the CU already fills that gap; removed.)
template <int COUT, bool FROM_GLOBAL>
__device__ __forceinline__ void ln_relu_bwd_rows(const void *ln_packed, const bf16 *xhat_n, float rstd, const bf16 *da_n, bf16 *dy_n, bf16 *dz_n,
                                                 bf16 *tile, float *red, BwdRegs<COUT> &rg, unsigned long long &stamp_prev, int layer)
{
    // A thread owns the same NIT items (pixel, 8 channels) in both passes.  Pass A is a two-stage pipeline over batches of CH
    // items: the global loads of batch b + 1 (x^, LayerNorm parameters, and da for the top layer) are requested before batch b
    // is consumed, and pass B re-uses the x^ and LayerNorm weights that pass A fetched (80 registers at COUT = 128; the GEMM's
    // registers are dead here) instead of loading them again.  (Requesting x^ even earlier, before the transposed convolution
    // that produces da, does not help: vmcnt retires in order, so the GEMM's first weight-fragment wait then also waits for the
    // HBM loads issued before it -- and 20-40 more live registers put the 2 x 10-tile GEMM into scratch: 2.45-2.70 vs 1.90 ms.)
    constexpr int CH8 = COUT / 8, ITEMS = BwdItems<COUT>::ITEMS, NIT = BwdItems<COUT>::NIT, CH = BwdItems<COUT>::CH;
    constexpr int NB = (NIT + CH - 1) / CH;
    // (hipcc hoists this thread's item addresses -- sample-invariant LayerNorm parameter pointers of all four layers -- out of the
    // sample loop and spills 39 of those registers: 156 B/lane of scratch, written once before the loop and re-loaded once per
    // sample OUTSIDE the GEMM loops.  Laundering `tid` per call removes the hoist: 214 VGPRs, no scratch -- and 1.89 vs 1.85 ms,
    // the address arithmetic then runs per sample.  The spill is the cheaper form; profiles/r04_trunk_bwd_experiments.md.)
    const int tid = threadIdx.x;
    bf16x8 (&xc)[NIT] = rg.xc;                           // x^ of the thread's items
    f16x8 (&l0)[2][CH] = rg.l0, (&l1)[2][CH] = rg.l1;
    f16x8 gc[NIT];                                       // their 8 LayerNorm weights
    float s1c[2] = { 0.0f, 0.0f }, s2c[2] = { 0.0f, 0.0f };   // two independent chains each (see conv_ln_relu's statistics; four spill this kernel: 320 B of scratch, 2.36 ms)
    bf16x8 dag[2][FROM_GLOBAL ? CH : 1];
    auto request = [&](int bt) {
        bwd_request<COUT>(rg, bt, ln_packed, xhat_n, tid);
        if (FROM_GLOBAL) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int k = bt * CH + j;
                if (k < NIT) dag[bt & 1][j] = reinterpret_cast<const bf16x8 *>(da_n)[min(tid + k * THREADS, ITEMS - 1)];
            }
        }
    };
    request(0);
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
        if (bt + 1 < NB) request(bt + 1);
        POL_PIN();
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int k = bt * CH + j, q = tid + k * THREADS;
            if (k < NIT && q < ITEMS) {
                const int p = q / CH8, part = q - p * CH8;
                bf16 *tp = tile + pos_of(p) * CSF + part * 8;
                const bf16x8 da = FROM_GLOBAL ? dag[bt & 1][FROM_GLOBAL ? j : 0] : *reinterpret_cast<const bf16x8 *>(tp);
                const bf16x8 xh = xc[k];
                bf16x8 dy;
                f16x8 gk;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const _Float16 gh = e < 4 ? l0[bt & 1][j][e] : l1[bt & 1][j][e - 4];
                    gk[e] = gh;
                    const float x = (float)xh[e], g = (float)gh, bb = (float)(e < 4 ? l0[bt & 1][j][4 + e] : l1[bt & 1][j][e]);
                    const float d = fmaf(x, g, bb) > 0.0f ? (float)da[e] : 0.0f;
                    dy[e] = (bf16)d;
                    const float dxh = (float)dy[e] * g;  // from the ROUNDED dy: pass B recomputes exactly this value
                    s1c[e & 1] += dxh;
                    s2c[e & 1] += dxh * x;
                }
                gc[k] = gk;
#ifndef POL_BWD_NO_DY
                // top layer (FROM_GLOBAL): dy = mask * dfeat can be recomputed by the parameter-sum kernel from dfeat, which is in HBM
                // anyway (ABI 8: grads.dy[3] == NULL) -- 0.63 GB per 16 384 samples that need not be written
                if (!FROM_GLOBAL || dy_n) reinterpret_cast<bf16x8 *>(dy_n)[q] = dy;
#endif
                *reinterpret_cast<bf16x8 *>(tp) = dy;
            }
        }
        POL_PIN();
    }
    float s1 = wave_sum(s1c[0] + s1c[1]);
    float s2 = wave_sum(s2c[0] + s2c[1]);
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) { red[wave] = s1; red[WAVES + wave] = s2; }
    STAMP(0 + layer);                                                // pass A (loads, dy, partial sums) of this wave
    __syncthreads();
    STAMP(4 + layer);                                                // wait for the other waves' sums
    float S1 = 0.0f, S2 = 0.0f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { S1 += red[w]; S2 += red[WAVES + w]; }
    const float inv_cnt = 1.0f / (float)(COUT * P);
    const float m1 = S1 * inv_cnt, m2 = S2 * inv_cnt;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {                      // same items as in pass A: this thread's own tile entries
        const int q = tid + k * THREADS;
        if (q < ITEMS) {
            const int p = q / CH8, part = q - p * CH8;
            bf16 *tp = tile + pos_of(p) * CSF + part * 8;
            const bf16x8 dy = *reinterpret_cast<const bf16x8 *>(tp);
            bf16x8 dz;
#pragma unroll
            for (int e = 0; e < 8; ++e) dz[e] = (bf16)(rstd * ((float)dy[e] * (float)gc[k][e] - m1 - (float)xc[k][e] * m2));
#ifndef POL_BWD_NO_DZ
            reinterpret_cast<bf16x8 *>(dz_n)[q] = dz;
#endif
            *reinterpret_cast<bf16x8 *>(tp) = dz;
        }
    }
    STAMP(8 + layer);                                                // pass B (dz to HBM and into the tile)
    __syncthreads();                                 // dz tile complete (B operand of the transposed convolution); red[] reusable
    STAMP(12 + layer);                                               // wait for the dz tile
}

// transposed conv of the dz tile -> da of the layer below, written back into the tile as bf16 (channels 0 .. COUT_BELOW)
template <int CIN_T, int COUT_BELOW, int MW, int NW, bool RESIDUAL>
__device__ __forceinline__ void conv_t_to_tile(const void *wt, bf16 *tile, int lane, int m0, int n0, BwdRegs<COUT_BELOW> &next, const void *next_ln,
                                               const bf16 *next_xhat, unsigned long long &stamp_prev, int layer)
{
    f32x4 acc[MW][NW];
    const int c16 = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            // z4 = conv4(a3) + b4 + a3: da3 = convT(dz4) + dz4 -- the accumulators START at this lane's own entries of the dz tile (the
            // B operand of this GEMM, unchanged by it), read before the MFMAs instead of after them
            acc[mi][i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
            if (RESIDUAL) {
                const int posi = (n0 + i + 1) * PW + c16 + 1;                                // dummy column: halo, zeros
                const bf16x4 r = *reinterpret_cast<const bf16x4 *>(tile + posi * CSF + (m0 + mi) * 16 + 4 * kg);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mi][i][e] = (float)r[e];
            }
        }
#ifdef POL_BWD_NO_GEMM
    for (int mi = 0; mi < MW; ++mi) for (int i = 0; i < NW; ++i) for (int q = 0; q < 4; ++q) acc[mi][i][q] = 0.001f * (float)(lane + mi + i);
#else
    conv_gemm_rows<CIN_T, MW, NW, NW == H>(wt, tile, lane, m0, n0, acc);
#endif
    STAMP(16 + layer);                                               // transposed-convolution GEMM of this wave
    __syncthreads();                                 // every wave has finished reading the dz tile
    STAMP(20 + layer);                                               // wait for the other waves' GEMMs
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const bool pv = c16 < W;
            const int posi = (n0 + i + 1) * PW + c16 + 1;
            if (pv) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)acc[mi][i][e];
                *reinterpret_cast<bf16x4 *>(tile + posi * CSF + (m0 + mi) * 16 + 4 * kg) = o;
            }
        }
    __syncthreads();
    STAMP(24 + layer);                                               // da into the tile + barrier
}

__global__ __launch_bounds__(THREADS, 2) void k_policy_trunk_bwd(inv_policy_bwd_params prm, inv_policy_saved sv, const bf16 *__restrict__ dfeat,
                                                                   inv_policy_grads gr, int N)
{
    __shared__ __attribute__((aligned(16))) bf16 tile[NPOSF * CSF];
    __shared__ float red[2 * WAVES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)__builtin_amdgcn_readfirstlane((uint32_t)(tid >> 6));
    for (int i = tid; i < NPOSF * CSF / 8; i += THREADS) {     // halo stays zero for the life of the block
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.0f;
        reinterpret_cast<bf16x8 *>(&tile[0])[i] = z;
    }
    __syncthreads();
    unsigned long long stamp_prev = 0;                               // (diagnostic builds: tools/bwd_stamps.py)
    STAMP_INIT();
#ifdef INV_POLICY_STAMPS
    stamp_prev = stamp_now();
#endif
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        int lane_i = lane;
        asm volatile("" : "+v"(lane_i));
        const size_t sn = (size_t)n * P;
        const float4 rs = *reinterpret_cast<const float4 *>(sv.rstd + (size_t)n * 4);
        const bf16 *xh3 = reinterpret_cast<const bf16 *>(sv.xhat[3]) + sn * 128, *xh2 = reinterpret_cast<const bf16 *>(sv.xhat[2]) + sn * 128;
        const bf16 *xh1 = reinterpret_cast<const bf16 *>(sv.xhat[1]) + sn * 64, *xh0 = reinterpret_cast<const bf16 *>(sv.xhat[0]) + sn * 32;
        BwdRegs<128> ra;
        BwdRegs<64> rb;
        BwdRegs<32> rc;
        ln_relu_bwd_rows<128, true>(prm.ln[3], xh3, rs.w, dfeat + sn * 128, gr.dy[3] ? reinterpret_cast<bf16 *>(gr.dy[3]) + sn * 128 : nullptr,
                                    reinterpret_cast<bf16 *>(gr.dz[3]) + sn * 128, tile, red, ra, stamp_prev, 3);
        conv_t_to_tile<128, 128, 2, 10, true>(prm.wt[2], tile, lane_i, wave * 2, 0, ra, prm.ln[2], xh2, stamp_prev, 3);
        ln_relu_bwd_rows<128, false>(prm.ln[2], xh2, rs.z, nullptr, reinterpret_cast<bf16 *>(gr.dy[2]) + sn * 128,
                                            reinterpret_cast<bf16 *>(gr.dz[2]) + sn * 128, tile, red, ra, stamp_prev, 2);
        conv_t_to_tile<128, 64, 1, 10, false>(prm.wt[1], tile, lane_i, wave, 0, rb, prm.ln[1], xh1, stamp_prev, 2);
        ln_relu_bwd_rows<64, false>(prm.ln[1], xh1, rs.y, nullptr, reinterpret_cast<bf16 *>(gr.dy[1]) + sn * 64,
                                           reinterpret_cast<bf16 *>(gr.dz[1]) + sn * 64, tile, red, rb, stamp_prev, 1);
        conv_t_to_tile<64, 32, 1, 5, false>(prm.wt[0], tile, lane_i, wave & 1, (wave >> 1) * 5, rc, prm.ln[0], xh0, stamp_prev, 1);
        ln_relu_bwd_rows<32, false>(prm.ln[0], xh0, rs.x, nullptr, reinterpret_cast<bf16 *>(gr.dy[0]) + sn * 32,
                                           reinterpret_cast<bf16 *>(gr.dz[0]) + sn * 32, tile, red, rc, stamp_prev, 0);
    }
    STAMP_FLUSH();
}


// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of one convolution:  dW[co][tap][ci] = sum_n sum_p dz[n][p][co] * x[n][p + tap][ci].
// The contraction runs over PIXELS, the strided dimension of the channels-last tensors, so both MFMA operands are
// fetched with gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 columns and each
// lane receives one column): rows = pixels (k), columns = channels (m for dz, n for x).  Every lane supplies its own row
// address, so the zero-haloed tile of the forward kernel serves all nine taps (row = position + tap offset).
// Block = 8 waves; wave (mslot, nb) accumulates MWW x 9 tiles  D[16 co][16 ci] of its channel block nb for all taps over
// the block's share of the samples, then adds them into dW with f32 atomics.  K = 150 pixels padded to 160 (5 k-steps;
// the padding rows of the dz tile are zero).
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_ptr;

__device__ __forceinline__ bf16x8 tr_frag(const bf16 *row_lo, const bf16 *row_hi)
{
    // rows k = 8g + q (lo) and 8g + 4 + q (hi) of the k-step, 4 consecutive columns at this lane's address
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(row_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(row_hi));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = { lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3] };
    return __builtin_bit_cast(bf16x8, v);
}

constexpr int WG_THREADS = 512, WG_WAVES = 8;

// Geometry of one work unit of the weight-gradient kernel: PU pixels of dz (the contraction index, KST k-steps of 32) against
// the XP pixels of x that their nine taps touch, laid out in a zero-haloed LDS tile of NPOS = 204 positions, PWg per tile row.
//   WgGeo15:   unit = a sample of the 10 x 15 board: 150 dz pixels, the same 150 x pixels, tile 12 x 17.
//   WgGeoBand: unit = a band of 4 rows of the 32 x 32 board (8 per sample): 128 dz pixels against the 6 rows (192 pixels) around
//              them -- contiguous in the channels-last tensors -- tile 6 x 34; rows outside the board are staged as zeros.
struct WgGeo15 {
    static constexpr int PU = P, KST = 5, XP = P, PWg = PW;
    __device__ static __forceinline__ int xpos(int i) { return pos_of(i); }
    __device__ static __forceinline__ int kpos(int pk) { return pos_of(pk); }
    __device__ static __forceinline__ long long x_first(int u) { return (long long)u * P; }
    __device__ static __forceinline__ bool x_valid(int, int) { return true; }
};
struct WgGeoBand {
    static constexpr int PU = 128, KST = 4, XP = 192, PWg = 34;
    __device__ static __forceinline__ int xpos(int i) { return (i >> 5) * 34 + (i & 31) + 1; }
    __device__ static __forceinline__ int kpos(int pk) { return ((pk >> 5) + 1) * 34 + (pk & 31) + 1; }
    __device__ static __forceinline__ long long x_first(int u) { return (long long)u * 128 - 32; }
    __device__ static __forceinline__ bool x_valid(int u, int i) { return (unsigned)(4 * (u & 7) - 1 + (i >> 5)) < 32u; }
};

// PF (the wide layers): one block per CU with twice the tiles per wave (MWW = 4: fewer output slices, so x / dz are
// re-read less often, and fewer LDS reads per MFMA), and the NEXT sample's dz / x chunks are requested into registers
// before the MFMAs of the current one, so the HBM / L2 latency of the tile fill overlaps the matrix work.
template <int CINP, int COUT, int MWW, int NBW, bool PF, class GEO = WgGeo15>
__global__ __launch_bounds__(WG_THREADS, PF ? 1 : 2) void k_policy_wgrad(const bf16 *__restrict__ x, const bf16 *__restrict__ dz, float *__restrict__ dw, float *__restrict__ db, int N)
{
    constexpr int PU = GEO::PU, KST = GEO::KST, KPADG = 32 * KST, XP = GEO::XP, PWg = GEO::PWg;     // N counts UNITS (GEO)
    constexpr int MSLOTS = WG_WAVES / NBW, MS = MWW * MSLOTS;            // m-blocks (16 channels) per block
    constexpr int MBLOCKS = COUT / 16, Q = (MBLOCKS + MS - 1) / MS;       // output slices (blockIdx.x % Q)
    constexpr int DCH = MS * 16, DSTRIDE = DCH + 8;                      // dz tile: [160][DCH] bf16, 16-B row pad
    constexpr int XSTRIDE = CINP + 8;                                    // x tile:  [204 positions][CINP] bf16, zero halo
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_lds[];
    bf16 *const dzt = reinterpret_cast<bf16 *>(wg_lds);
    bf16 *const xt = dzt + KPADG * DSTRIDE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)__builtin_amdgcn_readfirstlane((uint32_t)(tid >> 6));
    const int mslot = wave / NBW, nb = wave % NBW;
    const int slice = blockIdx.x % Q, sidx = blockIdx.x / Q, nslices = gridDim.x / Q;
    const int co_base = slice * DCH;                                     // first output channel of this block
    const int mb0 = mslot * MWW;                                         // wave's first m-block inside the slice
    const bool active = co_base + (mb0 + MWW) * 16 <= COUT;              // wave-uniform (layer 1: 2 of 8 m-slots exist)

    for (int i = tid; i < (KPADG * DSTRIDE + NPOS * XSTRIDE) / 8; i += WG_THREADS) {
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.0f;
        reinterpret_cast<bf16x8 *>(dzt)[i] = z;
    }
    f32x4 acc[MWW][9];
#pragma unroll
    for (int mi = 0; mi < MWW; ++mi)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mi][t][e] = 0.0f;

    // per-lane row addresses of the transposing reads: lane 4q + p of group g -> row 8g + 4h + q, columns 4p .. 4p+3
    const int g = lane >> 4, q = (lane >> 2) & 3, p4 = (lane & 3) * 4;
    int xrow[KST][2];
#pragma unroll
    for (int ks = 0; ks < KST; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pk = 32 * ks + 8 * g + 4 * h + q;                  // pixel of this k row; >= 150: the dz row is zero
            xrow[ks][h] = (GEO::kpos(min(pk, PU - 1)) - (PWg + 1)) * XSTRIDE + nb * 16 + p4;
        }
    const int arow = (8 * g + q) * DSTRIDE + mb0 * 16 + p4;

    constexpr int DZ_CHUNKS = DCH / 8, X_CHUNKS = CINP / 8;              // 16-byte chunks per pixel row
    static_assert(WG_THREADS % DZ_CHUNKS == 0, "a thread must always load the same channel chunk of dz");
    constexpr int DZC = (PU * DZ_CHUNKS + WG_THREADS - 1) / WG_THREADS, XC = (XP * X_CHUNKS + WG_THREADS - 1) / WG_THREADS;
    float zsum[8] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };         // sum over this thread's pixels and samples of its 8 channels of dz
    const int dcols = min(DCH, COUT - co_base);                          // channels of the slice that exist
    const bool dz_mine = (tid % DZ_CHUNKS) * 8 < dcols;                  // this thread's channel chunk exists in the slice
    bf16x8 rdz[DZC], rx[XC];                                             // one sample's chunks of this thread (PF: the next sample's)

    auto fetch = [&](int n) {                                            // HBM / L2 -> registers
        const bf16 *dzn = dz + (size_t)n * PU * COUT + co_base, *xn = x + GEO::x_first(n) * CINP;
#pragma unroll
        for (int k = 0; k < DZC; ++k) {
            const int c = tid + k * WG_THREADS, px = c / DZ_CHUNKS, part = c - px * DZ_CHUNKS;
            if (c < PU * DZ_CHUNKS && dz_mine) rdz[k] = *reinterpret_cast<const bf16x8 *>(dzn + (size_t)px * COUT + part * 8);
        }
#pragma unroll
        for (int k = 0; k < XC; ++k) {
            const int c = tid + k * WG_THREADS, px = c / X_CHUNKS, part = c - px * X_CHUNKS;
            if (c < XP * X_CHUNKS) {
                if (GEO::x_valid(n, px)) rx[k] = *reinterpret_cast<const bf16x8 *>(xn + (ptrdiff_t)px * CINP + part * 8);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) rx[k][j] = (bf16)0.0f;
                }
            }
        }
    };
    auto stage = [&]() {                                                 // registers -> LDS tiles (+ conv bias gradient on the way)
#pragma unroll
        for (int k = 0; k < DZC; ++k) {
            const int c = tid + k * WG_THREADS, px = c / DZ_CHUNKS, part = c - px * DZ_CHUNKS;
            if (c < PU * DZ_CHUNKS && dz_mine) {
                *reinterpret_cast<bf16x8 *>(dzt + px * DSTRIDE + part * 8) = rdz[k];
#pragma unroll
                for (int j = 0; j < 8; ++j) zsum[j] += (float)rdz[k][j];
            }
        }
#pragma unroll
        for (int k = 0; k < XC; ++k) {
            const int c = tid + k * WG_THREADS, px = c / X_CHUNKS, part = c - px * X_CHUNKS;
            if (c < XP * X_CHUNKS) *reinterpret_cast<bf16x8 *>(xt + GEO::xpos(px) * XSTRIDE + part * 8) = rx[k];
        }
    };
    auto mfmas = [&]() {
        if (active) {
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                bf16x8 a[MWW];
#pragma unroll
                for (int mi = 0; mi < MWW; ++mi)
                    a[mi] = tr_frag(dzt + arow + (32 * ks) * DSTRIDE + mi * 16, dzt + arow + (32 * ks + 4) * DSTRIDE + mi * 16);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int toff = ((t / 3) * PWg + (t % 3)) * XSTRIDE;
                    const bf16x8 b = tr_frag(xt + xrow[ks][0] + toff, xt + xrow[ks][1] + toff);
#pragma unroll
                    for (int mi = 0; mi < MWW; ++mi)
                        acc[mi][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b, acc[mi][t], 0, 0, 0);
                }
            }
        }
    };

    if (PF) {
        if (sidx < N) fetch(sidx);
        __syncthreads();                                                 // the zero fill is complete
        for (int n = sidx; n < N; n += nslices) {
            stage();                                                     // sample n: registers -> tiles
            __syncthreads();
            if (n + nslices < N) fetch(n + nslices);                     // sample n+1 in flight during the MFMAs of sample n
            mfmas();
            __syncthreads();                                             // all reads of the tiles are done
        }
    } else {
        for (int n = sidx; n < N; n += nslices) {
            __syncthreads();                                             // previous sample's reads are done (and the zero fill)
            fetch(n);
            stage();
            __syncthreads();
            mfmas();
        }
    }
    if (db) {                                                            // every dz element is loaded by exactly one block
        // block-level sum in LDS first (the dz tile is free now): one global atomic per channel and block, not per thread
        __syncthreads();
        float *zred = reinterpret_cast<float *>(wg_lds);
        if (tid < DCH) zred[tid] = 0.0f;
        __syncthreads();
        const int part = tid % DZ_CHUNKS;
        if (part * 8 < dcols) {
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(zred + part * 8 + j, zsum[j]);
        }
        __syncthreads();
        if (tid < dcols) atomicAdd(db + co_base + tid, zred[tid]);
    }
    if (active) {
        const int r = lane & 15;
#pragma unroll
        for (int mi = 0; mi < MWW; ++mi)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = co_base + (mb0 + mi) * 16 + 4 * g + e;
                    // [co][tap][ci]: a 16-lane group adds 64 consecutive bytes.  (Adding straight into the module's [co][ci][tap]
                    // order -- lanes 36 B apart -- doubled the kernel's time: 0.60 -> 1.36 ms at 128 -> 128; the layout change is
                    // k_wgrad_to_module's job, one launch for the four layers.)
                    atomicAdd(dw + ((size_t)co * 9 + t) * CINP + nb * 16 + r, acc[mi][t][e]);
                }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Batch reductions behind the LayerNorm gradients: for every element e of a layer's [150][C] map
//   g_w[e] = sum_n dy[n][e] * xhat[n][e]      g_b[e] = sum_n dy[n][e]
// One pass over the two bf16 tensors: a thread owns 8 consecutive channels (16-byte loads), blockIdx.y strides over the
// samples and writes its partial sums to part[slice][2][E]; the caller adds the slices.  (The conv bias gradient, the
// sum of dz, is collected by k_policy_wgrad while it stages dz.)
__global__ __launch_bounds__(256) void k_policy_param_grads(const bf16 *__restrict__ dy, const bf16 *__restrict__ xhat,
                                                        float *__restrict__ part, int E8, int N)
{
    const int e8 = blockIdx.x * 256 + threadIdx.x;
    if (e8 >= E8) return;
    float aw[8], ab[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) aw[j] = ab[j] = 0.0f;
    const size_t stride = (size_t)E8 * 8;
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const size_t o = (size_t)n * stride + (size_t)e8 * 8;
        const bf16x8 d = *reinterpret_cast<const bf16x8 *>(dy + o);
        const bf16x8 x = *reinterpret_cast<const bf16x8 *>(xhat + o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float df = (float)d[j];
            aw[j] = fmaf(df, (float)x[j], aw[j]);
            ab[j] += df;
        }
    }
    float *o = part + (size_t)blockIdx.y * 2 * stride + (size_t)e8 * 8;
    *reinterpret_cast<float4 *>(o) = make_float4(aw[0], aw[1], aw[2], aw[3]);
    *reinterpret_cast<float4 *>(o + 4) = make_float4(aw[4], aw[5], aw[6], aw[7]);
    *reinterpret_cast<float4 *>(o + stride) = make_float4(ab[0], ab[1], ab[2], ab[3]);
    *reinterpret_cast<float4 *>(o + stride + 4) = make_float4(ab[4], ab[5], ab[6], ab[7]);
}

// [Cout][9][CinP] (k_policy_wgrad's order, the four layers back to back) -> the module's [Cout][Cin][3][3]
struct WgUnpack {
    const float *src;
    float *dst[4];
    __host__ __device__ static constexpr int cout(int l) { return l == 0 ? 32 : l == 1 ? 64 : 128; }
    __host__ __device__ static constexpr int cin(int l) { return l == 0 ? 12 : l == 1 ? 32 : l == 2 ? 64 : 128; }
    __host__ __device__ static constexpr int cinp(int l) { return l == 0 ? 16 : cin(l); }
    __host__ __device__ static constexpr int size_mod(int l) { return cout(l) * cin(l) * 9; }
    __host__ __device__ static constexpr int size_src(int l) { return cout(l) * cinp(l) * 9; }
    __host__ __device__ static constexpr int total() { return size_mod(0) + size_mod(1) + size_mod(2) + size_mod(3); }
};

__global__ __launch_bounds__(256) void k_wgrad_to_module(WgUnpack u)
{
    int e = blockIdx.x * 256 + threadIdx.x;                              // one element of a module tensor
    if (e >= WgUnpack::total()) return;
    int l = 0, soff = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (e >= WgUnpack::size_mod(l)) { e -= WgUnpack::size_mod(l); soff += WgUnpack::size_src(l); ++l; }
    const int cin = WgUnpack::cin(l), cinp = WgUnpack::cinp(l);
    const int t = e % 9, ci = (e / 9) % cin, co = e / (9 * cin);
    float *dst = l == 0 ? u.dst[0] : l == 1 ? u.dst[1] : l == 2 ? u.dst[2] : u.dst[3];
    dst[e] = u.src[soff + ((size_t)co * 9 + t) * cinp + ci];
}

// All four layers' LayerNorm gradients in ONE launch, added straight into the module's tensors ([C][10][15] order: the parameters'
// .grad): no per-slice partial buffers, no reduction / transposition launches afterwards (round 2: 4 launches + 4 torch reductions
// + 8 layout copies per minibatch).  Item = (layer, pixel, 8 channels); blockIdx.y strides over the samples, four loads of each
// tensor in flight per thread; 16 f32 atomics per thread at the end (destinations zeroed by the caller).
// dfeat / ln3 (ABI 8, both or neither): the top layer's dy is not read but recomputed as mask * dfeat, mask = [x^ g + b > 0] from the
// packed fp16 LayerNorm parameters -- the very expression (and rounding: dy is dfeat or 0) of the backward kernel's pass A.
struct PgAll { const bf16 *dy[4]; const bf16 *xh[4]; float *gw[4]; float *gb[4]; const bf16 *dfeat; const _Float16 *ln3; };

__global__ __launch_bounds__(256) void k_policy_param_grads_all(PgAll a, int N)
{
    constexpr int I0 = P * 4, I1 = I0 + P * 8, I2 = I1 + P * 16, I3 = I2 + P * 16;     // item prefix sums: 32, 64, 128, 128 channels
    const int it = blockIdx.x * 256 + threadIdx.x;
    if (it >= I3) return;
    const int layer = it < I0 ? 0 : it < I1 ? 1 : it < I2 ? 2 : 3;
    const int base = layer == 0 ? 0 : layer == 1 ? I0 : layer == 2 ? I1 : I2;
    const int c8 = layer == 0 ? 4 : layer == 1 ? 8 : 16;                                // 16-byte chunks per pixel
    const int q = it - base, p = q / c8, part = q - p * c8;
    const bool top = layer == 3 && a.dfeat != nullptr;
    const bf16 *dyl = layer == 0 ? a.dy[0] : layer == 1 ? a.dy[1] : layer == 2 ? a.dy[2] : top ? a.dfeat : a.dy[3];
    const bf16 *xhl = layer == 0 ? a.xh[0] : layer == 1 ? a.xh[1] : layer == 2 ? a.xh[2] : a.xh[3];
    const size_t stride = (size_t)P * c8 * 8;                                           // elements per sample
    float aw[8], ab[8], mg[8], mb[8];                                                   // mg, mb: the item's mask parameters (0, 1 = keep everything)
#pragma unroll
    for (int j = 0; j < 8; ++j) { aw[j] = ab[j] = 0.0f; mg[j] = 0.0f; mb[j] = 1.0f; }
    if (top) {
        const f16x8 l0 = *reinterpret_cast<const f16x8 *>(a.ln3 + (size_t)q * 16), l1 = *reinterpret_cast<const f16x8 *>(a.ln3 + (size_t)q * 16 + 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mg[j] = (float)(j < 4 ? l0[j] : l1[j - 4]);
            mb[j] = (float)(j < 4 ? l0[4 + j] : l1[j]);
        }
    }
    constexpr int U = 4;
    int n = blockIdx.y;
    for (; n + (U - 1) * (int)gridDim.y < N; n += U * gridDim.y) {
        bf16x8 d[U], x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t o = (size_t)(n + u * gridDim.y) * stride + (size_t)q * 8;
            d[u] = *reinterpret_cast<const bf16x8 *>(dyl + o);
            x[u] = *reinterpret_cast<const bf16x8 *>(xhl + o);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xf = (float)x[u][j];
                const float df = fmaf(xf, mg[j], mb[j]) > 0.0f ? (float)d[u][j] : 0.0f;
                aw[j] = fmaf(df, xf, aw[j]);
                ab[j] += df;
            }
    }
    for (; n < N; n += gridDim.y) {
        const size_t o = (size_t)n * stride + (size_t)q * 8;
        const bf16x8 d = *reinterpret_cast<const bf16x8 *>(dyl + o);
        const bf16x8 x = *reinterpret_cast<const bf16x8 *>(xhl + o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xf = (float)x[j];
            const float df = fmaf(xf, mg[j], mb[j]) > 0.0f ? (float)d[j] : 0.0f;
            aw[j] = fmaf(df, xf, aw[j]);
            ab[j] += df;
        }
    }
    float *gw = layer == 0 ? a.gw[0] : layer == 1 ? a.gw[1] : layer == 2 ? a.gw[2] : a.gw[3];
    float *gb = layer == 0 ? a.gb[0] : layer == 1 ? a.gb[1] : layer == 2 ? a.gb[2] : a.gb[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = part * 8 + j;
        atomicAdd(gw + (size_t)c * P + p, aw[j]);
        atomicAdd(gb + (size_t)c * P + p, ab[j]);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Host glue of the training path as kernels (the parameters change with every optimizer step, so they are re-packed per
// minibatch: as torch index arithmetic this was ~60 small launches per minibatch).
// pack_conv: f32 [Cout][Cin][3][3] -> bf16 A fragments [Cout'/16][9*CinP/32][64][8] (inversus_hip.h, inv_policy_layer.w).
//   transposed = 0:  A[co][k = tap * CinP + ci] = W[co][ci][tap]                       (forward; Cout' = Cout, K side = Cin padded to CinP)
//   transposed = 1:  A[ci][k = tap * Cout + co] = W[co][ci][8 - tap]                   (data gradient; Cout' = Cin, K side = Cout)
__global__ __launch_bounds__(256) void k_pack_conv(const float *__restrict__ w, int cout, int cin, int cinp, int transposed, bf16 *__restrict__ out)
{
    const int m_rows = transposed ? cin : cout, kc = transposed ? cout : cinp, K = 9 * kc;
    const long long total = (long long)m_rows * K;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
    const long long frag = e >> 9;                                       // mt * (K / 32) + ks
    const int ksn = K / 32, ks = (int)(frag % ksn), mt = (int)(frag / ksn);
    const int row = 16 * mt + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + j;
    const int tap = k / kc, c = k - tap * kc;
    float v = 0.0f;
    if (!transposed) { if (c < cin) v = w[((size_t)row * cin + c) * 9 + tap]; }
    else v = w[((size_t)c * cin + row) * 9 + (8 - tap)];
    out[e] = (bf16)v;
}

// pack_ln: LayerNorm([C,10,15]) weight / bias f32 -> fp16 [150][C/4][weight x4 | bias x4]
__global__ __launch_bounds__(256) void k_pack_ln(const float *__restrict__ w, const float *__restrict__ b, int cout, int pixels, _Float16 *__restrict__ out)
{
    const int e = blockIdx.x * 256 + threadIdx.x;                        // one (pixel, channel)
    if (e >= pixels * cout) return;
    const int p = e / cout, c = e - p * cout;
    _Float16 *o = out + ((size_t)p * cout + (c & ~3)) * 2 + (c & 3);
    o[0] = (_Float16)w[(size_t)c * pixels + p];
    o[4] = (_Float16)b[(size_t)c * pixels + p];
}

// obs_to_x0: the observation as the weight-gradient kernel's layer-1 operand, bf16 channels-last [N][150][16] (12 planes + 4 zero channels),
// from f32 planes [N][12][150] or from the packed bitstrings [N][57]
__global__ __launch_bounds__(256) void k_obs_to_x0(const void *__restrict__ obs, int is_bits, bf16 *__restrict__ x0, int N)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;      // one (sample, pixel)
    if (e >= (long long)N * P) return;
    const int n = (int)(e / P), p = (int)(e - (long long)n * P);
    bf16x8 lo, hi;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float v = 0.0f;
        if (c < 12) {
            if (is_bits) {
                const int bit = c * P + p;
                v = (float)((reinterpret_cast<const uint32_t *>(obs)[(size_t)n * OBS_WORDS + (bit >> 5)] >> (bit & 31)) & 1u);
            } else {
                v = reinterpret_cast<const float *>(obs)[((size_t)n * 12 + c) * P + p];
            }
        }
        if (c < 8) lo[c] = (bf16)v; else hi[c - 8] = (bf16)v;
    }
    bf16x8 *dst = reinterpret_cast<bf16x8 *>(x0 + (size_t)e * 16);
    dst[0] = lo;
    dst[1] = hi;
}

// ------------------------------------------------------------------------------------------------------------------
// Backward of the BANDED trunk (32 x 32 board, BASELINE config C5 with the fused update).  A sample's layer does not fit
// LDS (262 KB of bf16 at 128 channels), so the backward runs LAYER BY LAYER over the whole batch, four kinds of kernels per
// layer, all on the tensors the SAVE forward wrote (z, statistics) and on channels-last bf16 tensors [N][P][C]:
//   k_band_ln_bwd      one workgroup per sample, two row passes over (da, z):  sums of dx^ = da [x^ w + b > 0] w and dx^ x^ (and the
//                      activation a = relu(x^ w + b) written on the way: the x operand of the layer above's weight gradient),
//                      then dz = rstd (dx^ - mean(dx^) - x^ mean(dx^ x^));  x^ = z rstd + shift is recomputed from the saved z
//                      with the forward's own expressions, so the ReLU mask is the forward's; dy is never stored
//   k_band_conv_t      da_below = convT(dz) (+ dz for the residual of layer 4): the banded implicit GEMM of the forward
//                      (band_conv) on the transposed weights, input bands copied raw
//   k_band_act         a = relu(LN(z)) as a tensor of its own (kept for callers that want it; the training path gets it from k_band_ln_bwd)
//   k_policy_wgrad<.., WgGeoBand>  weight / bias gradients, one band of 4 rows per work unit
//   k_band_param_grads LayerNorm weight / bias gradients: per-slice partial sums of dy x^ and dy over the samples, dy and x^ recomputed
//                      from (da, z)
// HBM-bound passes and MFMA-bound GEMMs alternate as separate launches (no cross-layer fusion: 4.3 GB of activations per
// 16 384-sample layer cannot stay on chip); what matters here is that C5 TRAINS through kernels of this library at all --
// the alternative is torch autograd over the f32 module (measured in bench.py extras beside it).

// dy is optional (tests): nothing downstream needs it as a tensor any more -- pass B and k_band_param_grads recompute the ReLU mask
// from (da, z, LayerNorm parameters), which costs arithmetic in HBM-bound kernels and saves one tensor write and one read per layer.
// act (layers 0..2): a = relu(x^ w + b), the x operand of the NEXT layer's weight-gradient kernel, written by pass A on the way
// (it forms that expression anyway for the mask) instead of by a pass of its own over z.
template <int C>
__global__ __launch_bounds__(THREADS) void k_band_ln_bwd(const bf16 *__restrict__ da, const bf16 *__restrict__ z, const float *__restrict__ stats,
                                                          int layer, const void *__restrict__ ln_packed, bf16 *__restrict__ dy, bf16 *__restrict__ dz,
                                                          bf16 *__restrict__ act, int pixels, int N)
{
    __shared__ float red[2 * WAVES];
    constexpr int CH = 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int items = pixels * (C / 8);
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        const float rstd = stats[(size_t)n * 8 + 2 * layer], shift = stats[(size_t)n * 8 + 2 * layer + 1];
        const bf16x8 *da8 = reinterpret_cast<const bf16x8 *>(da + (size_t)n * pixels * C);
        const bf16x8 *z8 = reinterpret_cast<const bf16x8 *>(z + (size_t)n * pixels * C);
        bf16x8 *dy8 = dy ? reinterpret_cast<bf16x8 *>(dy + (size_t)n * pixels * C) : nullptr;
        bf16x8 *dz8 = reinterpret_cast<bf16x8 *>(dz + (size_t)n * pixels * C);
        bf16x8 *act8 = act ? reinterpret_cast<bf16x8 *>(act + (size_t)n * pixels * C) : nullptr;
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll 1
        for (int q0 = tid; q0 < items; q0 += CH * THREADS) {                 // pass A: the two sums (+ the activation tensor)
            bf16x8 a[CH], zz[CH];
            f16x8 l0[CH], l1[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int q = min(q0 + j * THREADS, items - 1);
                a[j] = da8[q]; zz[j] = z8[q];
                const _Float16 *lq = reinterpret_cast<const _Float16 *>(ln_packed) + (size_t)q * 16;
                l0[j] = *reinterpret_cast<const f16x8 *>(lq);
                l1[j] = *reinterpret_cast<const f16x8 *>(lq + 8);
            }
            POL_PIN();
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int q = q0 + j * THREADS;
                if (q < items) {
                    bf16x8 av;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float g = (float)(e < 4 ? l0[j][e] : l1[j][e - 4]), bb = (float)(e < 4 ? l0[j][4 + e] : l1[j][e]);
                        const float xh = fmaf((float)zz[j][e], rstd, shift);
                        const float pre = fmaf(xh, g, bb);                    // ln_relu8's expression: the forward's mask
                        av[e] = (bf16)fmaxf(pre, 0.0f);
                        const float dxh = pre > 0.0f ? (float)a[j][e] * g : 0.0f;
                        s1 += dxh;
                        s2 = fmaf(dxh, xh, s2);
                    }
                    if (act8) act8[q] = av;
                }
            }
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) { red[wave] = s1; red[WAVES + wave] = s2; }
        __syncthreads();
        float S1 = 0.0f, S2 = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { S1 += red[w]; S2 += red[WAVES + w]; }
        const float inv_cnt = 1.0f / ((float)C * (float)pixels);
        const float m1 = S1 * inv_cnt, m2 = S2 * inv_cnt;
#pragma unroll 1
        for (int q0 = tid; q0 < items; q0 += CH * THREADS) {                 // pass B: the same items again
            bf16x8 a[CH], zz[CH];
            f16x8 l0[CH], l1[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int q = min(q0 + j * THREADS, items - 1);
                a[j] = da8[q]; zz[j] = z8[q];
                const _Float16 *lq = reinterpret_cast<const _Float16 *>(ln_packed) + (size_t)q * 16;
                l0[j] = *reinterpret_cast<const f16x8 *>(lq);
                l1[j] = *reinterpret_cast<const f16x8 *>(lq + 8);
            }
            POL_PIN();
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int q = q0 + j * THREADS;
                if (q < items) {
                    bf16x8 o, d;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float g = (float)(e < 4 ? l0[j][e] : l1[j][e - 4]), bb = (float)(e < 4 ? l0[j][4 + e] : l1[j][e]);
                        const float xh = fmaf((float)zz[j][e], rstd, shift);
                        d[e] = fmaf(xh, g, bb) > 0.0f ? a[j][e] : (bf16)0.0f;
                        o[e] = (bf16)(rstd * ((float)d[e] * g - m1 - xh * m2));
                    }
                    dz8[q] = o;
                    if (dy8) dy8[q] = d;
                }
            }
        }
        __syncthreads();                                                     // red[] is rewritten by the next sample
    }
}


// Round 4: the LayerNorm backward of a banded layer in TWO tensor passes over (da, z) instead of four, and the parameter sums with it:
//   k_band_ln_sums   pass A alone: per sample the sums of dx^ = da [x^ w + b > 0] w and of dx^ x^ (the same expressions, the same
//                    summation order as k_band_ln_bwd's pass A), written to sums [N][2]
//   k_band_ln_bwd2   ONE pass that turns (da, z, the sample's two sums) into dz AND the activation a = relu(x^ w + b) AND the layer's
//                    LayerNorm weight / bias gradient partial sums.  It is cut like k_band_param_grads, not like k_band_ln_bwd: a thread
//                    owns the same 8 channels of one pixel for EVERY sample of its slice (gridDim.y slices), so its LayerNorm
//                    parameters are loaded once, its parameter sums stay in registers, and the sample's scalars (rstd, shift, the two
//                    means) are wave-uniform loads; all tensor accesses are 16 B per thread, consecutive threads consecutive items.
// Per layer: 2 + 2 tensor reads and 2 writes instead of 2 + 2 + 2 reads and 2 writes (k_band_ln_bwd + k_band_param_grads); dz, the
// activation and the partial sums are bit-identical to the old path's.
template <int C>
__global__ __launch_bounds__(THREADS) void k_band_ln_sums(const bf16 *__restrict__ da, const bf16 *__restrict__ z, const float *__restrict__ stats,
                                                           int layer, const void *__restrict__ ln_packed, float *__restrict__ sums, int pixels, int N)
{
    __shared__ float red[2 * WAVES];
    constexpr int CH = 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int items = pixels * (C / 8);
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        const float rstd = stats[(size_t)n * 8 + 2 * layer], shift = stats[(size_t)n * 8 + 2 * layer + 1];
        const bf16x8 *da8 = reinterpret_cast<const bf16x8 *>(da + (size_t)n * pixels * C);
        const bf16x8 *z8 = reinterpret_cast<const bf16x8 *>(z + (size_t)n * pixels * C);
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll 1
        for (int q0 = tid; q0 < items; q0 += CH * THREADS) {
            bf16x8 a[CH], zz[CH];
            f16x8 l0[CH], l1[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int q = min(q0 + j * THREADS, items - 1);
                a[j] = da8[q]; zz[j] = z8[q];
                const _Float16 *lq = reinterpret_cast<const _Float16 *>(ln_packed) + (size_t)q * 16;
                l0[j] = *reinterpret_cast<const f16x8 *>(lq);
                l1[j] = *reinterpret_cast<const f16x8 *>(lq + 8);
            }
            POL_PIN();
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int q = q0 + j * THREADS;
                if (q < items) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float g = (float)(e < 4 ? l0[j][e] : l1[j][e - 4]), bb = (float)(e < 4 ? l0[j][4 + e] : l1[j][e]);
                        const float xh = fmaf((float)zz[j][e], rstd, shift);
                        const float pre = fmaf(xh, g, bb);
                        const float dxh = pre > 0.0f ? (float)a[j][e] * g : 0.0f;
                        s1 += dxh;
                        s2 = fmaf(dxh, xh, s2);
                    }
                }
            }
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) { red[wave] = s1; red[WAVES + wave] = s2; }
        __syncthreads();
        if (tid == 0) {
            float S1 = 0.0f, S2 = 0.0f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) { S1 += red[w]; S2 += red[WAVES + w]; }
            sums[(size_t)n * 2] = S1;
            sums[(size_t)n * 2 + 1] = S2;
        }
        __syncthreads();
    }
}

template <int C>
__global__ __launch_bounds__(256) void k_band_ln_bwd2(const bf16 *__restrict__ da, const bf16 *__restrict__ z, const float *__restrict__ stats,
                                                       int layer, const void *__restrict__ ln_packed, const float *__restrict__ sums,
                                                       bf16 *__restrict__ dz, bf16 *__restrict__ act, float *__restrict__ part, int E8, int N)
{
    const int e8 = blockIdx.x * 256 + threadIdx.x;
    if (e8 >= E8) return;
    float aw[8], ab[8], g[8], bb[8];
    {
        const _Float16 *lq = reinterpret_cast<const _Float16 *>(ln_packed) + (size_t)e8 * 16;
        const f16x8 l0 = *reinterpret_cast<const f16x8 *>(lq), l1 = *reinterpret_cast<const f16x8 *>(lq + 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            aw[j] = ab[j] = 0.0f;
            g[j] = (float)(j < 4 ? l0[j] : l1[j - 4]);
            bb[j] = (float)(j < 4 ? l0[4 + j] : l1[j]);
        }
    }
    const size_t stride = (size_t)E8 * 8;
    const float inv_cnt = 1.0f / (float)stride;                           // C * pixels elements per sample
    constexpr int U = 4;                                                  // samples in flight per thread
    for (int n0 = blockIdx.y; n0 < N; n0 += U * gridDim.y) {
        bf16x8 d[U], zz[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = min(n0 + u * (int)gridDim.y, N - 1);             // the tail re-loads the last sample and discards it
            const size_t o = (size_t)n * stride + (size_t)e8 * 8;
            d[u] = *reinterpret_cast<const bf16x8 *>(da + o);
            zz[u] = *reinterpret_cast<const bf16x8 *>(z + o);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = n0 + u * (int)gridDim.y;
            if (n < N) {
                const float rstd = stats[(size_t)n * 8 + 2 * layer], shift = stats[(size_t)n * 8 + 2 * layer + 1];
                const float m1 = sums[(size_t)n * 2] * inv_cnt, m2 = sums[(size_t)n * 2 + 1] * inv_cnt;
                bf16x8 o, av;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = fmaf((float)zz[u][j], rstd, shift);
                    const float pre = fmaf(xh, g[j], bb[j]);
                    av[j] = (bf16)fmaxf(pre, 0.0f);
                    const float df = pre > 0.0f ? (float)d[u][j] : 0.0f;
                    o[j] = (bf16)(rstd * (df * g[j] - m1 - xh * m2));
                    aw[j] = fmaf(df, xh, aw[j]);
                    ab[j] += df;
                }
                const size_t off = (size_t)n * stride + (size_t)e8 * 8;
                *reinterpret_cast<bf16x8 *>(dz + off) = o;
                if (act) *reinterpret_cast<bf16x8 *>(act + off) = av;
            }
        }
    }
    float *o = part + (size_t)blockIdx.y * 2 * stride + (size_t)e8 * 8;
    *reinterpret_cast<float4 *>(o) = make_float4(aw[0], aw[1], aw[2], aw[3]);
    *reinterpret_cast<float4 *>(o + 4) = make_float4(aw[4], aw[5], aw[6], aw[7]);
    *reinterpret_cast<float4 *>(o + stride) = make_float4(ab[0], ab[1], ab[2], ab[3]);
    *reinterpret_cast<float4 *>(o + stride + 4) = make_float4(ab[4], ab[5], ab[6], ab[7]);
}

// raw band loader: tile <- the band's R + 2 rows of a channels-last tensor [P][CSRC], zeros outside the board; rows carried over
// from the previous band like band_rows_norm's
template <class G, int R, int CSRC, int ROW0, int NROWS>
__device__ __forceinline__ void band_rows_raw(bf16 *tile, const bf16 *__restrict__ src, int band, bool carry)
{
    constexpr int WBc = G::PWB - 2, HBc = G::P / WBc, C8 = CSRC / 8, ROWI = WBc * C8, ITEMS = NROWS * ROWI, NIT = ITEMS / THREADS;
    static_assert(ITEMS % THREADS == 0 && NIT <= 8 && ((R - ROW0) * ROWI) % THREADS == 0, "whole passes; an item is carried or not by its index alone");
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int y0 = band * R + ROW0 - 1;
    bf16x8 v[NIT], old[NIT];
    int dsto[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int q = tid + j * THREADS;
        const int rr = q / ROWI, rem = q - rr * ROWI, x = rem / C8, part = rem - x * C8;
        const bool in = (unsigned)(y0 + rr) < (unsigned)HBc;
        dsto[j] = ((ROW0 + rr) * G::PWB + x + 1) * CSF + part * 8;
        if (ROW0 + (j * THREADS) / ROWI >= R && carry) old[j] = *reinterpret_cast<const bf16x8 *>(tile + dsto[j]);
        v[j] = *reinterpret_cast<const bf16x8 *>(src + (in ? ((size_t)(y0 + rr) * ROWI + rem) * 8 : 0));
        if (!in) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[j][e] = (bf16)0.0f;
        }
    }
    POL_PIN();
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        if (ROW0 + (j * THREADS) / ROWI >= R && carry) *reinterpret_cast<bf16x8 *>(tile + dsto[j] - R * G::PWB * CSF) = old[j];
        *reinterpret_cast<bf16x8 *>(tile + dsto[j]) = v[j];
    }
}

template <class G, int R, int CSRC>
__device__ __forceinline__ void band_load_raw(bf16 *tile, const bf16 *__restrict__ src, int band)
{
    if (band == 0) band_rows_raw<G, R, CSRC, 0, 2>(tile, src, band, false);
    band_rows_raw<G, R, CSRC, 2, R>(tile, src, band, band > 0);
}

// da_below[n] = convT(dz[n]) (+ dz[n]: RESIDUAL, the skip connection around conv4).  lp.w: the transposed packing of the layer's
// weights (inv_policy_pack_conv(transposed = 1)), lp.bias: zeros.
// SUMS (round 5): the write-back pass also accumulates what k_band_ln_sums would read da_below back for -- the sample's sums of
// dx^ = da [x^ w + b > 0] w and of dx^ x^ of the layer BELOW, from the bf16 chunks as they are written and the same expressions, so the
// mask is the forward's; only the order of the f32 additions differs.  One tensor read (da_below, 4.3 GB at 128 channels) and one
// launch less per layer.
template <int C>
struct LnSumsHook {
    const bf16x8 *z8;                 // the sample's z of the layer below
    const void *ln;                   // its packed LayerNorm parameters
    float rstd, shift, s1, s2;
    template <int NOUT> __device__ __forceinline__ void items(int first, int tid, const bf16x8 (&da)[NOUT])
    {
        bf16x8 zz[NOUT];
        f16x8 l0[NOUT], l1[NOUT];
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
            const int q = first + tid + j * THREADS;
            zz[j] = z8[q];
            const _Float16 *lq = reinterpret_cast<const _Float16 *>(ln) + (size_t)q * 16;
            l0[j] = *reinterpret_cast<const f16x8 *>(lq);
            l1[j] = *reinterpret_cast<const f16x8 *>(lq + 8);
        }
        POL_PIN();
#pragma unroll
        for (int j = 0; j < NOUT; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float g = (float)(e < 4 ? l0[j][e] : l1[j][e - 4]), bb = (float)(e < 4 ? l0[j][4 + e] : l1[j][e]);
                const float xh = fmaf((float)zz[j][e], rstd, shift);
                const float pre = fmaf(xh, g, bb);                        // ln_relu8's expression: the forward's mask
                const float dxh = pre > 0.0f ? (float)da[j][e] * g : 0.0f;
                s1 += dxh;
                s2 = fmaf(dxh, xh, s2);
            }
    }
};

struct BandSumsArgs { const bf16 *z; const float *stats; const void *ln; float *sums; int layer_below; };

template <int HB, int WB, int R, int CIN, int COUT, int MW, int NW, bool RESIDUAL, bool SPLIT, bool SUMS>
__global__ __launch_bounds__(THREADS, 2) void k_band_conv_t(inv_policy_layer lp, const bf16 *__restrict__ dz, bf16 *__restrict__ da_out, int N, BandSumsArgs sa)
{
    using G = BandGeo<HB, WB, R>;
    __shared__ __attribute__((aligned(16))) bf16 tile[G::NPOSB * CSF];
    __shared__ float red[2 * WAVES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)__builtin_amdgcn_readfirstlane((uint32_t)(tid >> 6));
    for (int i = tid; i < G::NPOSB * CSF / 8; i += THREADS) {
        bf16x8 zz;
#pragma unroll
        for (int j = 0; j < 8; ++j) zz[j] = (bf16)0.0f;
        reinterpret_cast<bf16x8 *>(&tile[0])[i] = zz;
    }
    __syncthreads();
    // SPLIT (COUT = 32): two row blocks x two halves of the band's tiles over the four waves, like layer 1 of the forward
    const int m0 = SPLIT ? (wave & 1) : wave * MW, n0 = SPLIT ? (wave >> 1) * NW : 0;
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        int lane_i = lane;
        asm volatile("" : "+v"(lane_i));
        const bf16 *src = dz + (size_t)n * G::P * CIN;
        bf16 *dst = da_out + (size_t)n * G::P * COUT;
        float s = 0.0f, ss = 0.0f;                                          // band_conv's statistics: unused here
        typename std::conditional<SUMS, LnSumsHook<COUT>, NoWriteBackHook>::type hook;
        if constexpr (SUMS) {
            hook.z8 = reinterpret_cast<const bf16x8 *>(sa.z + (size_t)n * G::P * COUT);
            hook.ln = sa.ln;
            hook.rstd = sa.stats[(size_t)n * 8 + 2 * sa.layer_below];
            hook.shift = sa.stats[(size_t)n * 8 + 2 * sa.layer_below + 1];
            hook.s1 = hook.s2 = 0.0f;
        }
#pragma unroll 1
        for (int band = 0; band < G::NB; ++band) {
            band_load_raw<G, R, CIN>(tile, src, band);
            __syncthreads();
            unsigned long long stamp_prev = 0;
            band_conv<G, R, CIN, COUT, MW, NW, RESIDUAL>(lp, tile, dst, band, lane_i, m0, n0, s, ss, stamp_prev, 24, hook);
        }
        if constexpr (SUMS) {
            const float s1 = wave_sum(hook.s1), s2 = wave_sum(hook.s2);
            if (lane == 0) { red[wave] = s1; red[WAVES + wave] = s2; }
            __syncthreads();
            if (tid == 0) {
                float S1 = 0.0f, S2 = 0.0f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) { S1 += red[w]; S2 += red[WAVES + w]; }
                sa.sums[(size_t)n * 2] = S1;
                sa.sums[(size_t)n * 2 + 1] = S2;
            }
            __syncthreads();                                                // red[] is rewritten by the next sample
        }
    }
}

// a = relu(LN(z)) for a whole tensor [N][P][C] (the forward applies it on the fly while it builds a band; the weight-gradient
// kernel wants it as its x operand)
template <int C>
__global__ __launch_bounds__(THREADS) void k_band_act(const bf16 *__restrict__ z, const float *__restrict__ stats, int layer,
                                                       const void *__restrict__ ln_packed, bf16 *__restrict__ act, int pixels, long long total_items)
{
    constexpr int CH = 4;
    const int items = pixels * (C / 8);
    for (long long g0 = (long long)blockIdx.x * THREADS * CH + threadIdx.x; g0 < total_items; g0 += (long long)gridDim.x * THREADS * CH) {
        bf16x8 zz[CH];
        f16x8 l0[CH], l1[CH];
        float rs[CH], sh[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const long long g = min(g0 + (long long)j * THREADS, total_items - 1);
            const long long n = g / items;
            const int q = (int)(g - n * items);
            zz[j] = reinterpret_cast<const bf16x8 *>(z)[g];
            const _Float16 *lq = reinterpret_cast<const _Float16 *>(ln_packed) + (size_t)q * 16;
            l0[j] = *reinterpret_cast<const f16x8 *>(lq);
            l1[j] = *reinterpret_cast<const f16x8 *>(lq + 8);
            rs[j] = stats[n * 8 + 2 * layer]; sh[j] = stats[n * 8 + 2 * layer + 1];
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const long long g = g0 + (long long)j * THREADS;
            if (g < total_items) reinterpret_cast<bf16x8 *>(act)[g] = ln_relu8(zz[j], l0[j], l1[j], rs[j], sh[j]);
        }
    }
}

// LayerNorm weight / bias gradients of a banded layer: like k_policy_param_grads, with everything recomputed from what the other
// kernels read anyway -- x^ = z rstd + shift from the saved z and the per-sample statistics, dy = da * [x^ w + b > 0] from the
// layer's incoming gradient (a thread owns the same 8 channels of one pixel for every sample, so its LayerNorm parameters are
// loaded once)
__global__ __launch_bounds__(256) void k_band_param_grads(const bf16 *__restrict__ da, const bf16 *__restrict__ z, const float *__restrict__ stats,
                                                           int layer, const void *__restrict__ ln_packed, float *__restrict__ part, int E8, int N)
{
    const int e8 = blockIdx.x * 256 + threadIdx.x;
    if (e8 >= E8) return;
    float aw[8], ab[8], g[8], bb[8];
    {
        const _Float16 *lq = reinterpret_cast<const _Float16 *>(ln_packed) + (size_t)e8 * 16;
        const f16x8 l0 = *reinterpret_cast<const f16x8 *>(lq), l1 = *reinterpret_cast<const f16x8 *>(lq + 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            aw[j] = ab[j] = 0.0f;
            g[j] = (float)(j < 4 ? l0[j] : l1[j - 4]);
            bb[j] = (float)(j < 4 ? l0[4 + j] : l1[j]);
        }
    }
    const size_t stride = (size_t)E8 * 8;
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const size_t o = (size_t)n * stride + (size_t)e8 * 8;
        const bf16x8 d = *reinterpret_cast<const bf16x8 *>(da + o);
        const bf16x8 zz = *reinterpret_cast<const bf16x8 *>(z + o);
        const float rstd = stats[(size_t)n * 8 + 2 * layer], shift = stats[(size_t)n * 8 + 2 * layer + 1];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = fmaf((float)zz[j], rstd, shift);
            const float df = fmaf(xh, g[j], bb[j]) > 0.0f ? (float)d[j] : 0.0f;
            aw[j] = fmaf(df, xh, aw[j]);
            ab[j] += df;
        }
    }
    float *o = part + (size_t)blockIdx.y * 2 * stride + (size_t)e8 * 8;
    *reinterpret_cast<float4 *>(o) = make_float4(aw[0], aw[1], aw[2], aw[3]);
    *reinterpret_cast<float4 *>(o + 4) = make_float4(aw[4], aw[5], aw[6], aw[7]);
    *reinterpret_cast<float4 *>(o + stride) = make_float4(ab[0], ab[1], ab[2], ab[3]);
    *reinterpret_cast<float4 *>(o + stride + 4) = make_float4(ab[4], ab[5], ab[6], ab[7]);
}

// obs_to_x0 for any board: bf16 channels-last [N][pixels][16] from f32 planes [N][12][pixels] or packed bitstrings [N][ceil(12 pixels / 32)]
__global__ __launch_bounds__(256) void k_obs_to_x0_any(const void *__restrict__ obs, int is_bits, bf16 *__restrict__ x0, int pixels, int N)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;      // one (sample, pixel)
    if (e >= (long long)N * pixels) return;
    const int n = (int)(e / pixels), p = (int)(e - (long long)n * pixels);
    const int words = (12 * pixels + 31) / 32;
    bf16x8 lo, hi;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float v = 0.0f;
        if (c < 12) {
            if (is_bits) {
                const int bit = c * pixels + p;
                v = (float)((reinterpret_cast<const uint32_t *>(obs)[(size_t)n * words + (bit >> 5)] >> (bit & 31)) & 1u);
            } else {
                v = reinterpret_cast<const float *>(obs)[((size_t)n * 12 + c) * pixels + p];
            }
        }
        if (c < 8) lo[c] = (bf16)v; else hi[c - 8] = (bf16)v;
    }
    bf16x8 *dst = reinterpret_cast<bf16x8 *>(x0 + (size_t)e * 16);
    dst[0] = lo;
    dst[1] = hi;
}

}  // namespace pol

#ifdef INV_POLICY_STAMPS
extern "C" int inv_policy_debug_stamps(unsigned long long *out16_host, int reset)
{
    if (hipMemcpyFromSymbol(out16_host, HIP_SYMBOL(pol::g_stamps), 128 * sizeof(unsigned long long)) != hipSuccess) return INV_ERR_HIP;
    if (reset) {
        unsigned long long z[128] = { 0 };
        if (hipMemcpyToSymbol(HIP_SYMBOL(pol::g_stamps), z, sizeof z) != hipSuccess) return INV_ERR_HIP;
    }
    return INV_OK;
}
#endif

static int trunk_grid(int N)
{
    const int cus = inv_cu_count();                  // of the current device (cached per device)
    return N < 2 * cus ? N : 2 * cus;               // two resident 256-thread workgroups per CU (55 KB of LDS each), persistent over samples
}

static int launch_status(const char *what)
{
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? INV_OK : inv_fail(INV_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}


extern "C" int inv_policy_pack_conv(const float *w, int32_t cout, int32_t cin, int32_t cin_pad, int32_t transposed, void *out_bf16, void *stream)
{
    const int kc = transposed ? cout : cin_pad, rows = transposed ? cin : cout;
    if (!w || !out_bf16 || cout < 1 || cin < 1 || cin_pad < cin || (rows & 15) || ((9 * kc) & 31))
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_pack_conv: NULL tensor or shapes that do not tile into 16 x 32 fragments");
    const long long total = (long long)rows * 9 * kc;
    hipLaunchKernelGGL(pol::k_pack_conv, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (int)cout, (int)cin, (int)cin_pad,
                       (int)(transposed != 0), reinterpret_cast<pol::bf16 *>(out_bf16));
    return launch_status("inv_policy_pack_conv");
}

extern "C" int inv_policy_pack_ln(const float *weight, const float *bias, int32_t cout, int32_t pixels, void *out_f16, void *stream)
{
    if (!weight || !bias || !out_f16 || cout < 4 || (cout & 3) || pixels < 1) return inv_fail(INV_ERR_BAD_ARG, "inv_policy_pack_ln: NULL tensor, cout not a multiple of 4 or pixels < 1");
    hipLaunchKernelGGL(pol::k_pack_ln, dim3((pixels * cout + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight, bias, (int)cout, (int)pixels,
                       reinterpret_cast<_Float16 *>(out_f16));
    return launch_status("inv_policy_pack_ln");
}

extern "C" int inv_policy_obs_to_x0(const void *obs, int32_t obs_is_bits, void *x0_bf16, int32_t N, void *stream)
{
    if (!obs || !x0_bf16 || N < 0) return inv_fail(INV_ERR_BAD_ARG, "inv_policy_obs_to_x0: NULL tensor or N < 0");
    if (N == 0) return INV_OK;
    const long long total = (long long)N * pol::P;
    hipLaunchKernelGGL(pol::k_obs_to_x0, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obs, (int)(obs_is_bits != 0),
                       reinterpret_cast<pol::bf16 *>(x0_bf16), (int)N);
    return launch_status("inv_policy_obs_to_x0");
}

static bool layers_ok(const inv_policy_params *params)
{
    for (int l = 0; l < 4; ++l)
        if (!params->layer[l].w || !params->layer[l].bias || !params->layer[l].ln) return false;
    return true;
}

static int launch_trunk(const inv_policy_params *params, const void *obs, bool bits, void *feat_bf16, int32_t N, void *stream)
{
    if (!params || !obs || !feat_bf16 || N < 0 || !layers_ok(params))
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_trunk: NULL params / obs / feat, a NULL layer tensor, or N < 0");
    if (N == 0) return INV_OK;
#if defined(POL_BAND_ONE_PER_CU) && defined(INV_TOOLS_BUILD)
    const int grid = (trunk_grid(N) + 1) / 2, dyn_lds = 49152;               // concurrency probe: one workgroup per CU (as for the banded kernel)
#else
    const int grid = trunk_grid(N), dyn_lds = 0;
#endif
    inv_policy_saved none;
    memset(&none, 0, sizeof none);
    if (bits)
        hipLaunchKernelGGL((pol::k_policy_trunk<true, false>), dim3(grid), dim3(pol::THREADS), dyn_lds, (hipStream_t)stream, *params, obs,
                           reinterpret_cast<pol::bf16 *>(feat_bf16), (int)N, none);
    else
        hipLaunchKernelGGL((pol::k_policy_trunk<false, false>), dim3(grid), dim3(pol::THREADS), dyn_lds, (hipStream_t)stream, *params, obs,
                           reinterpret_cast<pol::bf16 *>(feat_bf16), (int)N, none);
    return launch_status("inv_policy_trunk");
}


extern "C" int64_t inv_policy_trunk_banded_scratch_bytes(int32_t height, int32_t width)
{
    if (height != 32 || width != 32) return 0;
    return (int64_t)2 * inv_cu_count() * (int64_t)height * width * 128 * 2;              // 2 workgroups per CU, one [P][128] bf16 buffer each
}

extern "C" int inv_policy_trunk_banded(const inv_policy_params *params, const void *obs, int32_t obs_is_bits, int32_t height, int32_t width,
                                       void *feat_bf16, void *scratch, int64_t scratch_bytes, int32_t N, void *stream)
{
    if (!params || !obs || !feat_bf16 || !scratch || N < 0 || !layers_ok(params))
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_trunk_banded: NULL params / obs / feat / scratch, a NULL layer tensor, or N < 0");
    if (height != 32 || width != 32) return inv_fail(INV_ERR_BAD_DIMS, "inv_policy_trunk_banded: built for the 32 x 32 board (got %d x %d)", (int)width, (int)height);
    if (scratch_bytes < inv_policy_trunk_banded_scratch_bytes(height, width)) return inv_fail(INV_ERR_BAD_ARG, "inv_policy_trunk_banded: scratch too small");
    if (N == 0) return INV_OK;
#if defined(POL_BAND_ONE_PER_CU) && defined(INV_TOOLS_BUILD)
    // concurrency probe (tools/band_knock_ab.sh): 48 KiB of unused dynamic LDS leave room for ONE workgroup per CU, half the grid
    const int grid = (trunk_grid(N) + 1) / 2, dyn_lds = 49152;
#else
    const int grid = trunk_grid(N), dyn_lds = 0;
#endif
    inv_policy_band_saved none;
    memset(&none, 0, sizeof none);
    if (obs_is_bits)
        hipLaunchKernelGGL((pol::k_policy_trunk_banded<32, 32, 4, true>), dim3(grid), dim3(pol::THREADS), dyn_lds, (hipStream_t)stream, *params, obs,
                           reinterpret_cast<pol::bf16 *>(feat_bf16), reinterpret_cast<pol::bf16 *>(scratch), (int)N, none);
    else
        hipLaunchKernelGGL((pol::k_policy_trunk_banded<32, 32, 4, false>), dim3(grid), dim3(pol::THREADS), dyn_lds, (hipStream_t)stream, *params, obs,
                           reinterpret_cast<pol::bf16 *>(feat_bf16), reinterpret_cast<pol::bf16 *>(scratch), (int)N, none);
    return launch_status("inv_policy_trunk_banded");
}

// ---- BASELINE config C5 with the fused update: training forward and the layer-by-layer backward of the banded trunk (32 x 32)
static bool band_dims_ok(int32_t height, int32_t width) { return height == 32 && width == 32; }
static const int kBandC[4] = { 32, 64, 128, 128 };

extern "C" int inv_policy_trunk_banded_train_fwd(const inv_policy_params *params, const void *obs, int32_t obs_is_bits, int32_t height, int32_t width,
                                                 void *feat_bf16, const inv_policy_band_saved *saved, int32_t N, void *stream)
{
    if (!params || !obs || !feat_bf16 || !saved || N < 0 || !layers_ok(params) || !saved->stats || !saved->z[0] || !saved->z[1] || !saved->z[2] || !saved->z[3])
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_trunk_banded_train_fwd: NULL params / obs / feat / saved tensor, or N < 0");
    if (!band_dims_ok(height, width)) return inv_fail(INV_ERR_BAD_DIMS, "inv_policy_trunk_banded_train_fwd: built for the 32 x 32 board (got %d x %d)", (int)width, (int)height);
    if (N == 0) return INV_OK;
    const int grid = trunk_grid(N);
    if (obs_is_bits)
        hipLaunchKernelGGL((pol::k_policy_trunk_banded<32, 32, 4, true, true>), dim3(grid), dim3(pol::THREADS), 0, (hipStream_t)stream, *params, obs,
                           reinterpret_cast<pol::bf16 *>(feat_bf16), (pol::bf16 *)nullptr, (int)N, *saved);
    else
        hipLaunchKernelGGL((pol::k_policy_trunk_banded<32, 32, 4, false, true>), dim3(grid), dim3(pol::THREADS), 0, (hipStream_t)stream, *params, obs,
                           reinterpret_cast<pol::bf16 *>(feat_bf16), (pol::bf16 *)nullptr, (int)N, *saved);
    return launch_status("inv_policy_trunk_banded_train_fwd");
}

extern "C" int inv_policy_band_ln_bwd(int32_t layer, int32_t height, int32_t width, const void *da_bf16, const void *z_bf16, const float *stats,
                                      const void *ln_packed, void *dy_bf16, void *dz_bf16, void *act_bf16, int32_t N, void *stream)
{
    if (!da_bf16 || !z_bf16 || !stats || !ln_packed || !dz_bf16 || N < 0 || layer < 0 || layer > 3)
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_band_ln_bwd: NULL tensor, N < 0 or layer outside 0..3 (dy and act may be NULL)");
    if (!band_dims_ok(height, width)) return inv_fail(INV_ERR_BAD_DIMS, "inv_policy_band_ln_bwd: built for the 32 x 32 board");
    if (N == 0) return INV_OK;
    const int pixels = height * width, grid = N < 4 * inv_cu_count() ? N : 4 * inv_cu_count();
    using pol::bf16;
#define INV_LNB(C_) hipLaunchKernelGGL((pol::k_band_ln_bwd<C_>), dim3(grid), dim3(pol::THREADS), 0, (hipStream_t)stream, reinterpret_cast<const bf16 *>(da_bf16), \
        reinterpret_cast<const bf16 *>(z_bf16), stats, (int)layer, ln_packed, reinterpret_cast<bf16 *>(dy_bf16), reinterpret_cast<bf16 *>(dz_bf16), \
        reinterpret_cast<bf16 *>(act_bf16), pixels, (int)N)
    if (kBandC[layer] == 32) INV_LNB(32); else if (kBandC[layer] == 64) INV_LNB(64); else INV_LNB(128);
#undef INV_LNB
    return launch_status("inv_policy_band_ln_bwd");
}


extern "C" int inv_policy_band_ln_sums(int32_t layer, int32_t height, int32_t width, const void *da_bf16, const void *z_bf16, const float *stats,
                                       const void *ln_packed, float *sums, int32_t N, void *stream)
{
    if (!da_bf16 || !z_bf16 || !stats || !ln_packed || !sums || N < 0 || layer < 0 || layer > 3)
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_band_ln_sums: NULL tensor, N < 0 or layer outside 0..3");
    if (!band_dims_ok(height, width)) return inv_fail(INV_ERR_BAD_DIMS, "inv_policy_band_ln_sums: built for the 32 x 32 board");
    if (N == 0) return INV_OK;
    const int pixels = height * width, grid = N < 4 * inv_cu_count() ? N : 4 * inv_cu_count();
    using pol::bf16;
#define INV_LNS(C_) hipLaunchKernelGGL((pol::k_band_ln_sums<C_>), dim3(grid), dim3(pol::THREADS), 0, (hipStream_t)stream, reinterpret_cast<const bf16 *>(da_bf16), \
        reinterpret_cast<const bf16 *>(z_bf16), stats, (int)layer, ln_packed, sums, pixels, (int)N)
    if (kBandC[layer] == 32) INV_LNS(32); else if (kBandC[layer] == 64) INV_LNS(64); else INV_LNS(128);
#undef INV_LNS
    return launch_status("inv_policy_band_ln_sums");
}

extern "C" int inv_policy_band_ln_bwd2(int32_t layer, int32_t height, int32_t width, const void *da_bf16, const void *z_bf16, const float *stats,
                                       const void *ln_packed, const float *sums, void *dz_bf16, void *act_bf16, float *partial, int32_t slices,
                                       int32_t N, void *stream)
{
    if (!da_bf16 || !z_bf16 || !stats || !ln_packed || !sums || !dz_bf16 || !partial || N < 0 || slices < 1 || layer < 0 || layer > 3)
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_band_ln_bwd2: NULL tensor, N < 0, slices < 1 or layer outside 0..3 (act may be NULL)");
    if (!band_dims_ok(height, width)) return inv_fail(INV_ERR_BAD_DIMS, "inv_policy_band_ln_bwd2: built for the 32 x 32 board");
    const int e8 = height * width * kBandC[layer] / 8, bx = (e8 + 255) / 256;
    using pol::bf16;
#define INV_LNB2(C_) hipLaunchKernelGGL((pol::k_band_ln_bwd2<C_>), dim3(bx, slices), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const bf16 *>(da_bf16), \
        reinterpret_cast<const bf16 *>(z_bf16), stats, (int)layer, ln_packed, sums, reinterpret_cast<bf16 *>(dz_bf16), reinterpret_cast<bf16 *>(act_bf16), \
        partial, e8, (int)N)
    if (kBandC[layer] == 32) INV_LNB2(32); else if (kBandC[layer] == 64) INV_LNB2(64); else INV_LNB2(128);
#undef INV_LNB2
    return launch_status("inv_policy_band_ln_bwd2");
}

template <bool SUMS>
static int launch_band_conv_t(int32_t layer, const void *wt_packed, const float *zero_bias, const void *dz_bf16, void *da_below_bf16, int32_t N, void *stream,
                              const pol::BandSumsArgs &sa)
{
    const int grid = trunk_grid(N);
    inv_policy_layer lp;
    lp.w = wt_packed; lp.bias = zero_bias; lp.ln = nullptr;
    using pol::bf16;
    using G = pol::BandGeo<32, 32, 4>;
    const bf16 *src = reinterpret_cast<const bf16 *>(dz_bf16);
    bf16 *dst = reinterpret_cast<bf16 *>(da_below_bf16);
    if (layer == 3)        // dz4 [128] -> da3 [128], + dz4 (residual)
        hipLaunchKernelGGL((pol::k_band_conv_t<32, 32, 4, 128, 128, 2, G::NT, true, false, SUMS>), dim3(grid), dim3(pol::THREADS), 0, (hipStream_t)stream, lp, src, dst, (int)N, sa);
    else if (layer == 2)   // dz3 [128] -> da2 [64]
        hipLaunchKernelGGL((pol::k_band_conv_t<32, 32, 4, 128, 64, 1, G::NT, false, false, SUMS>), dim3(grid), dim3(pol::THREADS), 0, (hipStream_t)stream, lp, src, dst, (int)N, sa);
    else                   // dz2 [64] -> da1 [32]
        hipLaunchKernelGGL((pol::k_band_conv_t<32, 32, 4, 64, 32, 1, G::NT / 2, false, true, SUMS>), dim3(grid), dim3(pol::THREADS), 0, (hipStream_t)stream, lp, src, dst, (int)N, sa);
    return launch_status(SUMS ? "inv_policy_band_conv_t_sums" : "inv_policy_band_conv_t");
}

extern "C" int inv_policy_band_conv_t(int32_t layer, int32_t height, int32_t width, const void *wt_packed, const float *zero_bias,
                                      const void *dz_bf16, void *da_below_bf16, int32_t N, void *stream)
{
    if (!wt_packed || !zero_bias || !dz_bf16 || !da_below_bf16 || N < 0 || layer < 1 || layer > 3)
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_band_conv_t: NULL tensor, N < 0 or layer outside 1..3 (layer 0 has no data gradient)");
    if (!band_dims_ok(height, width)) return inv_fail(INV_ERR_BAD_DIMS, "inv_policy_band_conv_t: built for the 32 x 32 board");
    if (N == 0) return INV_OK;
    pol::BandSumsArgs none;
    memset(&none, 0, sizeof none);
    return launch_band_conv_t<false>(layer, wt_packed, zero_bias, dz_bf16, da_below_bf16, N, stream, none);
}

extern "C" int inv_policy_band_conv_t_sums(int32_t layer, int32_t height, int32_t width, const void *wt_packed, const float *zero_bias,
                                           const void *dz_bf16, void *da_below_bf16, const void *z_below_bf16, const float *stats,
                                           const void *ln_below_packed, float *sums_below, int32_t N, void *stream)
{
    if (!wt_packed || !zero_bias || !dz_bf16 || !da_below_bf16 || !z_below_bf16 || !stats || !ln_below_packed || !sums_below || N < 0 || layer < 1 || layer > 3)
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_band_conv_t_sums: NULL tensor, N < 0 or layer outside 1..3 (layer 0 has no data gradient)");
    if (!band_dims_ok(height, width)) return inv_fail(INV_ERR_BAD_DIMS, "inv_policy_band_conv_t_sums: built for the 32 x 32 board");
    if (N == 0) return INV_OK;
    pol::BandSumsArgs sa;
    sa.z = reinterpret_cast<const pol::bf16 *>(z_below_bf16); sa.stats = stats; sa.ln = ln_below_packed; sa.sums = sums_below; sa.layer_below = layer - 1;
    return launch_band_conv_t<true>(layer, wt_packed, zero_bias, dz_bf16, da_below_bf16, N, stream, sa);
}

extern "C" int inv_policy_band_act(int32_t layer, int32_t height, int32_t width, const void *z_bf16, const float *stats, const void *ln_packed,
                                   void *act_bf16, int32_t N, void *stream)
{
    if (!z_bf16 || !stats || !ln_packed || !act_bf16 || N < 0 || layer < 0 || layer > 2)
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_band_act: NULL tensor, N < 0 or layer outside 0..2");
    if (!band_dims_ok(height, width)) return inv_fail(INV_ERR_BAD_DIMS, "inv_policy_band_act: built for the 32 x 32 board");
    if (N == 0) return INV_OK;
    const int pixels = height * width, C_ = kBandC[layer];
    const long long total = (long long)N * pixels * (C_ / 8);
    long long blocks = (total + pol::THREADS * 4 - 1) / (pol::THREADS * 4);
    if (blocks > 16 * (long long)inv_cu_count()) blocks = 16 * (long long)inv_cu_count();
    using pol::bf16;
#define INV_ACT(C__) hipLaunchKernelGGL((pol::k_band_act<C__>), dim3((unsigned)blocks), dim3(pol::THREADS), 0, (hipStream_t)stream, reinterpret_cast<const bf16 *>(z_bf16), \
        stats, (int)layer, ln_packed, reinterpret_cast<bf16 *>(act_bf16), pixels, total)
    if (C_ == 32) INV_ACT(32); else if (C_ == 64) INV_ACT(64); else INV_ACT(128);
#undef INV_ACT
    return launch_status("inv_policy_band_act");
}

extern "C" int inv_policy_band_param_grads(int32_t layer, const void *da_bf16, const void *z_bf16, const float *stats, const void *ln_packed,
                                           float *partial, int32_t slices, int32_t elems_per_sample, int32_t N, void *stream)
{
    if (!da_bf16 || !z_bf16 || !stats || !ln_packed || !partial || N < 0 || slices < 1 || elems_per_sample <= 0 || (elems_per_sample & 7) || layer < 0 || layer > 3)
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_band_param_grads: NULL tensor, N < 0, slices < 1, layer outside 0..3 or elems_per_sample not a positive multiple of 8");
    const int e8 = elems_per_sample / 8, bx = (e8 + 255) / 256;
    hipLaunchKernelGGL(pol::k_band_param_grads, dim3(bx, slices), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const pol::bf16 *>(da_bf16),
                       reinterpret_cast<const pol::bf16 *>(z_bf16), stats, (int)layer, ln_packed, partial, e8, (int)N);
    return launch_status("inv_policy_band_param_grads");
}

extern "C" int inv_policy_obs_to_x0_any(const void *obs, int32_t obs_is_bits, int32_t pixels, void *x0_bf16, int32_t N, void *stream)
{
    if (!obs || !x0_bf16 || N < 0 || pixels < 1) return inv_fail(INV_ERR_BAD_ARG, "inv_policy_obs_to_x0_any: NULL tensor, N < 0 or pixels < 1");
    if (N == 0) return INV_OK;
    const long long total = (long long)N * pixels;
    hipLaunchKernelGGL(pol::k_obs_to_x0_any, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obs, (int)(obs_is_bits != 0),
                       reinterpret_cast<pol::bf16 *>(x0_bf16), (int)pixels, (int)N);
    return launch_status("inv_policy_obs_to_x0_any");
}

extern "C" int inv_policy_trunk(const inv_policy_params *params, const float *obs, void *feat_bf16, int32_t N, void *stream)
{
    return launch_trunk(params, obs, false, feat_bf16, N, stream);
}

extern "C" int inv_policy_trunk_bits(const inv_policy_params *params, const uint32_t *bits, void *feat_bf16, int32_t N, void *stream)
{
    return launch_trunk(params, bits, true, feat_bf16, N, stream);
}

static bool saved_ok(const inv_policy_saved *sv)
{
    if (!sv || !sv->rstd) return false;
    for (int l = 0; l < 4; ++l)
        if (!sv->xhat[l] || (l < 3 && !sv->act[l])) return false;
    return true;
}

extern "C" int inv_policy_trunk_train_fwd(const inv_policy_params *params, const void *obs, int32_t obs_is_bits, void *feat_bf16,
                                          const inv_policy_saved *saved, int32_t N, void *stream)
{
    if (!params || !obs || !feat_bf16 || N < 0 || !layers_ok(params) || (saved && !saved_ok(saved)))
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_trunk_train_fwd: NULL params / obs / feat, a NULL layer or saved tensor, or N < 0");
    if (N == 0) return INV_OK;
    const int grid = trunk_grid(N);
    inv_policy_saved sv;
    memset(&sv, 0, sizeof sv);
    if (!saved) {
        // saved == NULL (rollouts that a fused update follows): the inference kernel with the training forward's numerics -- the same
        // features bit for bit, without the training epilogue's extra barrier and row pass
        if (obs_is_bits)
            hipLaunchKernelGGL((pol::k_policy_trunk<true, false, true>), dim3(grid), dim3(pol::THREADS), 0, (hipStream_t)stream, *params, obs,
                               reinterpret_cast<pol::bf16 *>(feat_bf16), (int)N, sv);
        else
            hipLaunchKernelGGL((pol::k_policy_trunk<false, false, true>), dim3(grid), dim3(pol::THREADS), 0, (hipStream_t)stream, *params, obs,
                               reinterpret_cast<pol::bf16 *>(feat_bf16), (int)N, sv);
        return launch_status("inv_policy_trunk_train_fwd");
    }
    sv = *saved;
    if (obs_is_bits)
        hipLaunchKernelGGL((pol::k_policy_trunk<true, true>), dim3(grid), dim3(pol::THREADS), 0, (hipStream_t)stream, *params, obs,
                           reinterpret_cast<pol::bf16 *>(feat_bf16), (int)N, sv);
    else
        hipLaunchKernelGGL((pol::k_policy_trunk<false, true>), dim3(grid), dim3(pol::THREADS), 0, (hipStream_t)stream, *params, obs,
                           reinterpret_cast<pol::bf16 *>(feat_bf16), (int)N, sv);
    return launch_status("inv_policy_trunk_train_fwd");
}

extern "C" int inv_policy_trunk_bwd(const inv_policy_bwd_params *params, const inv_policy_saved *saved, const void *dfeat_bf16,
                                    const inv_policy_grads *grads, int32_t N, void *stream)
{
    if (!params || !dfeat_bf16 || !grads || N < 0 || !saved_ok(saved))
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_trunk_bwd: NULL params / dfeat / grads, an incomplete saved set, or N < 0");
    for (int l = 0; l < 4; ++l)
        if (!params->ln[l] || (!grads->dy[l] && l < 3) || !grads->dz[l] || (l < 3 && !params->wt[l]))      // dy[3] may be NULL (ABI 8)
            return inv_fail(INV_ERR_BAD_ARG, "inv_policy_trunk_bwd: NULL tensor for layer %d", l);
    if (N == 0) return INV_OK;
    hipLaunchKernelGGL(pol::k_policy_trunk_bwd, dim3(trunk_grid(N)), dim3(pol::THREADS), 0, (hipStream_t)stream, *params, *saved,
                       reinterpret_cast<const pol::bf16 *>(dfeat_bf16), *grads, (int)N);
    return launch_status("inv_policy_trunk_bwd");
}

template <int CINP, int COUT, int MWW, int NBW, bool PF, class GEO = pol::WgGeo15>
static int launch_wgrad(const void *x, const void *dz, float *dw, float *db, int N, hipStream_t st)
{
    constexpr int MS = MWW * (pol::WG_WAVES / NBW), Q = (COUT / 16 + MS - 1) / MS;
    constexpr int DSTRIDE = MS * 16 + 8, XSTRIDE = CINP + 8;
    constexpr size_t lds = (size_t)(32 * GEO::KST * DSTRIDE + pol::NPOS * XSTRIDE) * 2;
    static bool attr_set[64] = { false };            // per device ordinal: the attribute belongs to the device's code object
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&pol::k_policy_wgrad<CINP, COUT, MWW, NBW, PF, GEO>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return inv_fail(INV_ERR_HIP, "inv_policy_wgrad: hipFuncSetAttribute(%d B of LDS): %s", (int)lds, hipGetErrorString(e));
        attr_set[dev] = true;
    }
    const int blocks = PF ? trunk_grid(N) / 2 : trunk_grid(N);          // PF: one 8-wave block per CU; else two (<= 126 VGPRs, <= 68 KB LDS)
    const int per_slice = blocks / Q > 0 ? blocks / Q : 1;
    const int nslices = N < per_slice ? N : per_slice;
    hipLaunchKernelGGL((pol::k_policy_wgrad<CINP, COUT, MWW, NBW, PF, GEO>), dim3(nslices * Q), dim3(pol::WG_THREADS), lds, st,
                       reinterpret_cast<const pol::bf16 *>(x), reinterpret_cast<const pol::bf16 *>(dz), dw, db, N);
    return launch_status("inv_policy_wgrad");
}

extern "C" int inv_policy_wgrad(int32_t layer, const void *x_bf16, const void *dz_bf16, float *dw, float *dbias, int32_t N, void *stream)
{
    if (!x_bf16 || !dz_bf16 || !dw || N < 0 || layer < 0 || layer > 3)
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_wgrad: NULL x / dz / dw, N < 0 or layer outside 0..3 (dbias may be NULL)");
    if (N == 0) return INV_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (layer) {
#ifndef POL_WG0
#define POL_WG0 16, 32, 1, 1, false
#endif
#ifndef POL_WG1
#define POL_WG1 32, 64, 1, 2, true                      // prefetching form: 0.171 vs 0.200 ms (layer 1: 0.205 vs 0.139, stays plain)
#endif
    case 0: return launch_wgrad<POL_WG0>(x_bf16, dz_bf16, dw, dbias, N, st);
    case 1: return launch_wgrad<POL_WG1>(x_bf16, dz_bf16, dw, dbias, N, st);
    case 2: return launch_wgrad<64, 128, 4, 4, true>(x_bf16, dz_bf16, dw, dbias, N, st);
    default: return launch_wgrad<128, 128, 4, 8, true>(x_bf16, dz_bf16, dw, dbias, N, st);
    }
}

// The four layers' weight gradients from the kernel's [Cout][9][CinP] order into the module's tensors [Cout][Cin][3][3] (assigned):
// one launch instead of four slice / permute / copy chains.  src: the four [Cout][9][CinP] blocks back to back.
extern "C" int inv_policy_wgrad_to_module(const float *src, const inv_policy_conv_grads *out, void *stream)
{
    if (!src || !out) return inv_fail(INV_ERR_BAD_ARG, "inv_policy_wgrad_to_module: NULL argument");
    pol::WgUnpack u;
    for (int l = 0; l < 4; ++l) {
        if (!out->dw[l]) return inv_fail(INV_ERR_BAD_ARG, "inv_policy_wgrad_to_module: NULL tensor for layer %d", l);
        u.dst[l] = out->dw[l];
    }
    u.src = src;
    hipLaunchKernelGGL(pol::k_wgrad_to_module, dim3((pol::WgUnpack::total() + 255) / 256), dim3(256), 0, (hipStream_t)stream, u);
    return launch_status("inv_policy_wgrad_to_module");
}

// weight / bias gradients of the banded trunk: N samples of the 32 x 32 board = 8 N band units of 128 dz pixels (WgGeoBand)
extern "C" int inv_policy_band_wgrad(int32_t layer, int32_t height, int32_t width, const void *x_bf16, const void *dz_bf16, float *dw, float *dbias,
                                     int32_t N, void *stream)
{
    if (!x_bf16 || !dz_bf16 || !dw || N < 0 || layer < 0 || layer > 3)
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_band_wgrad: NULL x / dz / dw, N < 0 or layer outside 0..3 (dbias may be NULL)");
    if (!band_dims_ok(height, width)) return inv_fail(INV_ERR_BAD_DIMS, "inv_policy_band_wgrad: built for the 32 x 32 board");
    if (N == 0) return INV_OK;
    if ((long long)N * 8 > 0x7fffffffLL) return inv_fail(INV_ERR_BAD_ARG, "inv_policy_band_wgrad: N too large");
    hipStream_t st = (hipStream_t)stream;
    const int units = N * 8;
    switch (layer) {
    case 0: return launch_wgrad<16, 32, 1, 1, false, pol::WgGeoBand>(x_bf16, dz_bf16, dw, dbias, units, st);
    case 1: return launch_wgrad<32, 64, 1, 2, true, pol::WgGeoBand>(x_bf16, dz_bf16, dw, dbias, units, st);
    case 2: return launch_wgrad<64, 128, 4, 4, true, pol::WgGeoBand>(x_bf16, dz_bf16, dw, dbias, units, st);
    default: return launch_wgrad<128, 128, 4, 8, true, pol::WgGeoBand>(x_bf16, dz_bf16, dw, dbias, units, st);
    }
}

extern "C" int inv_policy_param_grads_all2(const inv_policy_grads *grads, const inv_policy_saved *saved, const inv_policy_ln_grads *out,
                                           const void *dfeat_bf16, const void *ln_top_packed, int32_t slices, int32_t N, void *stream);

extern "C" int inv_policy_param_grads_all(const inv_policy_grads *grads, const inv_policy_saved *saved, const inv_policy_ln_grads *out,
                                          int32_t slices, int32_t N, void *stream)
{
    return inv_policy_param_grads_all2(grads, saved, out, nullptr, nullptr, slices, N, stream);
}

extern "C" int inv_policy_param_grads_all2(const inv_policy_grads *grads, const inv_policy_saved *saved, const inv_policy_ln_grads *out,
                                           const void *dfeat_bf16, const void *ln_top_packed, int32_t slices, int32_t N, void *stream)
{
    if (!grads || !saved || !out || N < 0 || slices < 1) return inv_fail(INV_ERR_BAD_ARG, "inv_policy_param_grads_all: NULL argument, N < 0 or slices < 1");
    if ((dfeat_bf16 == nullptr) != (ln_top_packed == nullptr))
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_param_grads_all2: dfeat and the top layer's packed LayerNorm parameters come together");
    pol::PgAll a;
    a.dfeat = reinterpret_cast<const pol::bf16 *>(dfeat_bf16);
    a.ln3 = reinterpret_cast<const _Float16 *>(ln_top_packed);
    for (int l = 0; l < 4; ++l) {
        if ((!grads->dy[l] && !(l == 3 && dfeat_bf16)) || !saved->xhat[l] || !out->gw[l] || !out->gb[l]) return inv_fail(INV_ERR_BAD_ARG, "inv_policy_param_grads_all: NULL tensor for layer %d", l);
        a.dy[l] = reinterpret_cast<const pol::bf16 *>(grads->dy[l]); a.xh[l] = reinterpret_cast<const pol::bf16 *>(saved->xhat[l]);
        a.gw[l] = out->gw[l]; a.gb[l] = out->gb[l];
    }
    if (N == 0) return INV_OK;
    const int items = pol::P * (4 + 8 + 16 + 16), sl = slices < N ? slices : N;
    hipLaunchKernelGGL(pol::k_policy_param_grads_all, dim3((items + 255) / 256, sl), dim3(256), 0, (hipStream_t)stream, a, (int)N);
    return launch_status("inv_policy_param_grads_all");
}

extern "C" int inv_policy_param_grads(const void *dy_bf16, const void *xhat_bf16, float *partial, int32_t slices,
                                      int32_t elems_per_sample, int32_t N, void *stream)
{
    if (!dy_bf16 || !xhat_bf16 || !partial || N < 0 || slices < 1 || elems_per_sample <= 0 || (elems_per_sample & 7))
        return inv_fail(INV_ERR_BAD_ARG, "inv_policy_param_grads: NULL tensor, N < 0, slices < 1 or elems_per_sample not a positive multiple of 8");
    const int e8 = elems_per_sample / 8, bx = (e8 + 255) / 256;
    hipLaunchKernelGGL(pol::k_policy_param_grads, dim3(bx, slices), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const pol::bf16 *>(dy_bf16),
                       reinterpret_cast<const pol::bf16 *>(xhat_bf16), partial, e8, (int)N);
    return launch_status("inv_policy_param_grads");
}
