// inv_heads.hip -- the heads of InversusCNNPolicy after their first Linear, fused:
//   reference: inversus_rl/policies.py:60-75 (fc_actor / fc_critic: Linear(19204, 256) ReLU Linear(256, 128) ReLU Linear(128, 13 | 1))
//              and :102-106 (features ++ extra -> logits, value), run under autograd in ppo_agent.py:206-230.
// The big GEMM  pre = features @ W1[:, :19200]^T  ([B x 19200] x [19200 x 512] bf16) and its two backward products are
// csrc/inv_fc1.hip.  Everything after it is tiny per sample (0.13 MFLOP) but was ~25 torch launches forward and ~60 backward over
// [B x 512] / [B x 128] tensors:
//   inv_heads_fwd   h = relu(pre + extra W1e^T + b1);  a2 = relu(W2 h + b2);  logits / value = W3 a2 + b3          (one kernel)
//   inv_heads_bwd   d logits, d value -> d a2 -> d pre (bf16, the operand of the two big backward GEMMs) and the gradients of
//                   b1, W1e, b2, W3, b3                                                                             (one kernel)
// One work unit = (tile of 32 samples, head); the two heads never mix.  Layers 2 and 3 are MFMA 16x16x32 bf16 with f32
// accumulation: A = weight fragments (kept in registers for the life of the block), B = the sample tile in LDS.  The same code
// serves rollouts and the update, so stored log-probs / values are reproduced exactly by the update's first forward.
// Numerics: h and a2 are rounded to bf16 (like the trunk's activations), W2 / W3 are bf16; bias adds, ReLU and the extra-feature
// columns are f32.  Not bit-comparable with the f32 module; tests bound the difference (tests/test_gpu_heads.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/inversus_hip.h"
#include "inv_internal.h"

namespace hd {

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256, S = 32;                       // samples per tile = two MFMA column tiles
constexpr int HID = INV_HEADS_HID, MID = INV_HEADS_MID, NACT = INV_HEADS_ACTIONS;
// row strides of the LDS tiles (bf16 elements): 528 B and 272 B = 4 banks past a multiple of 256 B, so the 16 rows that one
// ds_read_b128 wave quarter touches fall into 16 different 16-byte bank groups
constexpr int HS = HID + 8, AS = MID + 8;
constexpr int DLK = 32;                                    // d logits tile: k padded to one MFMA k-step

static_assert(HID == 256 && MID == 128 && NACT == 13, "tile ownership below is written for 256 -> 128 -> 13 | 1");

__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// ------------------------------------------------------------------------------------------------------------------
// forward
template <bool SAVE>
__global__ __launch_bounds__(THREADS, 2) void k_heads_fwd(inv_heads_params prm, const float *__restrict__ pre, int nslabs, const float *__restrict__ extra,
                                                          float *__restrict__ logits, float *__restrict__ value,
                                                          bf16 *__restrict__ h_save, bf16 *__restrict__ a2_save, int nheads, int B)
{
    __shared__ __attribute__((aligned(16))) bf16 tileH[S * HS];
    __shared__ __attribute__((aligned(16))) bf16 tileA[S * AS];
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, g = lane >> 4;
    const int wave = (int)__builtin_amdgcn_readfirstlane((uint32_t)(tid >> 6));
    const int head = (int)blockIdx.x % nheads, blk = (int)blockIdx.x / nheads, nblk = (int)gridDim.x / nheads;
    const int ntiles = (B + S - 1) / S;
    const int RL = nheads * HID;                             // row length of pre / h
    // this thread's 8 columns of the first layer: bias and extra-feature weights, loaded once
    const int part = tid & 31, row0 = tid >> 5;
    float b1v[8];
    float4 wev[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = head * HID + part * 8 + e;
        b1v[e] = prm.b1[k];
        wev[e] = reinterpret_cast<const float4 *>(prm.we)[k];
    }
    // weight fragments of layers 2 and 3, loaded once: wave w owns output rows 32w .. 32w + 31 of layer 2
    bf16x8 a2f[2][HID / 32];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ks = 0; ks < HID / 32; ++ks)
            a2f[mi][ks] = reinterpret_cast<const bf16x8 *>(prm.w2[head])[((2 * wave + mi) * (HID / 32) + ks) * 64 + lane];
    bf16x8 a3f[MID / 32];
#pragma unroll
    for (int ks = 0; ks < MID / 32; ++ks) a3f[ks] = reinterpret_cast<const bf16x8 *>(prm.w3[head])[ks * 64 + lane];
    float b2v[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int q = 0; q < 4; ++q) b2v[mi][q] = prm.b2[head][(2 * wave + mi) * 16 + 4 * g + q];
    const int nact = head == 0 ? NACT : 1;
    float b3v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b3v[q] = 4 * g + q < nact ? prm.b3[head][4 * g + q] : 0.0f;

    for (int t = blk; t < ntiles; t += nblk) {
        const int s0 = t * S;
        // layer 1 epilogue: pre (the f32 split-K slabs of inv_fc1_fwd, added in slab order) + b1 + extra W1e^T, ReLU -> h (bf16) in the tile
#pragma unroll
        for (int i = 0; i < S / 8; ++i) {
            const int r = row0 + 8 * i, s = min(s0 + r, B - 1);           // rows past B re-read the last sample; never stored
            const float *pp = pre + (size_t)s * RL + head * HID + part * 8;
            float4 p0 = reinterpret_cast<const float4 *>(pp)[0], p1 = reinterpret_cast<const float4 *>(pp)[1];
            for (int k = 1; k < nslabs; ++k) {
                const float4 q0 = reinterpret_cast<const float4 *>(pp + (size_t)k * B * RL)[0], q1 = reinterpret_cast<const float4 *>(pp + (size_t)k * B * RL)[1];
                p0.x += q0.x; p0.y += q0.y; p0.z += q0.z; p0.w += q0.w;
                p1.x += q1.x; p1.y += q1.y; p1.z += q1.z; p1.w += q1.w;
            }
            const float pv[8] = { p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w };
            const float4 ex = reinterpret_cast<const float4 *>(extra)[s];
            bf16x8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = pv[e] + b1v[e];
                v = fmaf(ex.x, wev[e].x, v); v = fmaf(ex.y, wev[e].y, v); v = fmaf(ex.z, wev[e].z, v); v = fmaf(ex.w, wev[e].w, v);
                hv[e] = (bf16)fmaxf(v, 0.0f);
            }
            *reinterpret_cast<bf16x8 *>(tileH + r * HS + part * 8) = hv;
            if (SAVE && s0 + r < B) *reinterpret_cast<bf16x8 *>(h_save + (size_t)(s0 + r) * RL + head * HID + part * 8) = hv;
        }
        __syncthreads();
        // layer 2: D[j][s] = sum_k W2[j][k] h[s][k], rows 32 wave .. + 31, both column tiles
        f32x4 acc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[mi][ni][q] = b2v[mi][q];
#pragma unroll
        for (int ks = 0; ks < HID / 32; ++ks) {
            const bf16x8 bq0 = *reinterpret_cast<const bf16x8 *>(tileH + c16 * HS + ks * 32 + 8 * g);
            const bf16x8 bq1 = *reinterpret_cast<const bf16x8 *>(tileH + (16 + c16) * HS + ks * 32 + 8 * g);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                acc[mi][0] = mfma(a2f[mi][ks], bq0, acc[mi][0]);
                acc[mi][1] = mfma(a2f[mi][ks], bq1, acc[mi][1]);
            }
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                bf16x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (bf16)fmaxf(acc[mi][ni][q], 0.0f);
                *reinterpret_cast<bf16x4 *>(tileA + (ni * 16 + c16) * AS + (2 * wave + mi) * 16 + 4 * g) = o;
            }
        __syncthreads();
        // layer 3: 16 output rows (13 | 1 used), one column tile per wave 0 / 1
        if (wave < 2) {
            f32x4 o3;
#pragma unroll
            for (int q = 0; q < 4; ++q) o3[q] = b3v[q];
#pragma unroll
            for (int ks = 0; ks < MID / 32; ++ks)
                o3 = mfma(a3f[ks], *reinterpret_cast<const bf16x8 *>(tileA + (wave * 16 + c16) * AS + ks * 32 + 8 * g), o3);
            const int s = s0 + wave * 16 + c16;
            if (s < B) {
                if (head == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (4 * g + q < NACT) logits[(size_t)s * NACT + 4 * g + q] = o3[q];
                } else if (g == 0) {
                    value[s] = o3[0];
                }
            }
        }
        if (SAVE) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + i * THREADS, r = c >> 4, pc = c & 15;
                if (s0 + r < B)
                    *reinterpret_cast<bf16x8 *>(a2_save + (size_t)(s0 + r) * (nheads * MID) + head * MID + pc * 8) =
                        *reinterpret_cast<const bf16x8 *>(tileA + r * AS + pc * 8);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward.  Per unit: d a2 = (W3^T d out) * [a2 > 0];  d pre = (W2^T d a2) * [h > 0];  both go to HBM as bf16 (d a2 and h are the
// operands of the W2 gradient GEMM, d pre of the two big GEMMs of the first Linear).  The small gradients are accumulated per
// thread over the block's tiles and added to the f32 buffers once per block:
//   d b1[k] = sum_s d pre[s][k]     d W1e[k][e] = sum_s d pre[s][k] extra[s][e]     d b2[j] = sum_s d a2[s][j]
//   d W3[a][j] = sum_s d out[s][a] a2[s][j]          d b3[a] = sum_s d out[s][a]
__global__ __launch_bounds__(THREADS, 2) void k_heads_bwd(inv_heads_params prm, const float *__restrict__ dlogits, const float *__restrict__ dvalue,
                                                          const bf16 *__restrict__ h, const bf16 *__restrict__ a2, const float *__restrict__ extra,
                                                          bf16 *__restrict__ dpre, bf16 *__restrict__ dpre_t, int ldt, bf16 *__restrict__ da2,
                                                          inv_heads_grads gr, int B)
{
    __shared__ __attribute__((aligned(16))) bf16 tileH[S * HS];     // h, then d pre in place
    __shared__ __attribute__((aligned(16))) bf16 tileA[S * AS];     // a2
    __shared__ __attribute__((aligned(16))) bf16 tileD[S * AS];     // d a2
    __shared__ __attribute__((aligned(16))) bf16 tileL[S * DLK];    // d out as the MFMA operand (bf16, k padded with zeros)
    __shared__ float tileLf[S * 16];                                // d out, f32 (weight / bias gradients)
    __shared__ float tileE[S * 4];                                  // extra features
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, g = lane >> 4;
    const int wave = (int)__builtin_amdgcn_readfirstlane((uint32_t)(tid >> 6));
    const int head = (int)blockIdx.x & 1, blk = (int)blockIdx.x >> 1, nblk = (int)gridDim.x >> 1;
    const int ntiles = (B + S - 1) / S;
    const int nact = head == 0 ? NACT : 1;
    constexpr int RL = 2 * HID;

    bf16x8 a3t[2];                                                   // W3^T fragments: rows 32 wave .. + 31 of [128][32]
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) a3t[mi] = reinterpret_cast<const bf16x8 *>(prm.w3t[head])[(2 * wave + mi) * 64 + lane];
    bf16x8 a2t[4][MID / 32];                                         // W2^T fragments: rows 64 wave .. + 63 of [256][128]
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ks = 0; ks < MID / 32; ++ks)
            a2t[mi][ks] = reinterpret_cast<const bf16x8 *>(prm.w2t[head])[((4 * wave + mi) * (MID / 32) + ks) * 64 + lane];

    // per-thread gradient accumulators over the block's tiles
    float g_b1 = 0.0f, g_we[4] = { 0.0f, 0.0f, 0.0f, 0.0f };          // column k = tid of this head
    float g_b2 = 0.0f;                                                // j = tid & 127 (threads 0 .. 127)
    float g_w3[7] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };      // j = tid & 127, actions 7 (tid >> 7) .. + 6
    float g_b3 = 0.0f;                                                // action tid (threads 0 .. nact - 1)
    const int j3 = tid & 127, a3lo = 7 * (tid >> 7);

    for (int t = blk; t < ntiles; t += nblk) {
        const int s0 = t * S;
        // tiles: h and a2 rows (rows past B: zeros, so that they contribute nothing), d out, extra
#pragma unroll
        for (int i = 0; i < S / 8; ++i) {
            const int r = (tid >> 5) + 8 * i, pc = tid & 31;
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)0.0f;
            if (s0 + r < B) v = *reinterpret_cast<const bf16x8 *>(h + (size_t)(s0 + r) * RL + head * HID + pc * 8);
            *reinterpret_cast<bf16x8 *>(tileH + r * HS + pc * 8) = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * THREADS, r = c >> 4, pc = c & 15;
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)0.0f;
            if (s0 + r < B) v = *reinterpret_cast<const bf16x8 *>(a2 + (size_t)(s0 + r) * (2 * MID) + head * MID + pc * 8);
            *reinterpret_cast<bf16x8 *>(tileA + r * AS + pc * 8) = v;
        }
#pragma unroll
        for (int i = 0; i < S * DLK / THREADS; ++i) {
            const int c = tid + i * THREADS, r = c / DLK, a = c % DLK;
            float v = 0.0f;
            if (s0 + r < B && a < nact) v = head == 0 ? dlogits[(size_t)(s0 + r) * NACT + a] : dvalue[s0 + r];
            tileL[r * DLK + a] = (bf16)v;
            if (a < 16) tileLf[r * 16 + a] = v;
        }
        if (tid < S * 4) tileE[tid] = s0 + (tid >> 2) < B ? extra[(size_t)s0 * 4 + tid] : 0.0f;
        __syncthreads();
        // d a2: D[j][s] = sum_a W3[a][j] d out[s][a]   (K = one k-step), masked by a2 > 0
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const bf16x8 bq = *reinterpret_cast<const bf16x8 *>(tileL + (ni * 16 + c16) * DLK + 8 * g);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                f32x4 d = { 0.0f, 0.0f, 0.0f, 0.0f };
                d = mfma(a3t[mi], bq, d);
                const int off = (ni * 16 + c16) * AS + (2 * wave + mi) * 16 + 4 * g;
                const bf16x4 av = *reinterpret_cast<const bf16x4 *>(tileA + off);
                bf16x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (bf16)((float)av[q] > 0.0f ? d[q] : 0.0f);
                *reinterpret_cast<bf16x4 *>(tileD + off) = o;
            }
        }
        __syncthreads();
        // d pre: D[k][s] = sum_j W2[j][k] d a2[s][j], rows 64 wave .. + 63, masked by h > 0, written over h
        {
            f32x4 acc[4][2];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[mi][ni][q] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < MID / 32; ++ks) {
                const bf16x8 bq0 = *reinterpret_cast<const bf16x8 *>(tileD + c16 * AS + ks * 32 + 8 * g);
                const bf16x8 bq1 = *reinterpret_cast<const bf16x8 *>(tileD + (16 + c16) * AS + ks * 32 + 8 * g);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    acc[mi][0] = mfma(a2t[mi][ks], bq0, acc[mi][0]);
                    acc[mi][1] = mfma(a2t[mi][ks], bq1, acc[mi][1]);
                }
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    bf16 *hp = tileH + (ni * 16 + c16) * HS + (4 * wave + mi) * 16 + 4 * g;      // this lane's own four entries
                    const bf16x4 hv = *reinterpret_cast<const bf16x4 *>(hp);
                    bf16x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = (bf16)((float)hv[q] > 0.0f ? acc[mi][ni][q] : 0.0f);
                    *reinterpret_cast<bf16x4 *>(hp) = o;
                }
        }
        // (tileD / tileA / tileLf are only read in this phase) d a2 to HBM, and the gradients that need a2 / d a2 / d out
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * THREADS, r = c >> 4, pc = c & 15;
            if (s0 + r < B)
                *reinterpret_cast<bf16x8 *>(da2 + (size_t)(s0 + r) * (2 * MID) + head * MID + pc * 8) = *reinterpret_cast<const bf16x8 *>(tileD + r * AS + pc * 8);
        }
        if (tid < MID) {
            float sb = 0.0f;
#pragma unroll 8
            for (int r = 0; r < S; ++r) sb += (float)tileD[r * AS + tid];
            g_b2 += sb;
        }
        if (head == 0 || tid < MID) {
#pragma unroll 4
            for (int r = 0; r < S; ++r) {
                const float av = (float)tileA[r * AS + j3];
#pragma unroll
                for (int a = 0; a < 7; ++a)
                    if (a3lo + a < nact) g_w3[a] = fmaf(tileLf[r * 16 + a3lo + a], av, g_w3[a]);
            }
        }
        if (tid < nact) {
            float sb = 0.0f;
            for (int r = 0; r < S; ++r) sb += tileLf[r * 16 + tid];
            g_b3 += sb;
        }
        __syncthreads();
        // d pre to HBM; d b1 and d W1e from the ROUNDED d pre (what the big GEMMs consume)
#pragma unroll
        for (int i = 0; i < S / 8; ++i) {
            const int r = (tid >> 5) + 8 * i, pc = tid & 31;
            if (s0 + r < B)
                *reinterpret_cast<bf16x8 *>(dpre + (size_t)(s0 + r) * RL + head * HID + pc * 8) = *reinterpret_cast<const bf16x8 *>(tileH + r * HS + pc * 8);
        }
        {
            // column `tid` of the tile: its sums, and (dpre_t) the same 32 values as one 64-byte piece of row k of d pre^T -- the
            // K-contiguous operand of the weight-gradient product  dW1 = d pre^T @ features  (inv_fc1_dw)
            float sb = 0.0f, se[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll 8
            for (int r = 0; r < S; ++r) {
                const float d = (float)tileH[r * HS + tid];
                sb += d;
#pragma unroll
                for (int e = 0; e < 4; ++e) se[e] = fmaf(d, tileE[r * 4 + e], se[e]);
            }
            g_b1 += sb;
#pragma unroll
            for (int e = 0; e < 4; ++e) g_we[e] += se[e];
            if (dpre_t) {
                bf16 *dst = dpre_t + (size_t)(head * HID + tid) * ldt + s0;
                if (s0 + S <= B) {
                    for (int i = 0; i < S / 8; ++i) {                            // not unrolled: 4 registers live, not 16
                        bf16x8 col;
#pragma unroll
                        for (int e = 0; e < 8; ++e) col[e] = tileH[(8 * i + e) * HS + tid];
                        reinterpret_cast<bf16x8 *>(dst)[i] = col;
                    }
                } else {
                    for (int r = 0; r < B - s0; ++r) dst[r] = tileH[r * HS + tid];      // ragged last tile (no dynamic register indexing)
                }
            }
        }
        __syncthreads();
    }
    atomicAdd(gr.db1 + head * HID + tid, g_b1);
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(gr.dwe + (size_t)(head * HID + tid) * 4 + e, g_we[e]);
    if (tid < MID) atomicAdd(gr.db2[head] + tid, g_b2);
    if (head == 0 || tid < MID) {
#pragma unroll
        for (int a = 0; a < 7; ++a)
            if (a3lo + a < nact) atomicAdd(gr.dw3[head] + (size_t)(a3lo + a) * MID + j3, g_w3[a]);
    }
    if (tid < nact) atomicAdd(gr.db3[head] + tid, g_b3);
}


// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of the heads' second Linear:  dW2[h][j][k] = sum_s d a2[s][128 h + j] * hs[s][256 h + k]   (j < 128, k < 256).
// The contraction runs over SAMPLES, the row index of both saved tensors, so both MFMA operands come from the transposing LDS
// read (rows = samples).  Block = (head, chunk of 256 samples), 4 waves; wave w owns output rows 32 w .. 32 w + 31 x all 256 columns
// (32 accumulator tiles) over the chunk's eight 32-sample steps and ADDS them with f32 atomics (64 contiguous bytes per 16-lane group).
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_ptr;
__device__ __forceinline__ bf16x8 tr_frag(const bf16 *row_lo, const bf16 *row_hi)
{
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(row_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(row_hi));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = { lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3] };
    return __builtin_bit_cast(bf16x8, v);
}

constexpr int DW2_CHUNK = 256;

__global__ __launch_bounds__(THREADS) void k_heads_dw2(const bf16 *__restrict__ da2, const bf16 *__restrict__ hs, float *__restrict__ dw2a,
                                                       float *__restrict__ dw2c, int B)
{
    __shared__ __attribute__((aligned(16))) bf16 tileD[S * AS];      // d a2 [32 samples][128]
    __shared__ __attribute__((aligned(16))) bf16 tileH[S * HS];      // h    [32 samples][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)__builtin_amdgcn_readfirstlane((uint32_t)(tid >> 6));
    const int head = (int)blockIdx.x & 1, chunk = (int)blockIdx.x >> 1;
    const int g = lane >> 4, q = (lane >> 2) & 3, p4 = (lane & 3) * 4;
    f32x4 acc[2][16];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 16; ++ni) acc[mi][ni] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    const int send = min(B, (chunk + 1) * DW2_CHUNK);
    for (int s0 = chunk * DW2_CHUNK; s0 < send; s0 += S) {
        __syncthreads();                                             // the previous step's reads are done
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * THREADS, r = c >> 4, pc = c & 15;
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)0.0f;
            if (s0 + r < B) v = *reinterpret_cast<const bf16x8 *>(da2 + (size_t)(s0 + r) * (2 * MID) + head * MID + pc * 8);
            *reinterpret_cast<bf16x8 *>(tileD + r * AS + pc * 8) = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * THREADS, r = c >> 5, pc = c & 31;
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)0.0f;
            if (s0 + r < B) v = *reinterpret_cast<const bf16x8 *>(hs + (size_t)(s0 + r) * (2 * HID) + head * HID + pc * 8);
            *reinterpret_cast<bf16x8 *>(tileH + r * HS + pc * 8) = v;
        }
        __syncthreads();
        bf16x8 af[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
            af[mi] = tr_frag(tileD + (8 * g + q) * AS + (2 * wave + mi) * 16 + p4, tileD + (8 * g + 4 + q) * AS + (2 * wave + mi) * 16 + p4);
#pragma unroll
        for (int ni = 0; ni < 16; ++ni) {
            const bf16x8 bfr = tr_frag(tileH + (8 * g + q) * HS + ni * 16 + p4, tileH + (8 * g + 4 + q) * HS + ni * 16 + p4);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = mfma(af[mi], bfr, acc[mi][ni]);
        }
    }
    float *dst = head == 0 ? dw2a : dw2c;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 16; ++ni)
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(dst + (size_t)((2 * wave + mi) * 16 + 4 * g + e) * HID + ni * 16 + (lane & 15), acc[mi][ni][e]);
}

// ------------------------------------------------------------------------------------------------------------------
// packing.  MFMA A fragments of a row-major f32 matrix (or of its transpose), rows / k padded with zeros:
//   out[(mt * (kpad / 32) + ks) * 64 + lane][e] = M[mt * 16 + (lane & 15)][ks * 32 + (lane >> 4) * 8 + e]
__global__ __launch_bounds__(256) void k_pack_frag(inv_heads_pack_list list)
{
    const inv_heads_pack_item it = list.item[blockIdx.y];
    const int total = it.mpad * it.kpad;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int e = i & 7, ln = (i >> 3) & 63, blkk = i >> 9, ksn = it.kpad / 32, mt = blkk / ksn, ks = blkk - mt * ksn;
        const int row = mt * 16 + (ln & 15), k = ks * 32 + (ln >> 4) * 8 + e;
        float v = 0.0f;
        if (row < it.rows && k < it.cols) v = it.transposed ? it.src[(size_t)k * it.ld + row] : it.src[(size_t)row * it.ld + k];
        reinterpret_cast<bf16 *>(it.dst)[i] = (bf16)v;
    }
}

// First Linear of the heads: rows of `w` ([rows][128 P + 4] f32, feature index c * P + p as nn.Flatten leaves it) -> bf16
// [rows][128 P] in the trunk kernel's channels-last order p * 128 + c, and the 4 extra-feature columns as f32 [rows][4].
// Per row a [128][P] -> [P][128] transpose: tiles of 128 channels x 32 pixels through LDS, 128-byte reads and 256-byte writes.
__global__ __launch_bounds__(256) void k_pack_fc1(const float *__restrict__ w, int rows, int P, bf16 *__restrict__ wf, float *__restrict__ we)
{
    __shared__ float tl[128 * 33];
    const int r = blockIdx.y, nf = 128 * P, ld = nf + 4, p0 = blockIdx.x * 32;
    const float *src = w + (size_t)r * ld;
    for (int i = threadIdx.x; i < 128 * 32; i += 256) {
        const int c = i >> 5, dp = i & 31;
        tl[c * 33 + dp] = p0 + dp < P ? src[c * P + p0 + dp] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 128; i += 256) {
        const int dp = i >> 7, c = i & 127;
        if (p0 + dp < P) wf[(size_t)r * nf + (size_t)(p0 + dp) * 128 + c] = (bf16)tl[c * 33 + dp];
    }
    if (blockIdx.x == 0 && threadIdx.x < 4) we[r * 4 + threadIdx.x] = src[nf + threadIdx.x];
    (void)rows;
}

// bf16 [R][C] -> [C][R] (the K-contiguous copy of the first Linear's weights for  d features = d pre @ W: 0.40 vs 0.45 ms as NT)
__global__ __launch_bounds__(256) void k_transpose_bf16(const bf16 *__restrict__ in, int R, int C, bf16 *__restrict__ out)
{
    __shared__ bf16 tl[64 * 66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int dr = i >> 6, dc = i & 63;
        tl[dr * 66 + dc] = (r0 + dr < R && c0 + dc < C) ? in[(size_t)(r0 + dr) * C + c0 + dc] : (bf16)0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int dc = i >> 6, dr = i & 63;
        if (r0 + dr < R && c0 + dc < C) out[(size_t)(c0 + dc) * R + r0 + dr] = tl[dr * 66 + dc];
    }
}

} // namespace hd

static int heads_status(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return inv_fail(INV_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return INV_OK;
}

static bool heads_params_ok(const inv_heads_params *p, int nheads, bool need_t)
{
    if (!p || !p->we || !p->b1) return false;
    for (int h = 0; h < nheads; ++h) {
        if (!p->w2[h] || !p->w3[h] || !p->b2[h] || !p->b3[h]) return false;
        if (need_t && (!p->w2t[h] || !p->w3t[h])) return false;
    }
    return true;
}

extern "C" int inv_heads_pack(const inv_heads_pack_list *list, int32_t count, void *stream)
{
    if (!list || count < 0 || count > INV_HEADS_PACK_MAX) return inv_fail(INV_ERR_BAD_ARG, "inv_heads_pack: NULL list or count outside 0..%d", INV_HEADS_PACK_MAX);
    for (int i = 0; i < count; ++i) {
        const inv_heads_pack_item &it = list->item[i];
        if (!it.src || !it.dst || it.rows <= 0 || it.cols <= 0 || it.ld <= 0 || it.mpad < it.rows || it.kpad < it.cols || it.mpad % 16 || it.kpad % 32)
            return inv_fail(INV_ERR_BAD_ARG, "inv_heads_pack: item %d: NULL pointer, or mpad / kpad not the padded (16 / 32) sizes of rows / cols", i);
    }
    if (count == 0) return INV_OK;
    hipLaunchKernelGGL(hd::k_pack_frag, dim3(16, count), dim3(256), 0, (hipStream_t)stream, *list);
    return heads_status("inv_heads_pack");
}

extern "C" int inv_heads_pack_fc1(const float *w, int32_t rows, int32_t pixels, void *wf_bf16, float *we, void *stream)
{
    if (!w || !wf_bf16 || !we || rows <= 0 || pixels <= 0) return inv_fail(INV_ERR_BAD_ARG, "inv_heads_pack_fc1: NULL pointer or rows / pixels <= 0");
    hipLaunchKernelGGL(hd::k_pack_fc1, dim3((pixels + 31) / 32, rows), dim3(256), 0, (hipStream_t)stream, w, rows, pixels,
                       reinterpret_cast<hd::bf16 *>(wf_bf16), we);
    return heads_status("inv_heads_pack_fc1");
}

extern "C" int inv_transpose_bf16(const void *in_bf16, int32_t rows, int32_t cols, void *out_bf16, void *stream)
{
    if (!in_bf16 || !out_bf16 || rows <= 0 || cols <= 0) return inv_fail(INV_ERR_BAD_ARG, "inv_transpose_bf16: NULL pointer or rows / cols <= 0");
    hipLaunchKernelGGL(hd::k_transpose_bf16, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const hd::bf16 *>(in_bf16), (int)rows, (int)cols, reinterpret_cast<hd::bf16 *>(out_bf16));
    return heads_status("inv_transpose_bf16");
}

static int heads_grid(int B, int nheads)
{
    const int tiles = (B + hd::S - 1) / hd::S;
    const int per_head = tiles < 2 * inv_cu_count() ? tiles : 2 * inv_cu_count();     // two blocks per CU and head at most: the weights stay in registers
    return per_head * nheads;
}

extern "C" int inv_heads_fwd(const inv_heads_params *params, const float *pre_slabs, int32_t nslabs, const float *extra, float *logits, float *value,
                             void *h_save, void *a2_save, int32_t nheads, int32_t B, void *stream)
{
    if ((nheads != 1 && nheads != 2) || B < 0 || nslabs < 1 || !heads_params_ok(params, nheads, false) || !pre_slabs || !extra || !logits || (nheads == 2 && !value) ||
        ((h_save != nullptr) != (a2_save != nullptr)))
        return inv_fail(INV_ERR_BAD_ARG, "inv_heads_fwd: nheads not 1 | 2, B < 0, nslabs < 1, a NULL parameter / input / output, or only one of h_save / a2_save");
    if (B == 0) return INV_OK;
    const dim3 grid(heads_grid(B, nheads));
    if (h_save)
        hipLaunchKernelGGL(hd::k_heads_fwd<true>, grid, dim3(hd::THREADS), 0, (hipStream_t)stream, *params, pre_slabs, (int)nslabs, extra,
                           logits, value, reinterpret_cast<hd::bf16 *>(h_save), reinterpret_cast<hd::bf16 *>(a2_save), (int)nheads, (int)B);
    else
        hipLaunchKernelGGL(hd::k_heads_fwd<false>, grid, dim3(hd::THREADS), 0, (hipStream_t)stream, *params, pre_slabs, (int)nslabs, extra,
                           logits, value, nullptr, nullptr, (int)nheads, (int)B);
    return heads_status("inv_heads_fwd");
}

extern "C" int inv_heads_bwd(const inv_heads_params *params, const float *dlogits, const float *dvalue, const void *h, const void *a2, const float *extra,
                             void *dpre_bf16, void *dpre_t_bf16, int32_t dpre_t_ld, void *da2_bf16, const inv_heads_grads *grads, int32_t B, void *stream)
{
    if (dpre_t_bf16 && (dpre_t_ld < B || dpre_t_ld % 8 != 0))
        return inv_fail(INV_ERR_BAD_ARG, "inv_heads_bwd: dpre_t needs a leading dimension >= B that is a multiple of 8 (16-byte rows), got %d for B = %d", (int)dpre_t_ld, (int)B);
    if (B < 0 || !heads_params_ok(params, 2, true) || !dlogits || !dvalue || !h || !a2 || !extra || !dpre_bf16 || !da2_bf16 || !grads || !grads->db1 || !grads->dwe ||
        !grads->db2[0] || !grads->db2[1] || !grads->dw3[0] || !grads->dw3[1] || !grads->db3[0] || !grads->db3[1])
        return inv_fail(INV_ERR_BAD_ARG, "inv_heads_bwd: B < 0 or a NULL parameter / input / output / gradient buffer");
    if (B == 0) return INV_OK;
    hipLaunchKernelGGL(hd::k_heads_bwd, dim3(heads_grid(B, 2)), dim3(hd::THREADS), 0, (hipStream_t)stream, *params, dlogits, dvalue,
                       reinterpret_cast<const hd::bf16 *>(h), reinterpret_cast<const hd::bf16 *>(a2), extra, reinterpret_cast<hd::bf16 *>(dpre_bf16),
                       reinterpret_cast<hd::bf16 *>(dpre_t_bf16), (int)dpre_t_ld, reinterpret_cast<hd::bf16 *>(da2_bf16), *grads, (int)B);
    return heads_status("inv_heads_bwd");
}

extern "C" int inv_heads_dw2(const void *da2_bf16, const void *h_bf16, int32_t B, float *dw2_actor, float *dw2_critic, void *stream)
{
    if (!da2_bf16 || !h_bf16 || !dw2_actor || !dw2_critic || B < 0) return inv_fail(INV_ERR_BAD_ARG, "inv_heads_dw2: NULL pointer or B < 0");
    if (B == 0) return INV_OK;
    hipLaunchKernelGGL(hd::k_heads_dw2, dim3(2 * ((B + hd::DW2_CHUNK - 1) / hd::DW2_CHUNK)), dim3(hd::THREADS), 0, (hipStream_t)stream,
                       reinterpret_cast<const hd::bf16 *>(da2_bf16), reinterpret_cast<const hd::bf16 *>(h_bf16), dw2_actor, dw2_critic, (int)B);
    return heads_status("inv_heads_dw2");
}
