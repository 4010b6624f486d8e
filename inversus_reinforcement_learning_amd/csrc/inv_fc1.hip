// inv_fc1.hip -- the first Linear of the two policy heads as kernels of this library ("fused heads GEMM
// [B x 19 204] . [19 204 x 512]").  Reference: inversus_rl/policies.py:60-75 (fc_actor[0] / fc_critic[0] = Linear(128 P + 4, 256))
// and :102-106 (features ++ extra -> heads), differentiated by autograd in ppo_agent.py:211-232.  Three products, all bf16
// operands with f32 accumulation on v_mfma_f32_16x16x32_bf16:
//   inv_fc1_fwd     pre[b][j]   = sum_k feat[b][k] wf[j][k]          K = 128 P (19 200 | 131 072), j < 256 nheads      (rollouts + update)
//   inv_fc1_dfeat   dfeat[b][k] = sum_j dpre[b][j] wft[k][j]         K = 512: what the trunk's backward waits for
//   inv_fc1_dw      dwf[j][k]   = sum_b dpre_t[j][b] feat[b][k]      K = B: the weight gradient, straight into the module's layout
// (the 4 extra-feature columns, the bias and the ReLU of the layer are inv_heads_fwd's first-layer epilogue).
//
// One kernel core for all three: a 256 x 256 output tile per 512-thread workgroup (8 waves = 2 x 4, a wave owns 128 x 64 = 8 x 4 MFMA
// tiles = 128 accumulator registers), K-tiles of 64 streamed HBM / L2 -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane, no
// VGPR round trip) in 16-KB half-tiles (X rows 0-127 / 128-255, W rows 0-127 / 128-255; five 32-KiB buffers: the operand that streams
// from HBM gets three -- X in the forward, W in the weight gradient --, the other two; dfeat: 2 + 2 and an epilogue window -- = all
// 160 KiB of the CU).  A K-tile is two PHASES of [LDS fragment reads + four DMA requests]
// s_barrier [32 MFMAs] s_barrier, and waves 4-7 run ONE BARRIER BEHIND waves 0-3, so on every SIMD one wave computes while its
// partner loads (the two waves of a SIMD are w and w + 4).  DMA requests stay in flight across the barriers (raw s_barrier, counted
// vmcnt: every wait leaves exactly the youngest group of four requests outstanding).  The hazard table sits at the main loop.
// LDS image of a half-tile [128 rows][64 k]: 1-KB pieces of 8 rows x 128 B (one DMA wave-instruction = one piece = 8 full
// 128-byte lines of the operand), the 16-byte chunk index XORed with (row >> 1) & 7 on the SOURCE address (the DMA writes LDS
// lane-linearly) and on the fragment reads: every ds_read_b128 quarter-wave touches 16 different bank groups.
// The weight gradient's `feat` operand is K-STRIDED (its rows are the contraction index): it is staged as pieces of 8 k-rows x
// 64 columns and read with ds_read_b64_tr_b16 (the transposing LDS read), 32-byte column blocks XORed so that the eight rows of a
// half-wave fall into eight different bank groups.
// Work distribution (one 512-thread workgroup per CU: the pipeline buffers are 128 KiB of LDS):
//   forward  split-K so that 256 workgroups exist (B = 16 384: 64 x 2 tiles x 2 K-halves); the two column tiles of a row slab and
//            one K-half per XCD, so `feat` leaves HBM once (FETCH_SIZE 354 MB x 2 = 1.1 x its 629 MB).  The partial sums go to f32
//            slabs that inv_heads_fwd adds in slab order -- one rounding less than a bf16 `pre`.
//   dfeat    persistent workgroups walk the 4 800 tiles, the DMA stream runs across tile boundaries; every XCD walks its own
//            eighth of the tile sequence (its ~10 column tiles of W stay in its L2).  The bf16 tile leaves through a 4-KB LDS window
//            per wave as whole 128-byte rows, written through (measured: 8-byte-per-lane stores straight from the accumulator
//            layout cost as much as the product itself -- 16 partial lines per store instruction).
//   dw       stream-K over (column tile, K-tile) units of `feat`, 150 per workgroup pair; the two row tiles of a unit run side by
//            side on one XCD.  f32 segment slabs, summed and transposed into the module's [j][c P + p] order by k_fc1_dw_reduce.
// For a fixed split the result for a sample does not depend on the batch or on its position in it (fixed K order per output
// element, no atomics: bit-identical between launches).  Measured, same-process A/B against the GEMM library on random data
// (B = 16 384, tools/fc1_bench.py, profiles/r04_fc1_experiments.md): forward 0.25 vs 0.31 ms (1.29 PFLOP/s), dfeat 0.31-0.34 vs 0.37,
// dw + its reduce into the module layout 0.33 vs 0.36 + 0.04 (unpack); what holds them there is written down next to the numbers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/inversus_hip.h"
#include "inv_internal.h"

#if defined(FC1_KNOCK) && !defined(INV_TOOLS_BUILD)
#error "FC1_KNOCK builds compute wrong results on purpose: tools/build_variant.py only (it defines INV_TOOLS_BUILD)"
#endif
#ifndef FC1_KNOCK
#define FC1_KNOCK 0                       // tools-only timing probes (wrong results on purpose): 1 no fragment reads, 2 no DMA requests, 4 requests through zero-record descriptors (issued, dropped by the range check)
#endif

namespace fc1 {

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;

constexpr int THREADS = 512;
constexpr int BM = 256, BN = 256, BK = 64;
[[maybe_unused]] constexpr int HALF = 16384;              // bytes of a half-tile: 128 rows x 64 k (or 64 k-rows x 128 columns) of bf16
constexpr int LDS_BYTES = 163840;         // all 160 KiB of the CU (the map is above the kernel)

enum { MODE_FWD = 0, MODE_DFEAT = 1, MODE_DW = 2 };

struct Args {
    const bf16 *X, *W;                   // X: the operand whose rows land on MFMA lanes (output rows), W: rows on registers (output columns)
    void *out;
    int M, N, K;                         // valid rows of X, rows (MODE_DW: columns) of W, contraction length
    int ldx, ldw, ldo;                   // leading dimensions in elements
    int tiles_m, tiles_n, split, ktiles; // ktiles = K / 64
    int total;                           // work items (FWD: tiles x split; DFEAT: tiles) or K-tiles per workgroup (DW)
};

struct Item { int m0, n0, kt0, kt1, aux; };   // K-tile range [kt0, kt1); aux: slab (FWD) / segment (DW)

__device__ __forceinline__ int xcd_remap(int b, int total)
{
    // blocks b and b + 8 share an XCD (observed dispatch; speed only): give each XCD a contiguous range of work ids
    return (total & 7) == 0 ? (b & 7) * (total >> 3) + (b >> 3) : b;
}

template <int MODE>
__device__ __forceinline__ Item get_item(const Args &a, int seq)
{
    Item it;
    it.m0 = it.n0 = it.kt0 = it.kt1 = it.aux = 0;
    if (MODE == MODE_FWD) {
        if (seq == 0) {
            // n fastest (the two column tiles of an X slab run side by side on one XCD), K-split slowest (an XCD streams one K range of W)
            const int L = xcd_remap((int)blockIdx.x, a.total);
            const int n = L % a.tiles_n, r = L / a.tiles_n, m = r % a.tiles_m, s = r / a.tiles_m;
            it.m0 = m * BM; it.n0 = n * BN; it.aux = s;
            it.kt0 = (int)((long long)a.ktiles * s / a.split);
            it.kt1 = (int)((long long)a.ktiles * (s + 1) / a.split);
        }
    } else if (MODE == MODE_DFEAT) {
        // Persistent workgroups.  Each XCD (the blocks b with equal b & 7: observed dispatch, speed only) walks ONE eighth of the
        // tile sequence  [column group x][row tile m][column tile n of group x]  (groups = the column tiles cut in eight), so its
        // L2 holds its own ~10 column tiles of W (2.5 MB) and streams the rows of X once
        const int nb = (int)gridDim.x;
        if ((a.total & 7) == 0 && (nb & 7) == 0) {
            const int per = a.total >> 3, x = (int)blockIdx.x & 7, G = x * per + ((int)blockIdx.x >> 3) + (nb >> 3) * seq;
            if (G < (x + 1) * per) {
                int g = 0;                                                               // group of G: offsets tiles_m * floor(tiles_n * g / 8)
                while (g < 7 && a.tiles_m * ((a.tiles_n * (g + 1)) >> 3) <= G) ++g;
                const int n_lo = (a.tiles_n * g) >> 3, n_cnt = ((a.tiles_n * (g + 1)) >> 3) - n_lo, r = G - a.tiles_m * n_lo;
                it.m0 = (r / n_cnt) * BM; it.n0 = (n_lo + r % n_cnt) * BN; it.kt0 = 0; it.kt1 = a.ktiles;
            }
        } else {
            const int L = (int)blockIdx.x + nb * seq;
            if (L < a.total) {
                const int n = L % a.tiles_n, m = L / a.tiles_n;
                it.m0 = m * BM; it.n0 = n * BN; it.kt0 = 0; it.kt1 = a.ktiles;
            }
        }
    } else {
        // stream-K over UNITS u = (column tile n, K-tile) of the W operand (feat: streamed from HBM once); the tiles_m row tiles of
        // a unit run in lockstep on tiles_m workgroups of one XCD (blocks b, b + 8, ...), so a unit's W K-tile is fetched once and
        // shared through that XCD's L2.  Stream q = chunk of a.total consecutive units; it crosses column tiles (segments).
        const int nb = (int)gridDim.x, nq = nb / a.tiles_m;
        int m, q;
        if ((nb & 7) == 0 && ((nb >> 3) % a.tiles_m) == 0) {
            const int s = (int)blockIdx.x >> 3;
            m = s % a.tiles_m;
            q = ((int)blockIdx.x & 7) * ((nb >> 3) / a.tiles_m) + s / a.tiles_m;
        } else {
            m = (int)blockIdx.x % a.tiles_m;
            q = (int)blockIdx.x / a.tiles_m;
        }
        (void)nq;
        const long long lo = (long long)q * a.total, hi = lo + a.total, all = (long long)a.tiles_n * a.ktiles;
        const int n = (int)(lo / a.ktiles) + seq;
        const long long b0 = (long long)n * a.ktiles, b1 = b0 + a.ktiles;
        const long long e1 = hi < b1 ? hi : b1;
        const long long s0 = lo > b0 ? lo : b0, s1 = e1 < all ? e1 : all;
        if (s0 < s1) {
            it.m0 = m * BM; it.n0 = n * BN; it.kt0 = (int)(s0 - b0); it.kt1 = (int)(s1 - b0);
            it.aux = (q * a.split + seq) * a.tiles_m + m;                       // a.split = segments a stream can touch
        }
    }
    return it;
}

__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// Transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 columns of 16-bit elements, each lane receives a
// column), as INLINE ASSEMBLY: hipcc puts `s_waitcnt vmcnt(0)` in front of the builtin form whenever a `buffer_load ... lds` is in
// flight (the intrinsic carries no memory operand, so the waitcnt pass assumes it aliases every LDS-DMA write) -- one full drain of
// the DMA queue per K-tile (seen in the weight-gradient kernel's ISA).  The assembler form is invisible to that pass; in exchange
// the kernel waits for the results itself: tr_wait() before the first use.
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ s16x4 tr_read(const lds_byte *p)
{
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory");
    return v;
}
// s_waitcnt lgkmcnt(0), tied to eight results so that nothing that reads them can be scheduled in front of it
__device__ __forceinline__ void tr_wait(s16x4 (&v)[8])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
}
__device__ __forceinline__ bf16x8 tr_join(s16x4 lo, s16x4 hi)
{
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = { lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3] };
    return __builtin_bit_cast(bf16x8, v);
}

// A producer cursor: the K-tile whose half-tiles are requested next, with the buffer descriptors of its tile rows.
template <int MODE>
struct Cursor {
    Item it;
    int seq, kt;
    bool valid;
    __amdgpu_buffer_rsrc_t rx, rw;

    // descriptor from wave-uniform values, made PROVABLY uniform (readfirstlane): otherwise hipcc wraps every buffer_load in a
    // waterfall loop
    static __device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, long long bytes)
    {
        const uint64_t b = (uint64_t)base;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        const uint32_t n = (FC1_KNOCK & 4) ? 0u : __builtin_amdgcn_readfirstlane((uint32_t)(bytes < 0 ? 0 : (bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes)));
        return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)n, 0x00020000);
    }
    __device__ __forceinline__ void bind(const Args &a)
    {
        valid = it.kt0 < it.kt1;
        // X rows past M read as zeros (the descriptor's range check); W rows always exist
        int xrows = a.M - it.m0;
        xrows = xrows < 0 ? 0 : (xrows > BM ? BM : xrows);
        rx = make_rsrc(a.X + (size_t)it.m0 * a.ldx, valid ? (long long)xrows * a.ldx * 2 : 0);
        if (MODE == MODE_DW) bind_w_rows(a);
        else rw = make_rsrc(a.W + (size_t)it.n0 * a.ldw, valid ? (long long)BN * a.ldw * 2 : 0);
    }
    // MODE_DW: W = feat [K rows][N]; the descriptor starts at K-tile `kt`'s first row and ends behind the LAST VALID row's 256
    // columns, so that rows past K (= B) fail the range check on the per-lane offset alone and read as zeros (the contraction
    // length is padded to a multiple of 64 on the X side)
    __device__ __forceinline__ void bind_w_rows(const Args &a)
    {
        const long long left = (long long)a.K - (long long)kt * BK;                 // rows of this K-tile onwards
        rw = make_rsrc(a.W + it.n0 + (size_t)kt * BK * a.ldw, valid && left > 0 ? ((left - 1) * a.ldw + BN) * 2 : 0);
    }
    __device__ __forceinline__ void start(const Args &a)
    {
        seq = 0;
        it = get_item<MODE>(a, 0);
        kt = it.kt0;
        bind(a);
    }
    __device__ __forceinline__ void advance(const Args &a)
    {
        if (!valid) return;
        if (++kt == it.kt1) {
            it = get_item<MODE>(a, ++seq);
            kt = it.kt0;
            bind(a);
        } else if (MODE == MODE_DW) {
            bind_w_rows(a);
        }
    }
};

// LDS map (bytes): NX buffers of X (32 KiB each: rows 0-127 | 128-255 of the tile), then NW buffers of W, K-tile t in buffer t % N:
//   forward          NX 3, NW 2   X = feat streams from HBM: it gets the third buffer (requests four load segments ahead of their wait)
//   weight gradient  NX 2, NW 3   W = feat streams from HBM: it gets it
//   dfeat            NX 2, NW 2   + a 4-KiB epilogue window per wave (both operands are L2-resident)
// = all 160 KiB of the CU in every mode.
[[maybe_unused]] constexpr int LDS_BUF = 32768;

template <int MODE>
__global__ __launch_bounds__(THREADS, 2) void k_fc1_gemm(Args a)
{
#if defined(__HIP_DEVICE_COMPILE__)      // the host pass only needs the launch stub (it cannot instantiate a body built on buffer descriptors)
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    lds_byte *const lds = (lds_byte *)lds_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)__builtin_amdgcn_readfirstlane((uint32_t)(tid >> 6));
    const int wr = wave >> 2, wc = wave & 3;             // wr = the wave's GROUP: rows 128 wr .. + 127 of the tile; columns 64 wc .. + 63
    constexpr bool WT = MODE == MODE_DW;                 // W operand K-strided: transposing reads
    constexpr int NX = MODE == MODE_FWD ? 3 : 2, NW = MODE == MODE_DW ? 3 : 2;
    constexpr uint32_t LDS_W0 = NX * LDS_BUF, LDS_WIN = (NX + NW) * LDS_BUF;      // LDS_WIN: dfeat only (128 KiB)

    // ---- DMA source offsets.  A half-tile is 16 pieces of 1 KiB (one wave-instruction each):
    //   K-contiguous operand: piece pc = rows 8 pc .. 8 pc + 7; this lane: row 8 pc + (lane >> 3), chunk (lane & 7) ^ ((row >> 1) & 7)
    //   K-strided operand (WT): the half-tile is [64 k-rows][128 columns]; piece = (k-block kb = pc >> 1 of 8 rows, column half pc & 1):
    //   lane: k-row 8 kb + (lane >> 3), 16-byte chunk (lane & 7) ^ (2 * f), f = ((kb & 1) << 1) | ((row >> 1) & 1)  (32-byte blocks swap)
    // W (read by both groups): every wave requests pieces 8 j + wave (j = 0, 1) of both half-tiles.  X rows 0-127 are only ever read by
    // waves 0-3 and rows 128-255 by waves 4-7, so a group requests ALL of its own half-tile: pieces 4 j + (wave & 3), j = 0..3.
    uint32_t vw[2], vx[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pc = 8 * j + wave, r = 8 * pc + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
        if (WT) {
            const int kb = pc >> 1, ch = pc & 1, kr = 8 * kb + (lane >> 3), f = ((kb & 1) << 1) | ((kr >> 1) & 1);
            vw[j] = (uint32_t)kr * (uint32_t)a.ldw * 2u + (uint32_t)ch * 128u + (uint32_t)((lane & 7) ^ (2 * f)) * 16u;
        } else {
            vw[j] = (uint32_t)r * (uint32_t)a.ldw * 2u + (uint32_t)c * 16u;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pc = 4 * j + (wave & 3), r = 8 * pc + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
        vx[j] = (uint32_t)(r + 128 * wr) * (uint32_t)a.ldx * 2u + (uint32_t)c * 16u;
    }
    const uint32_t whalf = WT ? 256u : 128u * (uint32_t)a.ldw * 2u;         // second W half-tile: 128 rows (WT: 128 columns) further

    // ---- fragment read offsets inside an X / W buffer
    //   K-contiguous: row r = 16 T + (lane & 15) of the half-tile, chunk 4 ks + (lane >> 4):
    //     (r >> 3) * 1024 + (r & 7) * 128 + ((chunk ^ ((r >> 1) & 7)) * 16)  ->  T * 2048 + lanepart(ks)
    const uint32_t lp0 = (uint32_t)(((lane >> 3) & 1) * 1024 + (lane & 7) * 128 + (((lane >> 4) ^ ((lane >> 1) & 7)) * 16));
    const uint32_t xb0 = (uint32_t)wr * HALF + lp0, xb1 = xb0 ^ 64u;
    const uint32_t wb0 = (uint32_t)(wc >> 1) * HALF + (uint32_t)(wc & 1) * 4u * 2048u + lp0, wb1 = wb0 ^ 64u;
    //   K-strided (WT): half-tile (wc >> 1) = columns 128 (wc >> 1) .. + 127 of the tile; the wave's 64 columns = column half
    //   ch = wc & 1 of every k-block.  Fragment (column tile T of 16, k-step ks): lane 4 q + p of 16-lane group g reads k-row
    //   8 g + q (lo) / 8 g + 4 + q (hi) of k-block kb = 4 ks + g, columns 16 T + 4 p .. + 3: piece (2 kb + ch), byte
    //   krow * 128 + (T ^ f(kb, krow)) * 32 + p * 8
    uint32_t wt_lo[2][4], wt_hi[2][4];
    if (WT) {
        const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                const int kb = 4 * ks + g;
                const int flo = ((kb & 1) << 1) | ((q >> 1) & 1), fhi = ((kb & 1) << 1) | (((4 + q) >> 1) & 1);
                const uint32_t pbase = (uint32_t)(wc >> 1) * HALF + (uint32_t)(2 * kb + (wc & 1)) * 1024u;
                wt_lo[ks][T] = pbase + (uint32_t)(q * 128 + ((T ^ flo) * 32) + p * 8);
                wt_hi[ks][T] = pbase + (uint32_t)((4 + q) * 128 + ((T ^ fhi) * 32) + p * 8);
            }
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };

    // cursors: the consumer (K-tile t) and the two producers -- the operand with three buffers runs two K-tiles ahead, the other one
    Cursor<MODE> cons, px, pw;
    cons.start(a);
    px = cons;
    pw = cons;

    auto req_w = [&](const Cursor<MODE> &c, int wslot) {                     // both half-tiles of W, this wave's share: 4 requests
        const uint32_t so = WT ? 0u : (uint32_t)c.kt * 128u;                  // WT: the descriptor itself walks down the rows
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rw, (__attribute__((address_space(3))) void *)(lds + LDS_W0 + (uint32_t)wslot * LDS_BUF + (uint32_t)h * HALF +
                                                                                                        (uint32_t)wave * 1024u + 8192u * j),
                                                         16, vw[j] + (h ? whalf : 0u), so, 0, 0);
    };
    auto req_x = [&](const Cursor<MODE> &c, int xslot) {                     // the group's own X half-tile: 4 requests
        const uint32_t d = (uint32_t)xslot * LDS_BUF + (uint32_t)wr * HALF + (uint32_t)(wave & 3) * 1024u, so = (uint32_t)c.kt * 128u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rx, (__attribute__((address_space(3))) void *)(lds + d + 4096u * j), 16, vx[j], so, 0, 0);
    };
    auto seg = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: K-tile 0 complete; the three-buffer operand's K-tile 1 in flight
    req_x(px, 0);
    req_w(pw, 0);
    px.advance(a);
    pw.advance(a);
    if (NX == 3) {
        if (px.valid && !(FC1_KNOCK & 2)) {
            req_x(px, 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        px.advance(a);
    } else if (NW == 3) {
        if (pw.valid && !(FC1_KNOCK & 2)) {
            req_w(pw, 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        pw.advance(a);
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one barrier behind group 0 from here on

    // A finished tile -> memory, accumulators back to zero.  Runs behind the last MFMA (and the last DMA wait) of the item's last K-tile.
    //   f32 outputs (split-K slabs, stream-K segments): 16 B per lane straight from the accumulator layout (4 consecutive columns).
    //   bf16 output (dfeat, 629 MB per call): per lane only 8 B, 16 different 128-byte lines per store instruction -- measured: the
    //   store ISSUE (line transactions), not HBM, then costs as much as the whole product.  So 32-row slices of the wave's 128 x 64
    //   tile go through a private 4-KB LDS window and leave as whole 128-byte rows, 16 B per lane, 8 rows per store instruction,
    //   written through.  Window image: [32 rows][128 B], 16-byte chunk ^ (row & 7), 8-byte half ^ (row >> 3 & 1): ds_write_b64 and
    //   ds_read_b128 both conflict-free.
    auto epilogue = [&](const Item &it) {
        if (MODE == MODE_DFEAT) {
            int rows = a.M - it.m0;
            rows = rows > BM ? BM : rows;
            const __amdgpu_buffer_rsrc_t ro = Cursor<MODE>::make_rsrc(reinterpret_cast<bf16 *>(a.out) + (size_t)it.m0 * a.ldo + it.n0, (long long)rows * a.ldo * 2);
            lds_byte *const win = lds + LDS_WIN + (uint32_t)wave * 4096u;
            const int r16 = lane & 15, g = lane >> 4, rr = lane >> 3, c = lane & 7;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        bf16x4 h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = (bf16)acc[2 * u + mi][nt][e];
                        const int row = 16 * mi + r16;
                        *(__attribute__((address_space(3))) bf16x4 *)(win + row * 128 + (((2 * nt + (g >> 1)) ^ (row & 7)) * 16) + (((g & 1) ^ (r16 >> 3)) * 8)) = h;
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 8 * i + rr;
                    typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
                    v4u32 v = *(const __attribute__((address_space(3))) v4u32 *)(win + row * 128 + ((c ^ (row & 7)) * 16));
                    if (i & 1) v = v4u32{ v[2], v[3], v[0], v[1] };              // rows 8-15 / 24-31 of the window hold their 8-byte halves swapped
                    const uint32_t off = (uint32_t)(wr * 128 + 32 * u + row) * (uint32_t)a.ldo * 2u + (uint32_t)(wc * 64 + c * 8) * 2u;
                    __builtin_amdgcn_raw_buffer_store_b128(v, ro, off, 0, 16);    // aux 16 = sc1: whole lines, written through
                }
            }
        } else {
            float *o;
            int mrow, mlim;
            if (MODE == MODE_DW) {
                // stream-K segment: f32 [256 j][256 n] of this tile, row-major
                o = reinterpret_cast<float *>(a.out) + (size_t)it.aux * (BM * BN) + wc * 64 + 4 * (lane >> 4);
                mrow = wr * 128 + (lane & 15);
                mlim = BM;
            } else {
                o = reinterpret_cast<float *>(a.out) + (size_t)it.aux * a.M * a.ldo + it.n0 + wc * 64 + 4 * (lane >> 4);
                mrow = it.m0 + wr * 128 + (lane & 15);
                mlim = a.M;
            }
            const int ld = MODE == MODE_DW ? BN : a.ldo;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const int m = mrow + 16 * mt;
                if (m < mlim) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4 *>(o + (size_t)m * ld + 16 * nt) = acc[mt][nt];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    };

    // ---- main loop: one K-tile = two phases of [fragment reads + 4 DMA requests] s_barrier [32 MFMAs] s_barrier
    // Phase 0 reads all of W's fragments (both 32-column halves of the wave's tile stay in registers for the K-tile) and X rows 0-63;
    // phase 1 reads X rows 64-127.  Segments = barrier intervals; group 0 loads in segments 4 t and 4 t + 2, group 1 one later.
    // Requests (four per load segment wherever an operand has three buffers -- a request costs its wave ~100 cycles of issue time, and
    // eight of them in one load segment made that segment twice as long as the partner's 32 MFMAs, profiles/r04_fc1_knockouts.txt):
    //   weight gradient (NW 3): phase 0 requests the group's X half-tile of K-tile t + 1; phase 1 waits vmcnt(4) = W of t + 1 (requested
    //          a K-tile ago), then requests W of t + 2; behind the last MFMA vmcnt(4) = X of t + 1.
    //   forward (NX 3): phase 0 requests W of K-tile t + 1; phase 1 requests the group's X of t + 2, then waits vmcnt(4) = X of t + 1
    //          (requested a K-tile ago: the HBM stream gets four segments of flight) and W of t + 1 (two segments; it comes from L2).
    //   dfeat (2 + 2): phase 0 requests W and X of K-tile t + 1 (nothing of t + 2 has a free slot yet); phase 1 waits vmcnt(4) = W,
    //          behind the last MFMA vmcnt(0) = X.  Its operands come from L2, and moving X's requests to phase 1 measured equal.
    // Hazards:  a W buffer is re-filled >= 2 segments after its last fragment read (the readers' lgkmcnt waits and a barrier lie in
    // between); an X half-tile is re-filled by the group that read it, one barrier after that group's last read; a wave waits for its
    // own requests (counted vmcnt) at least one barrier before any wave reads them.
    bf16x8 xf[2][4], wf[2][4];
    int xs = 0, ws = 0;                                   // X buffer (t % NX), W buffer (t % NW)
    auto nxt = [](int v, int n, int d) { v += d; return v >= n ? v - n : v; };
    while (cons.valid) {
        const lds_byte *const lx = lds + (uint32_t)xs * LDS_BUF, *const lw = lds + LDS_W0 + (uint32_t)ws * LDS_BUF;
        const bool fresh = !(FC1_KNOCK & 1) || cons.kt == cons.it.kt0;       // probe 1: fragments stay what the first K-tile loaded
        // phase 0
        if (fresh) {
            s16x4 tlo[8], thi[8];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                if (WT) {
                    tlo[nt] = tr_read(lw + wt_lo[0][nt]);
                    thi[nt] = tr_read(lw + wt_hi[0][nt]);
                    tlo[4 + nt] = tr_read(lw + wt_lo[1][nt]);
                    thi[4 + nt] = tr_read(lw + wt_hi[1][nt]);
                } else {
                    wf[0][nt] = *(const __attribute__((address_space(3))) bf16x8 *)(lw + wb0 + (uint32_t)nt * 2048u);
                    wf[1][nt] = *(const __attribute__((address_space(3))) bf16x8 *)(lw + wb1 + (uint32_t)nt * 2048u);
                }
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                xf[0][mt] = *(const __attribute__((address_space(3))) bf16x8 *)(lx + xb0 + (uint32_t)mt * 2048u);
                xf[1][mt] = *(const __attribute__((address_space(3))) bf16x8 *)(lx + xb1 + (uint32_t)mt * 2048u);
            }
            if (WT) {
                tr_wait(tlo);
                tr_wait(thi);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    wf[0][nt] = tr_join(tlo[nt], thi[nt]);
                    wf[1][nt] = tr_join(tlo[4 + nt], thi[4 + nt]);
                }
            }
        }
        if (!(FC1_KNOCK & 2)) {
            if (NW == 2 && pw.valid) req_w(pw, ws ^ 1);                  // W of K-tile t + 1
            if (NX == 2 && px.valid) req_x(px, xs ^ 1);                  // X of K-tile t + 1
        }
        seg();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma(wf[ks][nt], xf[ks][mt], acc[mt][nt]);
        __builtin_amdgcn_s_setprio(0);
        seg();
        // phase 1
        if (fresh) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                xf[0][mt] = *(const __attribute__((address_space(3))) bf16x8 *)(lx + xb0 + (uint32_t)(4 + mt) * 2048u);
                xf[1][mt] = *(const __attribute__((address_space(3))) bf16x8 *)(lx + xb1 + (uint32_t)(4 + mt) * 2048u);
            }
        }
        if (NX == 3) {
            // forward: X of K-tile t + 2 goes out first, then the wait for everything older: X of t + 1 (requested a K-tile ago) and
            // W of t + 1 (phase 0).  W cannot wait until behind the MFMAs as X does in the other modes: W is read by BOTH groups, and
            // group 0's phase 0 of K-tile t + 1 runs in the same segment as group 1's phase-1 MFMAs of K-tile t.
            const bool more = px.valid && !(FC1_KNOCK & 2);
            if (more) req_x(px, nxt(xs, 3, 2));
            if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (px.valid && !(FC1_KNOCK & 2)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // W of K-tile t + 1 (its X, requested after it, may still fly)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (NW == 3 && pw.valid && !(FC1_KNOCK & 2)) req_w(pw, nxt(ws, 3, 2));               // W of K-tile t + 2 into the buffer K-tile t - 1 left
        }
        seg();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[4 + mt][nt] = mfma(wf[ks][nt], xf[ks][mt], acc[4 + mt][nt]);
        __builtin_amdgcn_s_setprio(0);
        if (NX == 2) {                                                   // the group's own X of K-tile t + 1 (weight gradient: W of t + 2 may still fly)
            if (NW == 3 && pw.valid && !(FC1_KNOCK & 2)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        px.advance(a);
        pw.advance(a);
        if (cons.kt + 1 == cons.it.kt1) epilogue(cons.it);
        cons.advance(a);
        seg();
        xs = nxt(xs, NX, 1);
        ws = nxt(ws, NW, 1);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
#else
    (void)a;
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient, second half: sum the stream-K segment slabs of every tile and write the f32 gradient in the MODULE's layout
// grad_w[j][c P + p] (nn.Flatten's order of the reference's features, policies.py:102) from the kernels' feature order p 128 + c,
// plus the four extra-feature columns (dwe, from inv_heads_bwd).  One workgroup = (row j, chunk of 32 pixels): 16 K-contiguous
// 1-KB rows per covering segment in, 128 runs of 32 consecutive floats out, transposed through LDS.
struct DwReduceArgs {
    const float *segs;                   // [streams][maxseg][tiles_m][256][256]
    const float *dwe;                    // [rows][4]
    float *gw[2];                        // per head: [256][128 P + 4]
    int P, tiles_m, ktiles, per, maxseg, nstreams;
};

__global__ __launch_bounds__(256) void k_fc1_dw_reduce(DwReduceArgs r)
{
    __shared__ float tl[128 * 33];
    const int j = blockIdx.y, p0 = blockIdx.x * 32, tid = threadIdx.x;
    const int np = min(32, r.P - p0);
    const int mt = j >> 8, jr = j & 255;
    // features [128 p0, 128 (p0 + np)): 4 per thread and pass; a column tile of 256 = 2 pixels
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = it * 256 + tid, dp = idx >> 5;                         // pixel p0 + dp, channels 4 (idx & 31) .. + 3
        if (dp < np) {
            const int f = (p0 + dp) * 128 + 4 * (idx & 31), n = f >> 8;
            const long long b0 = (long long)n * r.ktiles, b1 = b0 + r.ktiles;
            const int q0 = (int)(b0 / r.per), q1 = (int)((b1 - 1) / r.per);
            float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (int q = q0; q <= q1 && q < r.nstreams; ++q) {
                const int seq = n - (int)(((long long)q * r.per) / r.ktiles);
                const float4 v = *reinterpret_cast<const float4 *>(r.segs + (((size_t)(q * r.maxseg + seq) * r.tiles_m + mt) * 256 + jr) * 256 + (f & 255));
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            const int c = 4 * (idx & 31);
            tl[(c + 0) * 33 + dp] = s.x; tl[(c + 1) * 33 + dp] = s.y; tl[(c + 2) * 33 + dp] = s.z; tl[(c + 3) * 33 + dp] = s.w;
        }
    }
    __syncthreads();
    const int nf = 128 * r.P, ld = nf + 4;
    float *dst = r.gw[j >> 8] + (size_t)(j & 255) * ld;
    for (int i = tid; i < 128 * 32; i += 256) {
        const int c = i >> 5, dp = i & 31;
        if (dp < np) dst[c * r.P + p0 + dp] = tl[c * 33 + dp];
    }
    if (blockIdx.x == 0 && tid < 4) dst[nf + tid] = r.dwe[j * 4 + tid];
}

} // namespace fc1

static int fc1_status(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return inv_fail(INV_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return INV_OK;
}

template <int MODE>
static int fc1_launch(const fc1::Args &a, int grid, hipStream_t st, const char *what)
{
    // per launch: a cheap host call, and correct on every device ordinal and from every host thread (a cached "done" flag per device
    // would be a data race between threads and wrong past its table size)
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fc1::k_fc1_gemm<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, fc1::LDS_BYTES);
    if (e != hipSuccess) return inv_fail(INV_ERR_HIP, "%s: hipFuncSetAttribute(%d B of LDS): %s", what, fc1::LDS_BYTES, hipGetErrorString(e));
    hipLaunchKernelGGL((fc1::k_fc1_gemm<MODE>), dim3(grid), dim3(fc1::THREADS), fc1::LDS_BYTES, st, a);
    return fc1_status(what);
}

static int fc1_fwd_split(int B, int nout, int K)
{
    const int tiles = ((B + fc1::BM - 1) / fc1::BM) * (nout / fc1::BN), ktiles = K / fc1::BK;
    int s = inv_cu_count() / (tiles > 0 ? tiles : 1);
    if (s > 8) s = 8;
    if (s > ktiles / 8) s = ktiles / 8;
    return s < 1 ? 1 : s;
}

extern "C" int32_t inv_fc1_fwd_slabs(int32_t B, int32_t nout, int32_t K)
{
    if (B <= 0 || nout <= 0 || nout % fc1::BN || K <= 0 || K % fc1::BK) return 0;
    return fc1_fwd_split(B, nout, K);
}

extern "C" int inv_fc1_fwd(const void *feat_bf16, const void *wf_bf16, int32_t B, int32_t nout, int32_t K, float *pre_slabs, void *stream)
{
    if (!feat_bf16 || !wf_bf16 || !pre_slabs || B < 0 || nout <= 0 || nout % fc1::BN || K <= 0 || K % fc1::BK)
        return inv_fail(INV_ERR_BAD_ARG, "inv_fc1_fwd: NULL pointer, B < 0, nout not a multiple of 256 or K not a multiple of 64");
    if (B == 0) return INV_OK;
    if ((long long)fc1::BM * K * 2 > 0x7fffffffLL) return inv_fail(INV_ERR_BAD_ARG, "inv_fc1_fwd: K too large for 32-bit tile offsets");
    fc1::Args a = {};
    a.X = reinterpret_cast<const fc1::bf16 *>(feat_bf16);
    a.W = reinterpret_cast<const fc1::bf16 *>(wf_bf16);
    a.out = pre_slabs;
    a.M = B; a.N = nout; a.K = K; a.ldx = K; a.ldw = K; a.ldo = nout;
    a.tiles_m = (B + fc1::BM - 1) / fc1::BM; a.tiles_n = nout / fc1::BN; a.ktiles = K / fc1::BK;
    a.split = fc1_fwd_split(B, nout, K);
    a.total = a.tiles_m * a.tiles_n * a.split;
    return fc1_launch<fc1::MODE_FWD>(a, a.total, (hipStream_t)stream, "inv_fc1_fwd");
}

extern "C" int inv_fc1_dfeat(const void *dpre_bf16, const void *wft_bf16, int32_t B, int32_t nhid, int32_t nfeat, void *dfeat_bf16, void *stream)
{
    if (!dpre_bf16 || !wft_bf16 || !dfeat_bf16 || B < 0 || nhid <= 0 || nhid % fc1::BK || nfeat <= 0 || nfeat % fc1::BN)
        return inv_fail(INV_ERR_BAD_ARG, "inv_fc1_dfeat: NULL pointer, B < 0, nhid not a multiple of 64 or nfeat not a multiple of 256");
    if (B == 0) return INV_OK;
    fc1::Args a = {};
    a.X = reinterpret_cast<const fc1::bf16 *>(dpre_bf16);
    a.W = reinterpret_cast<const fc1::bf16 *>(wft_bf16);
    a.out = dfeat_bf16;
    a.M = B; a.N = nfeat; a.K = nhid; a.ldx = nhid; a.ldw = nhid; a.ldo = nfeat;
    a.tiles_m = (B + fc1::BM - 1) / fc1::BM; a.tiles_n = nfeat / fc1::BN; a.ktiles = nhid / fc1::BK;
    a.split = 1;
    a.total = a.tiles_m * a.tiles_n;
    const int cus = inv_cu_count();
    return fc1_launch<fc1::MODE_DFEAT>(a, a.total < cus ? a.total : cus, (hipStream_t)stream, "inv_fc1_dfeat");
}

struct Fc1DwPlan { int tiles_m, tiles_n, ktiles, per, streams, maxseg; };
static Fc1DwPlan fc1_dw_plan(int Bpad, int nhid, int nfeat)
{
    Fc1DwPlan p;
    p.tiles_m = nhid / fc1::BM; p.tiles_n = nfeat / fc1::BN; p.ktiles = Bpad / fc1::BK;
    const long long units = (long long)p.tiles_n * p.ktiles;                      // (column tile, K-tile) pairs, each worked on by tiles_m workgroups
    int want = inv_cu_count() / p.tiles_m;
    if (want < 1) want = 1;
    p.per = (int)((units + want - 1) / want);
    p.streams = (int)((units + p.per - 1) / p.per);
    p.maxseg = p.per / p.ktiles + 2;
    return p;
}

extern "C" int64_t inv_fc1_dw_workspace_bytes(int32_t B, int32_t nhid, int32_t nfeat)
{
    if (B <= 0 || nhid <= 0 || nhid % fc1::BM || nfeat <= 0 || nfeat % fc1::BN) return 0;
    const Fc1DwPlan p = fc1_dw_plan((B + 63) / 64 * 64, nhid, nfeat);
    return (int64_t)p.streams * p.maxseg * p.tiles_m * fc1::BM * fc1::BN * (int64_t)sizeof(float);
}

extern "C" int inv_fc1_dw(const void *dpre_t_bf16, const void *feat_bf16, const float *dwe, int32_t B, int32_t nhid, int32_t pixels,
                          float *grad_w_actor, float *grad_w_critic, void *workspace, void *stream)
{
    const int nfeat = 128 * pixels, Bpad = (B + 63) / 64 * 64;
    if (!dpre_t_bf16 || !feat_bf16 || !dwe || !grad_w_actor || (nhid > 256 && !grad_w_critic) || !workspace || B < 0 || (nhid != 256 && nhid != 512) ||
        pixels <= 0 || nfeat % fc1::BN)
        return inv_fail(INV_ERR_BAD_ARG, "inv_fc1_dw: NULL pointer, B < 0, nhid not 256 | 512 or 128 * pixels not a multiple of 256");
    if (B == 0) return INV_OK;
    if ((long long)fc1::BM * Bpad * 2 > 0x7fffffffLL || 64LL * nfeat * 2 > 0x7fffffffLL) return inv_fail(INV_ERR_BAD_ARG, "inv_fc1_dw: B or pixels too large for 32-bit tile offsets");
    const Fc1DwPlan p = fc1_dw_plan(Bpad, nhid, nfeat);
    fc1::Args a = {};
    a.X = reinterpret_cast<const fc1::bf16 *>(dpre_t_bf16);      // [nhid][Bpad], columns past B zero
    a.W = reinterpret_cast<const fc1::bf16 *>(feat_bf16);        // [B][nfeat]
    a.out = workspace;
    a.M = nhid; a.N = nfeat; a.K = B; a.ldx = Bpad; a.ldw = nfeat; a.ldo = fc1::BN;
    a.tiles_m = p.tiles_m; a.tiles_n = p.tiles_n; a.ktiles = p.ktiles; a.split = p.maxseg; a.total = p.per;
    const int rc = fc1_launch<fc1::MODE_DW>(a, p.streams * p.tiles_m, (hipStream_t)stream, "inv_fc1_dw");
    if (rc != INV_OK) return rc;
    fc1::DwReduceArgs r = { reinterpret_cast<const float *>(workspace), dwe, { grad_w_actor, grad_w_critic }, pixels, p.tiles_m, p.ktiles, p.per, p.maxseg, p.streams };
    hipLaunchKernelGGL(fc1::k_fc1_dw_reduce, dim3((pixels + 31) / 32, nhid), dim3(256), 0, (hipStream_t)stream, r);
    return fc1_status("inv_fc1_dw (reduce)");
}
