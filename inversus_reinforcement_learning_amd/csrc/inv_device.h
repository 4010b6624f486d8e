// inv_device.h -- device-side building blocks of the gfx950 Inversus env kernels.
//
// Execution model: ONE WAVEFRONT (64 lanes) PER ENV.  Per-env scalars (players, ammo,
// counters, reward) are wave-uniform and live in SGPRs (the wave index is made uniform with
// readfirstlane); lane i < K owns bullet slot i; the tile grid (1 bit per tile, row-major
// bitstring) and the packed observation bitstring are staged in the wave's private LDS
// slice.  No workgroup barriers anywhere: waves of a block never communicate, only lanes of
// one wave do.
//
// The kernels are instruction-issue bound before they are HBM bound (measured: ~1700
// instructions per env-step in the first, runtime-dimensioned version), so every function
// is templated on a dims policy: StaticDims<W,H,K> folds all index arithmetic, loop bounds
// and masks for the configurations that matter (15x10 and 32x32), RuntimeDims serves any
// other 3..64 x 3..64 grid with the same code.
//
// Reference semantics restated here (file:line relative to the reference checkout):
//   inversus/core.py:55-154 (reset), :249-381 (move / shoot / wide shot), :383-397
//   (reload), :399-475 (bullet advance, cancel, merge, flip, kill), :497-531 (tick order);
//   inversus_rl/env_wrappers.py:20-66 (action decode), :69-170 (dummy opponent),
//   :173-245 (12-channel obs), :318-442 (reward / done / info).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/inversus_hip.h"

namespace inv {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kHeadWords = 8;      // per-env head record, 32 B
constexpr int kMtStride = 640;     // u32 words per env in the MT array (624 used), 2560 B = 20 x 128-B lines
constexpr int kTabWords = 64;      // nibble -> float4 table, 16 x 16 B

// head record words
enum { HD_P1 = 0, HD_P2 = 1, HD_FLAGS = 2, HD_STEP = 3, HD_OPP = 4, HD_MTI = 5, HD_RET_LO = 6, HD_RET_HI = 7 };
// HD_P1 / HD_P2: x | y<<8 | ammo<<16 | reload<<24
// HD_FLAGS: bit0 alive1, bit1 alive2, bit2 prev_alive1, bit3 prev_alive2, bits 8..15 live bullet count
constexpr uint32_t FL_ALIVE1 = 1u, FL_ALIVE2 = 2u, FL_PREV1 = 4u, FL_PREV2 = 8u;

// sticky device error bits (inv_check)
constexpr uint32_t ERR_BAD_ACTION = 1u, ERR_BULLET_OVERFLOW = 2u;

// ------------------------------------------------------------------ dims policies
struct Dims {            // runtime description (host fills it; also the RuntimeDims payload)
    int N, W, H, WH, K, GW;
    int OW;              // u32 words of one packed observation bitstring: ceil(12*WH/32)
    int F4;              // float4 stores per observation: 3*WH
    int scr_words;       // LDS words per wave
    int max_steps;
};

template <int W_, int H_, int K_>
struct StaticDims {
    static constexpr int W = W_, H = H_, WH = W_ * H_, K = K_, GW = (W_ * H_ + 31) / 32;
    static constexpr int OW = (12 * W_ * H_ + 31) / 32, F4 = 3 * W_ * H_;
    static constexpr int SCR = (OW > 624 ? OW : 624);
    static constexpr int scr_words = (kTabWords + GW + K + SCR + 3) & ~3;
    __device__ __host__ explicit StaticDims(const Dims &) {}
};

struct RuntimeDims {
    int W, H, WH, K, GW, OW, F4, scr_words;
    __device__ __host__ explicit RuntimeDims(const Dims &d)
        : W(d.W), H(d.H), WH(d.WH), K(d.K), GW(d.GW), OW(d.OW), F4(d.F4), scr_words(d.scr_words) {}
};

inline int scr_words_for(int GW, int K, int OW) { return (kTabWords + GW + K + (OW > 624 ? OW : 624) + 3) & ~3; }

struct State {
    uint32_t *head;      // [N][8]
    uint32_t *grid;      // [N][GW]
    uint16_t *bullets;   // [N][K]
    uint32_t *mt;        // [N][kMtStride]
    uint32_t *err;       // [1]
    const double *prox;  // [W+H+1]: 0.002 * (1.0 - dist / (W+H)), env_wrappers.py:382, computed on the host in IEEE f64
};

struct StepArgs {
    Dims d;
    State s;
    const uint8_t *mask;  // optional: step only envs with a non-zero byte
    const uint8_t *a1;
    const uint8_t *a2;
    inv_step_out o;
    uint32_t flags;
    int opponent, hard;
    uint32_t opp_seed;
    uint32_t env_offset;
    int T;               // vector steps in this launch (inv_rollout); 1 for inv_step
    int obs_wt;          // f32 observation stores write through (every store instruction of the launch covers whole 128-B lines)
};

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one), each with its own L2.  Consecutive envs
// share cache lines of every output array (reward / done / info: 32..128 envs per 128-B line; the 7 200-B observation regions
// of neighbouring envs meet inside a line), so with the identity mapping every such line is written in pieces by several
// L2s and reaches HBM as partial-line writes.  This permutation gives each XCD a contiguous range of blocks.  Speed only.
#ifndef INV_NO_XCD_REMAP
__device__ __forceinline__ int xcd_block(int b, int nblocks)
{
    const int per = nblocks >> 3;
    return b < (per << 3) ? (b & 7) * per + (b >> 3) : b;
}
#else
__device__ __forceinline__ int xcd_block(int b, int) { return b; }
#endif

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int rfl(int v) { return (int)__builtin_amdgcn_readfirstlane((uint32_t)v); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return __builtin_amdgcn_readlane(v, l); }

// Orders this wave's LDS traffic: LDS ops of one wave execute in issue order, so the only
// thing needed is that the compiler keeps that order.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave-uniform reads of read-only inputs through the SCALAR cache (s_load) instead of a vector load + readfirstlane:
// they do not queue behind the CU's vector stores of the observation stream.  Safe for data no kernel on the
// stream writes while this kernel runs (actions, masks, the proximity table).
typedef const __attribute__((address_space(4))) uint32_t *const_u32_ptr;
typedef const __attribute__((address_space(4))) double *const_f64_ptr;

// The byte is fetched as the ALIGNED 32-bit word that holds it, so up to 3 bytes beyond a buffer whose end is not 4-byte
// aligned are read (never used).  They are inside the allocation for every allocator this library is fed from (hipMalloc
// and torch round sizes up to >= 256 B); include/inversus_hip.h states the requirement for other callers.
__device__ __forceinline__ uint32_t sload_u8(const uint8_t *p, size_t idx)
{
    const uintptr_t a = (uintptr_t)p + idx;
    const uint32_t w = *(const_u32_ptr)(a & ~(uintptr_t)3);        // the aligned word that holds the byte
    return (w >> (8u * (uint32_t)(a & 3u))) & 0xffu;
}

__device__ __forceinline__ double sload_f64(const double *p, int idx) { return ((const_f64_ptr)p)[idx]; }

__device__ __forceinline__ uint32_t tile_bit(const uint32_t *g, int pos) { return (g[pos >> 5] >> (pos & 31)) & 1u; }

__device__ __forceinline__ uint32_t pack_bullet(int x, int y, int dir, int owner)
{
    return (uint32_t)x | ((uint32_t)y << 6) | ((uint32_t)dir << 12) | ((uint32_t)owner << 14);
}

// direction code (0 UP, 1 RIGHT, 2 DOWN, 3 LEFT) -> step, from 2-bit LUTs holding step + 1
__device__ __forceinline__ int dir_dx(int dir) { return (int)((0x19u >> (2 * dir)) & 3u) - 1; }   // 0, +1, 0, -1
__device__ __forceinline__ int dir_dy(int dir) { return (int)((0x64u >> (2 * dir)) & 3u) - 1; }   // -1, 0, +1, 0

// ------------------------------------------------------------------ per-wave env context
struct Player { int x, y, ammo, reload, alive; };

struct Env {
    Player p[2];
    int prev1, prev2;
    int nb;
    int step_count;
    uint32_t opp_ctr;
    int mti;
    double ep_ret;
    uint32_t bw;         // lane i < nb: bullet i (canonical u16 packing)
};

template <class D>
__device__ __forceinline__ void load_env(const D &d, const State &s, int env, int lane, uint32_t *g, Env &e)
{
    const uint32_t *hp = s.head + (size_t)env * kHeadWords;
    const uint32_t *gp = s.grid + (size_t)env * d.GW;
    const uint16_t *bp = s.bullets + (size_t)env * d.K;
    // branch-free: every lane loads (clamped index), so no exec-mask juggling around the loads
    const uint32_t hv = hp[lane & (kHeadWords - 1)];
    const uint32_t bwv = (uint32_t)bp[min(lane, d.K - 1)];
    if (d.GW <= kWave) {
        const uint32_t gv = gp[min(lane, d.GW - 1)];
        if (lane < d.GW) g[lane] = gv;
    } else {
        for (int w = lane; w < d.GW; w += kWave) g[w] = gp[w];
    }
    uint32_t p1 = rdlane(hv, HD_P1), p2 = rdlane(hv, HD_P2), fl = rdlane(hv, HD_FLAGS);
    e.p[0] = { (int)(p1 & 255u), (int)((p1 >> 8) & 255u), (int)((p1 >> 16) & 255u), (int)(p1 >> 24), (int)(fl & FL_ALIVE1) != 0 };
    e.p[1] = { (int)(p2 & 255u), (int)((p2 >> 8) & 255u), (int)((p2 >> 16) & 255u), (int)(p2 >> 24), (int)(fl & FL_ALIVE2) != 0 };
    e.prev1 = (fl & FL_PREV1) != 0;
    e.prev2 = (fl & FL_PREV2) != 0;
    e.nb = (int)((fl >> 8) & 255u);
    e.step_count = (int)rdlane(hv, HD_STEP);
    e.opp_ctr = rdlane(hv, HD_OPP);
    e.mti = (int)rdlane(hv, HD_MTI);
    uint64_t rb = (uint64_t)rdlane(hv, HD_RET_LO) | ((uint64_t)rdlane(hv, HD_RET_HI) << 32);
    e.ep_ret = __longlong_as_double((long long)rb);
    e.bw = lane < e.nb ? bwv : 0u;
    wave_sync();
}

template <class D>
__device__ __forceinline__ void store_env(const D &d, const State &s, int env, int lane, const uint32_t *g, const Env &e)
{
    uint32_t *hp = s.head + (size_t)env * kHeadWords;
    uint32_t *gp = s.grid + (size_t)env * d.GW;
    uint16_t *bp = s.bullets + (size_t)env * d.K;
    uint64_t rb = (uint64_t)__double_as_longlong(e.ep_ret);
    uint32_t fl = (e.p[0].alive ? FL_ALIVE1 : 0u) | (e.p[1].alive ? FL_ALIVE2 : 0u) | (e.prev1 ? FL_PREV1 : 0u) |
                  (e.prev2 ? FL_PREV2 : 0u) | ((uint32_t)e.nb << 8);
    uint32_t p1 = (uint32_t)e.p[0].x | ((uint32_t)e.p[0].y << 8) | ((uint32_t)e.p[0].ammo << 16) | ((uint32_t)e.p[0].reload << 24);
    uint32_t p2 = (uint32_t)e.p[1].x | ((uint32_t)e.p[1].y << 8) | ((uint32_t)e.p[1].ammo << 16) | ((uint32_t)e.p[1].reload << 24);
    uint32_t hv = lane == HD_P1 ? p1 : lane == HD_P2 ? p2 : lane == HD_FLAGS ? fl : lane == HD_STEP ? (uint32_t)e.step_count
                : lane == HD_OPP ? e.opp_ctr : lane == HD_MTI ? (uint32_t)e.mti : lane == HD_RET_LO ? (uint32_t)rb : (uint32_t)(rb >> 32);
    if (lane < kHeadWords) hp[lane] = hv;
    for (int w = lane; w < d.GW; w += kWave) gp[w] = g[w];
    if (lane < d.K) bp[lane] = (uint16_t)(lane < e.nb ? e.bw : 0u);
}

// ------------------------------------------------------------------ actions (core.py:249-381, 510-525)
template <class D>
__device__ __forceinline__ void emit_bullet(const D &d, int lane, Env &e, uint32_t word, bool valid, uint32_t &err)
{
    if (!valid) return;
    if (e.nb >= d.K) { err |= ERR_BULLET_OVERFLOW; return; }
    if (lane == e.nb) e.bw = word;
    e.nb += 1;
}

template <class D>
__device__ __forceinline__ void apply_action(const D &d, const uint32_t *g, int lane, int pid, int a, Env &e, uint32_t &err)
{
    Player &p = e.p[pid];
    if (!p.alive || a == 0) return;
    const int type = (a + 3) >> 2, dir = (a - 1) & 3;          // env_wrappers.py:24-37
    if (type == 1) {                                           // MOVE, core.py:249-296
        const int nx = p.x + dir_dx(dir), ny = p.y + dir_dy(dir);
        if ((unsigned)nx < (unsigned)d.W && (unsigned)ny < (unsigned)d.H) {
            // walkable <=> tile != player's colour; P1 is BLACK (bit 0), P2 is WHITE (bit 1)
            if (rfl(tile_bit(g, ny * d.W + nx)) != (uint32_t)pid) { p.x = nx; p.y = ny; }
        }
    } else if (type == 2) {                                    // SHOOT, core.py:298-326
        if (p.ammo > 0) {
            p.ammo -= 1;
            emit_bullet(d, lane, e, pack_bullet(p.x, p.y, dir, pid), true, err);
        }
    } else {                                                   // CHARGE_SHOOT, core.py:328-381
        if (p.ammo >= 3) {
            p.ammo -= 3;
            const int vertical = (dir & 1) == 0;               // UP / DOWN: side lanes at x-1, x+1; else y-1, y+1
            const int ax = vertical ? p.x - 1 : p.x, ay = vertical ? p.y : p.y - 1;
            const int bx = vertical ? p.x + 1 : p.x, by = vertical ? p.y : p.y + 1;
            emit_bullet(d, lane, e, pack_bullet(p.x, p.y, dir, pid), true, err);
            emit_bullet(d, lane, e, pack_bullet(ax, ay, dir, pid), ax >= 0 && ay >= 0, err);
            emit_bullet(d, lane, e, pack_bullet(bx, by, dir, pid), bx < d.W && by < d.H, err);
        }
    }
}

__device__ __forceinline__ void reload(Player &p)               // core.py:383-397
{
    if (!p.alive) return;
    if (p.ammo < 6) {
        p.reload += 1;
        if (p.reload >= 30) { p.ammo += 1; p.reload = 0; }
    }
}

// ------------------------------------------------------------------ bullets (core.py:399-475)
// Returns tile_diff = (#tiles turned WHITE) - (#tiles turned BLACK) of this update.
template <class D>
__device__ __forceinline__ int update_bullets(const D &d, uint32_t *g, uint32_t *slots, int lane, Env &e)
{
    if (e.nb == 0) return 0;                                                       // wave-uniform shortcut
    const int bx = (int)(e.bw & 63u), by = (int)((e.bw >> 6) & 63u), dir = (int)((e.bw >> 12) & 3u), own = (int)((e.bw >> 14) & 1u);
    const int nx = bx + dir_dx(dir), ny = by + dir_dy(dir);
    const bool live = lane < e.nb && (unsigned)nx < (unsigned)d.W && (unsigned)ny < (unsigned)d.H;   // :428-431
    const int pos = live ? ny * d.W + nx : -1;
    const int key = live ? (pos | (own << 16)) : -1;                               // tile + owner in one compare

    // Tile groups: a bullet meeting a different owner's bullet on its target tile vanishes
    // (:444-449); among same-owner arrivals only the first in list order survives (:453).
    bool cancel = false, dup = false;
    uint64_t m = __ballot(live);
    while (m) {
        const int j = __ffsll((unsigned long long)m) - 1;
        m &= m - 1;
        const int kj = (int)rdlane((uint32_t)key, j);
        cancel |= (key ^ kj) == 0x10000;                                           // same tile, other owner
        dup |= key == kj && j < lane;                                              // same tile, same owner, earlier in the list
    }
    const bool keep = live && !cancel && !dup;

    // flip only tiles of the owner's colour (:458-461): P1 turns BLACK->WHITE, P2 WHITE->BLACK
    uint32_t tb = 0u;
    if (keep) tb = tile_bit(g, pos);
    const bool f1 = keep && own == 0 && tb == 0u;
    const bool f2 = keep && own == 1 && tb == 1u;
    wave_sync();
    if (f1) atomicOr(&g[pos >> 5], 1u << (pos & 31));
    if (f2) atomicAnd(&g[pos >> 5], ~(1u << (pos & 31)));
    const int tile_diff = __popcll(__ballot(f1)) - __popcll(__ballot(f2));

    // kills (:464-470): a surviving bullet on an alive non-owner's tile
    const int key1 = (e.p[0].y * d.W + e.p[0].x) | 0x10000, key2 = e.p[1].y * d.W + e.p[1].x;
    if (e.p[0].alive && __ballot(keep && key == key1) != 0ull) e.p[0].alive = 0;  // P2's bullet on P1's tile
    if (e.p[1].alive && __ballot(keep && key == key2) != 0ull) e.p[1].alive = 0;  // P1's bullet on P2's tile

    // stable compaction: survivors keep list order == first-arrival order of their tiles (:473)
    const uint64_t km = __ballot(keep);
    const int idx = __popcll(km & ((1ull << lane) - 1ull));
    if (keep) slots[idx] = pack_bullet(nx, ny, dir, own);
    e.nb = __popcll(km);
    wave_sync();
    e.bw = lane < e.nb ? slots[lane] : 0u;
    return tile_diff;
}

// ------------------------------------------------------------------ MT19937 (CPython random.Random, core.py:41,69-70,86-87)
__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// Regenerate all 624 words (genrand_uint32's block), 64 words per pass.  Passes run in
// index order; inside a pass every lane reads before any lane writes, which reproduces the
// sequential recurrence exactly: word k needs OLD k, OLD k+1 (NEW word 0 for k = 623) and
// word k+397 (OLD for k < 227, else NEW word k-227, written >= 3 passes earlier).
__device__ __attribute__((noinline)) void mt_twist(uint32_t *mtg, uint32_t *scr, int lane)
{
    for (int i = lane; i < 624; i += kWave) scr[i] = mtg[i];
    wave_sync();
    for (int base = 0; base < 624; base += kWave) {
        const int kk = base + lane;
        uint32_t nv = 0u;
        if (kk < 624) {
            const uint32_t a = scr[kk];
            const uint32_t b = scr[kk + 1 == 624 ? 0 : kk + 1];
            const uint32_t c = scr[kk + 397 >= 624 ? kk - 227 : kk + 397];
            const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
            nv = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        wave_sync();
        if (kk < 624) scr[kk] = nv;
        wave_sync();
    }
    for (int i = lane; i < 624; i += kWave) mtg[i] = scr[i];
    wave_sync();
}

// InversusEnv.reset() (core.py:55-154) + SingleInversusRLEnv.reset() bookkeeping
// (env_wrappers.py:278-283).  scr: >= 624 LDS words.
template <class D>
__device__ inline void reset_env(const D &d, const State &s, int env, int lane, uint32_t *g, uint32_t *scr, Env &e)
{
    uint32_t *mtg = s.mt + (size_t)env * kMtStride;
    const uint32_t nx = (uint32_t)(d.W - 2), ny = (uint32_t)(d.H - 2);
    const int kx = 32 - __clz((int)nx), ky = 32 - __clz((int)ny);       // n.bit_length()
    int stage = 0, tries = 0, fin = 0;
    int p1x = 0, p1y = 0, p2x = 0, p2y = 0;
    int mti = e.mti;
    while (!fin) {
        if (mti >= 624) { mt_twist(mtg, scr, lane); mti = 0; }
        const int avail = min(kWave, 624 - mti);
        const uint32_t y = mt_temper(lane < avail ? mtg[mti + lane] : 0u);
        int used = 0;
        for (int i = 0; i < avail && !fin; ++i) {
            const uint32_t v = rdlane(y, i);
            used += 1;
            const bool isx = (stage & 1) == 0;
            const uint32_t r = v >> (32 - (isx ? kx : ky));                // getrandbits(k)
            if (r < (isx ? nx : ny)) {                                     // _randbelow rejection loop
                const int val = 1 + (int)r;                                // randint(1, n)
                if (stage == 0) { p1x = val; stage = 1; }
                else if (stage == 1) { p1y = val; stage = 2; }
                else if (stage == 2) { p2x = val; stage = 3; }
                else {
                    p2y = val;
                    tries += 1;
                    const int dist = abs(p2x - p1x) + abs(p2y - p1y);
                    if (dist > 4 || tries == 20) fin = 1;                  // :85-90, last try kept regardless
                    else stage = 2;
                }
            }
        }
        mti += used;
    }
    e.mti = mti;

    // grid: BLACK everywhere, WHITE plus at (1,1) (config.py:20-56), P2 plus WHITE, P1 plus WHITE,
    // then P2 plus BLACK (core.py:96-146)  ==  (plus(1,1) | plus(P1)) & ~plus(P2)
    wave_sync();
    for (int w = lane; w < d.GW; w += kWave) {
        uint32_t word = 0u, clr = 0u;
        const int cx[3] = { 1, p1x, p2x }, cy[3] = { 1, p1y, p2y };
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int x = cx[c] + (q == 1) - (q == 2), yy = cy[c] + (q == 3) - (q == 4);
                if ((unsigned)x < (unsigned)d.W && (unsigned)yy < (unsigned)d.H) {
                    const int pos = yy * d.W + x;
                    if ((pos >> 5) == w) {
                        if (c < 2) word |= 1u << (pos & 31); else clr |= 1u << (pos & 31);
                    }
                }
            }
        }
        g[w] = word & ~clr;
    }
    wave_sync();
    e.p[0] = { p1x, p1y, 6, 0, 1 };
    e.p[1] = { p2x, p2y, 6, 0, 1 };
    e.nb = 0;
    e.bw = 0u;
    e.step_count = 0;
    e.ep_ret = 0.0;
    e.prev1 = e.prev2 = 1;
}

// ------------------------------------------------------------------ Philox4x32-10 + dummy opponent (env_wrappers.py:69-170)
struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return { c0, c1, c2, c3 };
}

constexpr uint32_t kTagDummy = 0x44554d59u, kTagAction = 0x4143544eu, kKey1 = 0x1234abcdu;

// random.shuffle's loop on a 4-element list kept as packed nibbles (element i in bits 4i..4i+3):
// for i = 3..1: j = randbelow(i+1); swap(i, j), with randbelow(m) = mulhi(word, m).
__device__ __forceinline__ uint32_t set_nibble(uint32_t p, int k, uint32_t v) { return (p & ~(15u << (4 * k))) | (v << (4 * k)); }

__device__ __forceinline__ uint32_t fy_perm4(uint32_t w0, uint32_t w1, uint32_t w2)
{
    uint32_t perm = 0x3210u;
    const uint32_t ws[3] = { w0, w1, w2 };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = 3 - k;
        const int j = (int)__umulhi(ws[k], (uint32_t)(i + 1));
        const uint32_t vi = (perm >> (4 * i)) & 15u, vj = (perm >> (4 * j)) & 15u;
        perm = set_nibble(perm, i, vj);
        perm = set_nibble(perm, j, vi);
    }
    return perm;
}

__device__ __forceinline__ uint32_t apply_perm4(uint32_t list, uint32_t perm)
{
    uint32_t t = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) t |= ((list >> (4 * ((perm >> (4 * i)) & 15u))) & 15u) << (4 * i);
    return t;
}

// Draws of the device dummy: a counter hash, one 32-bit word per draw, generated only where a branch
// consumes it (the reference draws from the process-global `random`, env_wrappers.py:96-155, which
// cannot be reproduced per env; parity of the draws is statistical, parity of the decision function
// given the draws is exact, tests/golden/g6_dummy.npz).
//     key     = lowbias32(global_env * 0x9E3779B9 + seed)
//     word(j) = lowbias32(key + (ctr * 16 + j) * 0x85EBCA6B)          ctr = per-env step counter
// lowbias32 is the 2-multiply xorshift-multiply finaliser (bias 0.17); Philox4x32-10 cost ~100 scalar
// instructions per env-step here, a fifth of the whole step, for draws that feed three coin flips.
//   j = 0: u0 (shoot)   1: u1 (random move)   2: u2 (easy idle)   3: swap of the 2 candidates
//   j = 4..6: shuffle A words   j = 8..10: shuffle B words
// u = word * 2^-32 is only ever compared with a constant, so compares are done on the integer word:
//   u < 0.2   <=> word <= 858993459      u < 0.05 <=> word <= 214748364      u > 0.001 <=> word >= 4294968
// (0.2 * 2^32 = 858993459.2, 0.05 * 2^32 = 214748364.8, 0.001 * 2^32 = 4294967.296).
__device__ __forceinline__ uint32_t lowbias32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ uint32_t dummy_word(uint32_t key, uint32_t ctr, uint32_t j)
{
    return lowbias32(key + (ctr * 16u + j) * 0x85EBCA6Bu);
}

template <class D>
__device__ inline int dummy_action(const D &d, const uint32_t *g, int lane, const Env &e, int hard, uint32_t seed, uint32_t genv, uint32_t ctr)
{
    const Player &p1 = e.p[0], &p2 = e.p[1];
    if (!p2.alive) return 0;                                               // :77-78
    const uint32_t key = lowbias32(genv * 0x9E3779B9u + seed);
    if (!hard) {                                                           // easy: shoot_prob = random_move_prob = 0
        if (dummy_word(key, ctr, 2u) >= 4294968u) return 0;                // :122-124, u2 > move_prob
    } else {
        const bool x_al = p2.x == p1.x, y_al = p2.y == p1.y;
        if ((x_al || y_al) && p2.ammo > 0 && dummy_word(key, ctr, 0u) <= 858993459u) {   // :96-101, shoot_prob 0.2
            if (x_al) return 5 + (p1.y < p2.y ? 0 : 2);
            return 5 + (p1.x < p2.x ? 3 : 1);
        }
    }

    // walkable-for-P2 mask over directions (in bounds and tile != WHITE), :116-118 etc.: lane d < 4 tests
    // direction d and one ballot collects the mask (instead of four scalar bounds-check + LDS-read +
    // readfirstlane sequences)
    uint32_t walk;
    {
        const int dl = lane & 3;
        const int nx = p2.x + dir_dx(dl), ny = p2.y + dir_dy(dl);
        const bool inb = (unsigned)nx < (unsigned)d.W && (unsigned)ny < (unsigned)d.H;
        const bool ok = lane < 4 && inb && tile_bit(g, inb ? ny * d.W + nx : 0) == 0u;
        walk = (uint32_t)__ballot(ok) & 15u;
    }
    uint32_t dirs = 0u | (2u << 4) | (3u << 8) | (1u << 12);               // [UP, DOWN, LEFT, RIGHT], :104
    if (hard && dummy_word(key, ctr, 1u) <= 214748364u) {                  // :105-119, random_move_prob 0.05
        dirs = apply_perm4(dirs, fy_perm4(dummy_word(key, ctr, 4u), dummy_word(key, ctr, 5u), dummy_word(key, ctr, 6u)));
        const int c = (int)(dirs & 15u);
        if ((walk >> c) & 1u) return 1 + c;
    }
    const int dx = p1.x - p2.x, dy = p1.y - p2.y;                          // :127-138
    int c0 = -1, c1 = -1;
    if (dx != 0) c0 = dx > 0 ? 1 : 3;
    if (dy != 0) { const int c = dy > 0 ? 2 : 0; if (c0 < 0) c0 = c; else c1 = c; }
    if (c1 >= 0 && __umulhi(dummy_word(key, ctr, 3u), 2u) == 0u) { const int t = c0; c0 = c1; c1 = t; }
    if (c0 >= 0 && ((walk >> c0) & 1u)) return 1 + c0;                     // :140-152
    if (c1 >= 0 && ((walk >> c1) & 1u)) return 1 + c1;
    dirs = apply_perm4(dirs, fy_perm4(dummy_word(key, ctr, 8u), dummy_word(key, ctr, 9u), dummy_word(key, ctr, 10u)));   // :155-168
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (int)((dirs >> (4 * i)) & 15u);
        if ((walk >> c) & 1u) return 1 + c;
    }
    return 0;                                                              // :170
}

// Observation f32 stores go through the buffer-store builtin, whose cache-policy bits are an immediate: AUX = 0 plain (the line
// stays dirty in the XCD's L2 until it is evicted or the kernel ends), AUX = 16 `sc1` = write-through.  Write-through is what a
// 100+ MB stream wants -- nothing is left for the end-of-kernel L2 write-back (15x10, same box: 20.2 -> 18.5 us per vector step
// on a cache-resident buffer, 27.2 -> 25.1 us streaming into 32 buffers) -- but ONLY for stores that cover whole 128-B lines: a
// partially written line goes to memory as a masked write (a pure-store kernel over 7 200-B regions, 2 partial lines in 57:
// 20.9 us plain, 31.0 us write-through; tools/wstream2.hip, profiles/r03_store_policy.md).  So the kernels take it as a launch
// parameter (StepArgs::obs_wt) that the host sets when every store instruction of the launch is line-aligned.
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));

template <int AUX>
__device__ __forceinline__ void store_f4(float4 v, __amdgpu_buffer_rsrc_t rs, int byte_off)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, v), rs, byte_off, 0, AUX);
}
constexpr int kStorePlain = 0, kStoreWriteThrough = 16;

// ------------------------------------------------------------------ observation (env_wrappers.py:173-245)
// f32(k / 6) for k = 0..6, folded at compile time (IEEE f64 divide, then one rounding to f32) -- the values
// `player.ammo / MAX_AMMO` takes in env_wrappers.py:239-240.  Other ammo values exist only via inv_set_state.
__constant__ float kSixths[7] = { 0.0f, (float)(1.0 / 6.0), (float)(2.0 / 6.0), (float)(3.0 / 6.0),
                                  (float)(4.0 / 6.0), (float)(5.0 / 6.0), 1.0f };

// 32 bits of a WH-bit string starting at bit `start` (zero outside [0, WH)); the string is
// the grid (invert = false) or its complement (invert = true).
__device__ __forceinline__ uint32_t grid_window(const uint32_t *g, int GW, uint32_t lastmask, int start, bool invert)
{
    const int sidx = start >> 5, sh = start & 31;
    uint32_t lo = 0u, hi = 0u;
    if (sidx >= 0 && sidx < GW) { lo = invert ? ~g[sidx] : g[sidx]; if (sidx == GW - 1) lo &= lastmask; }
    if (sidx + 1 >= 0 && sidx + 1 < GW) { hi = invert ? ~g[sidx + 1] : g[sidx + 1]; if (sidx + 1 == GW - 1) hi &= lastmask; }
    return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
}

// The wave's nibble -> float4 table (16 entries x 16 B in LDS): entry n = (bit0, bit1, bit2, bit3) of n as 0.0f / 1.0f.
__device__ __forceinline__ void init_nibble_table(float *tab, int lane)
{
    tab[lane] = ((lane >> 2) >> (lane & 3)) & 1 ? 1.0f : 0.0f;
}

// Render player `pid`'s 12-channel observation of the current state.  The 12*W*H floats are
// 0/1, so the wave first assembles them as ONE packed bitstring in LDS (channels 0/1 by a
// funnel-shifted copy of the grid words, channels 2..11 by per-lane LDS OR of single bits),
// then every lane expands nibbles to float4 and streams them out: consecutive lanes write
// consecutive 16 B, 1 KiB per wave store instruction.
template <class D>
__device__ inline void render_obs(const D &d, const uint32_t *g, uint32_t *ob, const float4 *tab, int lane, int pid,
                                  const Env &e, float *__restrict__ obs_env, float *__restrict__ extra_env, bool as_bits = false, bool wt = false)
{
    const int WH = d.WH;
    const uint32_t lastmask = (WH & 31) ? ((1u << (WH & 31)) - 1u) : 0xffffffffu;
    for (int w = lane; w < d.OW; w += kWave) {
        const int lo = w * 32;
        uint32_t word = 0u;
        if (lo < 2 * WH) word = grid_window(g, d.GW, lastmask, lo, true) | grid_window(g, d.GW, lastmask, lo - WH, false);
        ob[w] = word;
    }
    wave_sync();
    // select by value: a runtime index into e.p[] would push the context to scratch
    const Player me = pid ? e.p[1] : e.p[0], en = pid ? e.p[0] : e.p[1];
    if (lane == 0 && me.alive && me.x < d.W && me.y < d.H) { const int b = 2 * WH + me.y * d.W + me.x; atomicOr(&ob[b >> 5], 1u << (b & 31)); }
    if (lane == 1 && en.alive && en.x < d.W && en.y < d.H) { const int b = 3 * WH + en.y * d.W + en.x; atomicOr(&ob[b >> 5], 1u << (b & 31)); }
    {
        const int bx = (int)(e.bw & 63u), by = (int)((e.bw >> 6) & 63u), dir = (int)((e.bw >> 12) & 3u), own = (int)((e.bw >> 14) & 1u);
        if (lane < e.nb && bx < d.W && by < d.H) {                          // :224-235
            const int b = ((own == pid ? 4 : 8) + dir) * WH + by * d.W + bx;
            atomicOr(&ob[b >> 5], 1u << (b & 31));
        }
    }
    wave_sync();
    // expand: float4 j of the observation = bits 4j..4j+3 of the bitstring, i.e. nibble (j & 7) of word
    // j >> 3, looked up in the wave's 16-entry float4 table (one conflict-free ds_read_b128, no VALU
    // selects); lanes 8q..8q+7 share a word and the nibble position of a lane never changes.
    if (as_bits) {
        // INV_STEP_OBS_BITS: the bitstring itself is the output (bit c*W*H + y*W + x <=> obs[c][y][x] == 1.0f);
        // obs_env then points at uint32_t[OW] (pad bits of the last word are zero)
        uint32_t *bdst = reinterpret_cast<uint32_t *>(obs_env);
        for (int w = lane; w < d.OW; w += kWave) bdst[w] = ob[w];
    } else {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(obs_env, 0, d.F4 * 16, 0x00020000);   // obs_env is wave-uniform
        const uint32_t sh = (uint32_t)(lane & 7) * 4u;
        if (wt) {
#pragma unroll 4
            for (int j = lane; j < d.F4; j += kWave) store_f4<kStoreWriteThrough>(tab[(ob[j >> 3] >> sh) & 15u], rs, j * 16);
        } else {
#pragma unroll 4
            for (int j = lane; j < d.F4; j += kWave) store_f4<kStorePlain>(tab[(ob[j >> 3] >> sh) & 15u], rs, j * 16);
        }
    }
    if (extra_env && lane < 4) {                                            // :238-243: f32(ammo / 6) computed in f64
        const Player &q = (lane & 1) ? en : me;
        float val;
        if (lane < 2) val = !q.alive ? 0.0f : (q.ammo <= 6 ? kSixths[q.ammo] : (float)((double)q.ammo / 6.0));
        else val = q.alive ? 1.0f : 0.0f;
        extra_env[lane] = val;
    }
    wave_sync();
}

}  // namespace inv
