// inv_step4.h -- the step kernel with FOUR ENVS PER WAVEFRONT (one env per 16-lane row).
//
// Why: the wave-per-env kernel (inv_device.h, k_step) is instruction-issue bound wherever the observation stream does not
// hide it -- ~680 wave instructions per env-step, most of them "uniform" work (players, ammo, reward, the dummy opponent) that
// occupies a whole wave instruction for one env.  Here a wave instruction serves four envs: everything that was wave-uniform
// becomes row-uniform (the same value in the 16 lanes of a row), branches become selects, lane l of a row owns bullet slot
// l (K <= 16), and the O(K) readlane loop of the bullet update is replaced by one LDS atomic per bullet on a per-env tile
// table.  Per env-step: ~150 wave instructions instead of ~680 (vs_dummy hard, f32 planes).
//
// Same state layout, same outputs, same semantics as k_step: the reference lines cited in inv_device.h apply unchanged
// (inversus/core.py:249-475, 497-531; inversus_rl/env_wrappers.py:69-170, 173-245, 318-442; training.py:148-151).  Resets are
// rare (one env in ~25..400 steps) and keep the wave-cooperative MT19937 code of inv_device.h: the wave handles its rows'
// resets one after the other with all 64 lanes.
//
// Used for full steps (INV_STEP_DEFAULT) without a mask on the two static board sizes; partial-phase and masked steps
// (the state-poking test surface) stay on k_step.  tests/ run every parity test through both kernels.
#pragma once

#include <utility>

#include "inv_device.h"

namespace inv4 {

using namespace inv;

constexpr int kRows = 4, kRowLanes = 16;

// LDS words of one wave: [nibble table 64][row 0][row 1][row 2][row 3][MT scratch 624]
// row: [head 8][grid GWp][slots 16][region R]   region = tile table (WH words, bullet update) / observation bitstring (OW words)
template <class D>
struct Lds4 {
    static constexpr int GWp = (D::GW + 3) & ~3;
    static constexpr int R = ((D::WH > D::OW ? D::WH : D::OW) + 3) & ~3;
    static constexpr int row_words = 8 + GWp + 16 + R;
    static constexpr int wave_words = kTabWords + kRows * row_words + 624;
};

struct Row {              // per lane; every field except bw is uniform within a 16-lane row
    int x1, y1, am1, rl1, al1;
    int x2, y2, am2, rl2, al2;
    int prev1, prev2, nb, step_count, mti;
    uint32_t opp_ctr;
    double ep_ret;
    uint32_t bw;          // lane l of the row: bullet l (canonical u16 packing), 0 beyond nb
};

// The four action bytes of a wave's envs (a[idx0 .. idx0 + 3], idx0 wave-uniform, `total` bytes in the buffer) through the
// SCALAR cache: the aligned dword that holds a[idx0] and, only when the four bytes straddle it AND the next dword still holds
// a byte of the buffer, that next dword.  Scalar loads are not ordered behind the wave's vector stores (vmcnt is in-order for
// loads and stores alike), so in the multi-step kernel the actions of step t + 1 do not wait for the observation stores of
// step t to drain.  Like sload_u8 this never reads beyond the 4-byte-rounded end of the buffer (an earlier version always
// read two dwords and faulted on an action buffer that ended exactly at the end of a mapped segment).
__device__ __forceinline__ uint32_t sload_u8x4(const uint8_t *p, size_t idx0, size_t total)
{
    const uintptr_t a = (uintptr_t)p + idx0, base = a & ~(uintptr_t)3;
    const uint32_t lo = *(const_u32_ptr)base;
    uint32_t hi = 0u;
    if ((a & 3u) != 0u && base + 4 < (uintptr_t)p + total) hi = *(const_u32_ptr)(base + 4);
    const uint64_t v = (uint64_t)lo | ((uint64_t)hi << 32);
    return (uint32_t)(v >> (8u * (uint32_t)(a & 3u)));
}

// this row's 16 bits of a wave ballot
__device__ __forceinline__ uint32_t row_bits(uint64_t m, int lane) { return (uint32_t)(m >> (lane & 48)) & 0xffffu; }

template <class D>
__device__ __forceinline__ void load_rows(const State &s, int env, int l, bool valid, uint32_t *rowp, Row &e)
{
    uint32_t *hd = rowp, *g = rowp + 8;
    const int envc = valid ? env : 0;                                       // rows past N read env 0 and are never stored
    const uint32_t hv = s.head[(size_t)envc * kHeadWords + (l & 7)];
    const uint32_t bwv = (uint32_t)s.bullets[(size_t)envc * D::K + (l < D::K ? l : D::K - 1)];
    if (l < 8) hd[l] = hv;
    for (int w = l; w < D::GW; w += kRowLanes) g[w] = s.grid[(size_t)envc * D::GW + w];
    wave_sync();
    const uint4 h0 = *reinterpret_cast<const uint4 *>(hd), h1 = *reinterpret_cast<const uint4 *>(hd + 4);   // row broadcast
    e.x1 = (int)(h0.x & 255u); e.y1 = (int)((h0.x >> 8) & 255u); e.am1 = (int)((h0.x >> 16) & 255u); e.rl1 = (int)(h0.x >> 24);
    e.x2 = (int)(h0.y & 255u); e.y2 = (int)((h0.y >> 8) & 255u); e.am2 = (int)((h0.y >> 16) & 255u); e.rl2 = (int)(h0.y >> 24);
    const uint32_t fl = h0.z;
    e.al1 = (int)(fl & 1u); e.al2 = (int)((fl >> 1) & 1u); e.prev1 = (int)((fl >> 2) & 1u); e.prev2 = (int)((fl >> 3) & 1u);
    e.nb = (int)((fl >> 8) & 255u);
    e.step_count = (int)h0.w;
    e.opp_ctr = h1.x;
    e.mti = (int)h1.y;
    e.ep_ret = __longlong_as_double((long long)((uint64_t)h1.z | ((uint64_t)h1.w << 32)));
    e.bw = l < e.nb ? bwv : 0u;
}

template <class D>
__device__ __forceinline__ void store_rows(const State &s, int env, int l, bool valid, const uint32_t *rowp, const Row &e)
{
    if (!valid) return;
    const uint32_t *g = rowp + 8;
    const uint64_t rb = (uint64_t)__double_as_longlong(e.ep_ret);
    const uint32_t fl = (uint32_t)e.al1 | ((uint32_t)e.al2 << 1) | ((uint32_t)e.prev1 << 2) | ((uint32_t)e.prev2 << 3) | ((uint32_t)e.nb << 8);
    const uint32_t p1 = (uint32_t)e.x1 | ((uint32_t)e.y1 << 8) | ((uint32_t)e.am1 << 16) | ((uint32_t)e.rl1 << 24);
    const uint32_t p2 = (uint32_t)e.x2 | ((uint32_t)e.y2 << 8) | ((uint32_t)e.am2 << 16) | ((uint32_t)e.rl2 << 24);
    const uint32_t hv = l == HD_P1 ? p1 : l == HD_P2 ? p2 : l == HD_FLAGS ? fl : l == HD_STEP ? (uint32_t)e.step_count
                      : l == HD_OPP ? e.opp_ctr : l == HD_MTI ? (uint32_t)e.mti : l == HD_RET_LO ? (uint32_t)rb : (uint32_t)(rb >> 32);
    if (l < kHeadWords) s.head[(size_t)env * kHeadWords + l] = hv;
    for (int w = l; w < D::GW; w += kRowLanes) s.grid[(size_t)env * D::GW + w] = g[w];
    if (l < D::K) s.bullets[(size_t)env * D::K + l] = (uint16_t)(l < e.nb ? e.bw : 0u);
}

// ------------------------------------------------------------------ actions (core.py:249-381, 510-525), branch-free
template <class D>
__device__ __forceinline__ void emit4(int l, Row &e, uint32_t word, bool cond, uint32_t &err)
{
    const bool ovf = cond && e.nb >= D::K;
    const bool go = cond && !ovf;
    if (go && l == e.nb) e.bw = word;
    e.nb += go ? 1 : 0;
    err |= ovf ? ERR_BULLET_OVERFLOW : 0u;
}

template <class D, int PID>
__device__ __forceinline__ void apply_action4(const uint32_t *g, int l, int a, bool ok, Row &e, uint32_t &err)
{
    int &x = PID ? e.x2 : e.x1, &y = PID ? e.y2 : e.y1, &ammo = PID ? e.am2 : e.am1;
    const int alive = PID ? e.al2 : e.al1;
    const bool act = ok && alive && a != 0;
    const int type = (a + 3) >> 2, dir = (a - 1) & 3;                       // env_wrappers.py:24-37
    const int dx = dir_dx(dir), dy = dir_dy(dir);
    const int nx = x + dx, ny = y + dy;
    const bool inb = (unsigned)nx < (unsigned)D::W && (unsigned)ny < (unsigned)D::H;
    const uint32_t tb = tile_bit(g, inb ? ny * D::W + nx : 0);
    const bool mv = act && type == 1 && inb && tb != (uint32_t)PID;         // walkable <=> tile != own colour (core.py:238-247)
    const bool sh = act && type == 2 && ammo > 0;                           // core.py:298-326
    const bool ch = act && type == 3 && ammo >= 3;                          // core.py:328-381
    const int sx = x, sy = y;                                               // shots leave from the pre-move tile (a player moves OR shoots)
    x = mv ? nx : x;
    y = mv ? ny : y;
    ammo -= sh ? 1 : (ch ? 3 : 0);
    const bool vertical = (dir & 1) == 0;
    const int ax = vertical ? sx - 1 : sx, ay = vertical ? sy : sy - 1;
    const int bx = vertical ? sx + 1 : sx, by = vertical ? sy : sy + 1;
    emit4<D>(l, e, pack_bullet(sx, sy, dir, PID), sh || ch, err);
    emit4<D>(l, e, pack_bullet(ax & 63, ay & 63, dir, PID), ch && ax >= 0 && ay >= 0, err);
    emit4<D>(l, e, pack_bullet(bx & 63, by & 63, dir, PID), ch && bx < D::W && by < D::H, err);
}

__device__ __forceinline__ void reload4(int alive, int &ammo, int &rl, bool ok)          // core.py:383-397
{
    const bool inc = ok && alive && ammo < 6;
    rl += inc ? 1 : 0;
    const bool up = inc && rl >= 30;
    ammo += up ? 1 : 0;
    rl = up ? 0 : rl;
}

// ------------------------------------------------------------------ bullets (core.py:399-475)
// tt: this row's tile table (WH words).  Every live bullet ORs bit (l + 16 * owner) into the word of its target tile: the low
// half lists P1's arrivals, the high half P2's -- both non-empty = mixed owners, everything on the tile vanishes (:444-449);
// an earlier lane in the own half = a same-owner bullet ahead in list order, which survives instead (:453).
template <class D>
__device__ __forceinline__ int update_bullets4(uint32_t *g, uint32_t *slots, uint32_t *tt, int lane, int l, bool ok, Row &e)
{
    if (__ballot(ok && e.nb > 0) == 0ull) return 0;                         // wave-uniform shortcut
    const int bx = (int)(e.bw & 63u), by = (int)((e.bw >> 6) & 63u), dir = (int)((e.bw >> 12) & 3u), own = (int)((e.bw >> 14) & 1u);
    const int nx = bx + dir_dx(dir), ny = by + dir_dy(dir);
    const bool live = ok && l < e.nb && (unsigned)nx < (unsigned)D::W && (unsigned)ny < (unsigned)D::H;   // :428-431
    const int pos = live ? ny * D::W + nx : 0;
    if (live) tt[pos] = 0u;
    wave_sync();
    if (live) atomicOr(&tt[pos], 1u << (l + 16 * own));
    wave_sync();
    const uint32_t v = live ? tt[pos] : 0u;
    const bool cancel = (v & 0xffffu) != 0u && (v >> 16) != 0u;
    const uint32_t mine = own ? (v >> 16) : (v & 0xffffu);
    const bool dup = (mine & ((1u << l) - 1u)) != 0u;
    const bool keep = live && !cancel && !dup;

    // flip only tiles of the owner's colour (:458-461): P1 turns BLACK->WHITE, P2 WHITE->BLACK.  A tile carries at most one
    // surviving bullet, so no other lane reads or flips the same bit.
    const uint32_t tb = tile_bit(g, pos);
    const bool f1 = keep && own == 0 && tb == 0u;
    const bool f2 = keep && own == 1 && tb == 1u;
    wave_sync();
    if (f1) atomicOr(&g[pos >> 5], 1u << (pos & 31));
    if (f2) atomicAnd(&g[pos >> 5], ~(1u << (pos & 31)));
    const int tile_diff = __popc(row_bits(__ballot(f1), lane)) - __popc(row_bits(__ballot(f2), lane));

    // kills (:464-470): a surviving bullet on an alive non-owner's tile
    const int p1pos = e.y1 * D::W + e.x1, p2pos = e.y2 * D::W + e.x2;
    const bool hit1 = row_bits(__ballot(keep && own == 1 && pos == p1pos), lane) != 0u;
    const bool hit2 = row_bits(__ballot(keep && own == 0 && pos == p2pos), lane) != 0u;
    e.al1 = (ok && hit1) ? 0 : e.al1;
    e.al2 = (ok && hit2) ? 0 : e.al2;

    // stable compaction: survivors keep list order == first-arrival order of their tiles (:473)
    const uint32_t km = row_bits(__ballot(keep), lane);
    const int idx = __popc(km & ((1u << l) - 1u));
    if (keep) slots[idx] = pack_bullet(nx, ny, dir, own);
    const int nb_new = __popc(km);
    wave_sync();
    const uint32_t nbw = slots[l];
    e.nb = ok ? nb_new : e.nb;
    e.bw = ok ? (l < nb_new ? nbw : 0u) : e.bw;
    wave_sync();
    return ok ? tile_diff : 0;
}

// ------------------------------------------------------------------ dummy opponent (env_wrappers.py:69-170), branch-free
// Same draws and the same decision function as inv_device.h::dummy_action (word j of the per-env counter hash feeds the same
// decision); all words a decision could consume are generated, the result is selected.
template <class D>
__device__ __forceinline__ int dummy_action4(const uint32_t *g, int lane, int l, const Row &e, int hard, uint32_t seed, uint32_t genv)
{
    const uint32_t key = lowbias32(genv * 0x9E3779B9u + seed), ctr = e.opp_ctr;
    const bool easy_idle = !hard && dummy_word(key, ctr, 2u) >= 4294968u;   // :122-124
    const bool dead = !e.al2;
    // most steps of the easy dummy end here for all four envs of the wave
    if (__ballot(!(dead || easy_idle)) == 0ull) return 0;
    const bool x_al = e.x2 == e.x1, y_al = e.y2 == e.y1;
    const bool shoot = hard && (x_al || y_al) && e.am2 > 0 && dummy_word(key, ctr, 0u) <= 858993459u;   // :96-101
    const int shoot_a = x_al ? 5 + (e.y1 < e.y2 ? 0 : 2) : 5 + (e.x1 < e.x2 ? 3 : 1);
    // walkable-for-P2 mask over the four directions: lane l & 3 of the row tests direction l & 3
    uint32_t walk;
    {
        const int dl = l & 3;
        const int nx = e.x2 + dir_dx(dl), ny = e.y2 + dir_dy(dl);
        const bool inb = (unsigned)nx < (unsigned)D::W && (unsigned)ny < (unsigned)D::H;
        const bool okw = inb && tile_bit(g, inb ? ny * D::W + nx : 0) == 0u;
        walk = row_bits(__ballot(okw), lane) & 15u;
    }
    uint32_t dirs = 0u | (2u << 4) | (3u << 8) | (1u << 12);               // [UP, DOWN, LEFT, RIGHT], :104
    const bool rnd = hard && dummy_word(key, ctr, 1u) <= 214748364u;        // :105-119
    const uint32_t dirs_a = apply_perm4(dirs, fy_perm4(dummy_word(key, ctr, 4u), dummy_word(key, ctr, 5u), dummy_word(key, ctr, 6u)));
    dirs = rnd ? dirs_a : dirs;
    const int ca = (int)(dirs_a & 15u);
    const bool rnd_go = rnd && ((walk >> ca) & 1u);
    const int dx = e.x1 - e.x2, dy = e.y1 - e.y2;                           // :127-138
    int c0 = -1, c1 = -1;
    if (dx != 0) c0 = dx > 0 ? 1 : 3;
    {
        const int c = dy > 0 ? 2 : 0;
        const bool hasy = dy != 0;
        c1 = (hasy && c0 >= 0) ? c : -1;
        c0 = (hasy && c0 < 0) ? c : c0;
    }
    const bool swap = c1 >= 0 && __umulhi(dummy_word(key, ctr, 3u), 2u) == 0u;
    const int d0 = swap ? c1 : c0, d1 = swap ? c0 : c1;
    const bool go0 = d0 >= 0 && ((walk >> (d0 & 3)) & 1u), go1 = d1 >= 0 && ((walk >> (d1 & 3)) & 1u);
    const uint32_t dirs_b = apply_perm4(dirs, fy_perm4(dummy_word(key, ctr, 8u), dummy_word(key, ctr, 9u), dummy_word(key, ctr, 10u)));   // :155-168
    int fb = 0;
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        const int c = (int)((dirs_b >> (4 * i)) & 15u);
        fb = ((walk >> c) & 1u) ? 1 + c : fb;
    }
    int a = fb;
    a = go1 ? 1 + d1 : a;
    a = go0 ? 1 + d0 : a;
    a = rnd_go ? 1 + ca : a;
    a = shoot ? shoot_a : a;
    a = (dead || easy_idle) ? 0 : a;
    return a;
}

// ------------------------------------------------------------------ observation (env_wrappers.py:173-245)
// One store instruction (64 lanes x 16 B) of a wave's GROUP stream: the kRows regions of a wave's envs are adjacent in the
// observation buffer (F4 float4s each), so the wave writes them as one run of kRows * F4 float4s -- at 15x10 that is
// 28 800 B = 225 whole 128-B lines starting on a line (4 regions of 7 200 B; one region alone starts and ends inside a line),
// 28 full store instructions + one of 8 lanes instead of 4 x (7 + a 2-lane tail).  float4 j of the group = float4
// q = j - rr * F4 of row rr = nibble (q & 7) of word q >> 3 of that row's bitstring.  I is a compile-time instruction index,
// so rr, the LDS word offset and the store offset fold into immediates; wq / sh hold the per-lane parts of (q >> 3) and
// 4 * (q & 7), which depend on rr only through (rr * F4) mod 8.
template <class D, int AUX, int I>
__device__ __forceinline__ void group_store(__amdgpu_buffer_rsrc_t rs, const uint32_t *ob0, const float4 *tab, int lane, int row_words,
                                            const int (&wq)[kRows], const uint32_t (&sh)[kRows])
{
    constexpr int F4 = D::F4, total = kRows * F4, lo = 64 * I, hi = 64 * I + 63;
    if constexpr (lo < total) {
        constexpr int hic = hi < total ? hi : total - 1;
        constexpr int ra = lo / F4, rb = hic / F4;                           // F4 >= 64: a 64-lane window meets at most one boundary
        // q = lane + c with c = lo - rr * F4 = 8 * cd + cm (0 <= cm < 8): q >> 3 = ((lane + cm) >> 3) + cd
        constexpr int ca = lo - ra * F4, cda = (ca - (((ca % 8) + 8) % 8)) / 8;
        constexpr int cb = lo - rb * F4, cdb = (cb - (((cb % 8) + 8) % 8)) / 8;
        const int j = lane + lo;
        uint32_t word, shift;
        if constexpr (ra == rb) {
            word = ob0[ra * row_words + cda + wq[ra]];
            shift = sh[ra];
        } else {
            const bool up = j >= rb * F4;
            word = ob0[up ? rb * row_words + cdb + wq[rb] : ra * row_words + cda + wq[ra]];
            shift = up ? sh[rb] : sh[ra];
        }
        const float4 v = tab[(word >> shift) & 15u];
        if (hi < total || j < total) store_f4<AUX>(v, rs, j * 16);
    }
}

template <class D, int AUX, int... Is>
__device__ __forceinline__ void group_store_all(__amdgpu_buffer_rsrc_t rs, const uint32_t *ob0, const float4 *tab, int lane, int row_words,
                                                const int (&wq)[kRows], const uint32_t (&sh)[kRows], std::integer_sequence<int, Is...>)
{
    (group_store<D, AUX, Is>(rs, ob0, tab, lane, row_words, wq, sh), ...);
}

// obs_wave: the region of the wave's FIRST env (wave-uniform pointer); row r's region follows r * obs_stride elements later.
template <class D>
__device__ __forceinline__ void render4(const uint32_t *g, uint32_t *ob, const float4 *tab, int lane, int pid, const Row &e, bool valid,
                                        int nvalid, int row_words, float *__restrict__ obs_wave, float *__restrict__ extra_env, bool as_bits, bool wt)
{
    const int l = lane & (kRowLanes - 1), r = lane >> 4;
    constexpr int WH = D::WH;
    constexpr uint32_t lastmask = (WH & 31) ? ((1u << (WH & 31)) - 1u) : 0xffffffffu;
    for (int w = l; w < D::OW; w += kRowLanes) {
        const int lo = w * 32;
        uint32_t word = 0u;
        if (lo < 2 * WH) word = grid_window(g, D::GW, lastmask, lo, true) | grid_window(g, D::GW, lastmask, lo - WH, false);
        ob[w] = word;
    }
    wave_sync();
    const int mx = pid ? e.x2 : e.x1, my = pid ? e.y2 : e.y1, mal = pid ? e.al2 : e.al1, mam = pid ? e.am2 : e.am1;
    const int ex = pid ? e.x1 : e.x2, ey = pid ? e.y1 : e.y2, eal = pid ? e.al1 : e.al2, eam = pid ? e.am1 : e.am2;
    if (l == 0 && mal && mx < D::W && my < D::H) { const int b = 2 * WH + my * D::W + mx; atomicOr(&ob[b >> 5], 1u << (b & 31)); }
    if (l == 1 && eal && ex < D::W && ey < D::H) { const int b = 3 * WH + ey * D::W + ex; atomicOr(&ob[b >> 5], 1u << (b & 31)); }
    {
        const int bx = (int)(e.bw & 63u), by = (int)((e.bw >> 6) & 63u), dir = (int)((e.bw >> 12) & 3u), own = (int)((e.bw >> 14) & 1u);
        if (l < e.nb && bx < D::W && by < D::H) {                           // :224-235
            const int b = ((own == pid ? 4 : 8) + dir) * WH + by * D::W + bx;
            atomicOr(&ob[b >> 5], 1u << (b & 31));
        }
    }
    wave_sync();
    const uint32_t *ob0 = ob - (ptrdiff_t)r * row_words;                    // row 0's bitstring (this lane's region is row r's)
    if (as_bits) {
        if (valid) {
            uint32_t *bdst = reinterpret_cast<uint32_t *>(obs_wave) + (size_t)r * D::OW;
            for (int w = l; w < D::OW; w += kRowLanes) bdst[w] = ob[w];
        }
    } else if ((D::F4 % kWave) != 0 && nvalid == kRows) {
        // f32 planes, regions that are not whole store instructions (15x10: 450 float4s): one group stream, see group_store
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(obs_wave, 0, kRows * D::F4 * 16, 0x00020000);
        int wq[kRows];
        uint32_t sh[kRows];
#pragma unroll
        for (int rr = 0; rr < kRows; ++rr) {
            const int cm = (8 - (rr * D::F4) % 8) % 8;                       // (-rr * F4) mod 8
            wq[rr] = (lane + cm) >> 3;
            sh[rr] = (uint32_t)((lane + cm) & 7) * 4u;
        }
        constexpr auto seq = std::make_integer_sequence<int, (kRows * D::F4 + kWave - 1) / kWave>{};
        if (wt) group_store_all<D, kStoreWriteThrough>(rs, ob0, tab, lane, row_words, wq, sh, seq);
        else group_store_all<D, kStorePlain>(rs, ob0, tab, lane, row_words, wq, sh, seq);
    } else {
        // f32 planes: the WHOLE wave expands one row's bitstring after the other, so that every store instruction writes
        // 1 KiB of consecutive addresses (64 lanes x 16 B) like the wave-per-env kernel -- with each row expanding its own
        // env (4 x 256-B pieces per instruction) the same stream ran 10 % slower once it no longer fit the Infinity Cache.
        // float4 j of an env = nibble (j & 7) of word j >> 3; j = lane (mod 64) keeps the nibble position per lane.
        const uint32_t shl = (uint32_t)(lane & 7) * 4u;
#pragma unroll 1
        for (int rr = 0; rr < kRows; ++rr) {
            if (rr >= nvalid) break;                                        // wave-uniform
            const uint32_t *obr = ob0 + (ptrdiff_t)rr * row_words;
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(obs_wave + (size_t)rr * 12 * WH, 0, D::F4 * 16, 0x00020000);
            if (wt) {
#pragma unroll 4
                for (int j = lane; j < D::F4; j += kWave) store_f4<kStoreWriteThrough>(tab[(obr[j >> 3] >> shl) & 15u], rs, j * 16);
            } else {
#pragma unroll 4
                for (int j = lane; j < D::F4; j += kWave) store_f4<kStorePlain>(tab[(obr[j >> 3] >> shl) & 15u], rs, j * 16);
            }
        }
    }
    if (valid && extra_env && l < 4) {                                      // :238-243: f32(ammo / 6) computed in f64
        const int qa = (l & 1) ? eal : mal, qm = (l & 1) ? eam : mam;
        float val;
        if (l < 2) val = !qa ? 0.0f : (qm <= 6 ? kSixths[qm] : (float)((double)qm / 6.0));
        else val = qa ? 1.0f : 0.0f;
        extra_env[l] = val;
    }
    wave_sync();
}

}  // namespace inv4
