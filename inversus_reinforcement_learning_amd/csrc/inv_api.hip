// inv_api.hip -- kernels and the C ABI (include/inversus_hip.h) of libinversus_hip.so.
//
// gfx950 only.  One wavefront per env (see inv_device.h); 4 envs per 256-thread block, no
// workgroup barriers.  Every entry point is asynchronous on the caller's stream.
#if defined(INV_STEP_STAMPS) && !defined(INV_TOOLS_BUILD)
#error "INV_STEP_STAMPS is a diagnostic build (in-kernel stamps): tools/build_variant.py only (it defines INV_TOOLS_BUILD)"
#endif
#include "inv_device.h"
#include "inv_step4.h"
#include "inv_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

using namespace inv;

// =================================================================== kernels

// Diagnostic build only (-DINV_STEP_STAMPS, never the shipped library): per-wave phase cycles of the LAST launch,
// one 8-word record per env (7 phases + a "this wave reset" flag), written once per wave to a buffer of its own.
#ifdef INV_STEP_STAMPS
constexpr int kStampEnvs = 16384;
__device__ unsigned long long g_step_stamps[kStampEnvs * 8];
__device__ __forceinline__ unsigned long long step_stamp_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define SSTAMP(slot) do { const unsigned long long n_ = step_stamp_now(); st_acc[slot] += n_ - st_prev; st_prev = n_; } while (0)
// k_step4: ABSOLUTE times of the chip-wide 100 MHz clock, one record per wave (slot = wave index), so that the waves' phases can
// be laid on one time axis (tools/step4_timeline.py)
__device__ __forceinline__ unsigned long long wall_stamp_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define WSTAMP(slot) do { wst[slot] = wall_stamp_now(); } while (0)
#else
#define SSTAMP(slot) do { } while (0)
#define WSTAMP(slot) do { } while (0)
#endif

// LDS slice of a wave: [nibble table 64][grid GW][bullet slots K][scratch >= 624]
template <class D>
struct WaveLds {
    uint32_t *tabw, *g, *slots, *scr;
    __device__ WaveLds(const D &d, uint32_t *lds, int wv)
    {
        tabw = lds + wv * d.scr_words;
        g = tabw + kTabWords;
        slots = g + d.GW;
        scr = slots + d.K;
    }
};

// MultiEnvRunner.step for every env of the shard, T times (T = 1: inv_step; T > 1: inv_rollout).
template <class D, bool SINGLE>
__global__ __launch_bounds__(kWave * kWavesPerBlock) void k_step(StepArgs a)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (kWave - 1), wv = rfl((int)(threadIdx.x >> 6));   // wave index: uniform -> SGPR
    const int env = xcd_block((int)blockIdx.x, (int)gridDim.x) * kWavesPerBlock + wv;
    const int N = a.d.N;
    if (env >= N) return;                           // wave-uniform; no block barriers below
    if (a.mask && sload_u8(a.mask, (size_t)env) == 0u) return;
    const D d(a.d);
    const WaveLds<D> L(d, lds, wv);
    uint32_t *g = L.g;
    const float4 *tab = reinterpret_cast<const float4 *>(L.tabw);

    // issue the action loads of the first step together with the state loads
    uint32_t act1v = sload_u8(a.a1, (size_t)env), act2v = a.a2 ? sload_u8(a.a2, (size_t)env) : 0u;
    init_nibble_table(reinterpret_cast<float *>(L.tabw), lane);
#ifdef INV_STEP_STAMPS
    unsigned long long st_acc[7] = { 0, 0, 0, 0, 0, 0, 0 }, st_prev = step_stamp_now();
    int st_reset = 0;
#endif
    Env e;
    load_env(d, a.s, env, lane, g, e);
    SSTAMP(0);                                      // launch -> state loaded
    uint32_t err = 0u;
    const bool obits = (a.flags & INV_STEP_OBS_BITS) != 0u;       // obs / obs_p2 receive the packed bitstring
    const size_t obs_stride = obits ? (size_t)d.OW : (size_t)12 * d.WH;   // in 4-byte elements either way
    const bool full = (a.flags & INV_STEP_DEFAULT) == INV_STEP_DEFAULT;

    const int T = SINGLE ? 1 : a.T;
    for (int t = 0; t < T; ++t) {
        const size_t idx = (size_t)t * N + env;
        // Output pointers: in the multi-step (rollout) variant they are re-read from the kernarg segment
        // every step (10 scalar loads, K$-resident) instead of living in 20 SGPRs across the loop, which
        // otherwise spills the wave's scalar state to VGPR lanes (348 v_readlane/v_writelane).
        inv_step_out o;
        typedef const volatile StepArgs __attribute__((address_space(4))) *KernargPtr;
        const KernargPtr ka = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
        if (SINGLE) {
            o = a.o;
        } else {
            o.obs = ka->o.obs; o.extra = ka->o.extra; o.obs_p2 = ka->o.obs_p2; o.extra_p2 = ka->o.extra_p2;
            o.reward = ka->o.reward; o.reward64 = ka->o.reward64; o.done = ka->o.done; o.info_flags = ka->o.info_flags;
            o.episode_steps = ka->o.episode_steps; o.episode_return = ka->o.episode_return;
        }
        if (t > 0) { act1v = sload_u8(a.a1, idx); act2v = a.a2 ? sload_u8(a.a2, idx) : 0u; }
        const int act1 = rfl((int)act1v);
        int act2;
        if (a.a2) {
            act2 = rfl((int)act2v);
        } else {
            act2 = rfl(dummy_action(d, g, lane, e, a.hard, a.opp_seed, a.env_offset + (uint32_t)env, e.opp_ctr));
            e.opp_ctr += 1u;
        }
        SSTAMP(1);                                  // actions / dummy opponent
        double reward = 0.0;
        int done = 0;
        uint32_t info = 0u;
        const bool bad = act1 > 12 || act2 > 12;    // ValueError in the reference (env_wrappers.py:66): env left untouched
        if (bad) err |= ERR_BAD_ACTION;

        if (!bad) {
            if (full || (a.flags & INV_PHASE_ACTIONS)) {      // core.py:510-525: P1 then P2
                apply_action(d, g, lane, 0, act1, e, err);
                apply_action(d, g, lane, 1, act2, e, err);
            }
            if (full || (a.flags & INV_PHASE_RELOAD)) { reload(e.p[0]); reload(e.p[1]); }            // :528
            int tile_diff = 0;
            if (full || (a.flags & INV_PHASE_BULLETS)) tile_diff = update_bullets(d, g, L.slots, lane, e);   // :531

            if (full || (a.flags & INV_PHASE_WRAPPER)) {      // env_wrappers.py:333-442, f64 adds in the reference's order
                // Branch-free: every term the reference skips is added as +0.0 instead (exact: the running sum is
                // never -0.0), so the sequence of roundings is the reference's; selects instead of a dozen uniform
                // branches (A/B on one box: neutral for speed, kept for the straight-line code).
                e.step_count += 1;
                const Player &p1 = e.p[0], &p2 = e.p[1];
                const int al1 = p1.alive, al2 = p2.alive, both = al1 & al2, over = both ^ 1;
                const int landed = e.prev2 & (al2 ^ 1), got = e.prev1 & (al1 ^ 1);
                const int win = al1 & (al2 ^ 1), lose = al2 & (al1 ^ 1);
                const int xeq = p1.x == p2.x, yeq = p1.y == p2.y, aligned = xeq | yeq;
                const int dist = abs(p1.x - p2.x) + abs(p1.y - p2.y);
                const int shot = (act1 - 1) & 3;
                const int aim_v = ((p1.y < p2.y) & (shot == 2)) | ((p1.y > p2.y) & (shot == 0));
                const int aim_h = ((p1.x < p2.x) & (shot == 1)) | ((p1.x > p2.x) & (shot == 3));
                const int aiming = both & aligned & (act1 >= 5) & (p1.ammo > 0) & (xeq ? aim_v : aim_h);
                const int timeout = e.step_count >= a.d.max_steps;
                const double prox = sload_f64(SINGLE ? a.s.prox : ka->s.prox, dist);        // dist <= W + H - 2: always in the table
                reward += tile_diff > 0 ? (double)tile_diff * 0.01 : 0.0;                    // :349-354
                reward += landed ? 1.0 : 0.0;                                                // :357-362
                reward += got ? -0.01 : 0.0;                                                 // :364-369
                reward += (al1 & (p1.ammo == 0)) ? -0.001 : 0.0;                             // :372-373
                reward += both ? prox : 0.0;                                                 // :378-382
                reward += (both & aligned) ? 0.002 : 0.0;                                    // :385-387
                reward += aiming ? 0.05 : 0.0;                                               // :391-405
                reward += over ? (win ? 10.0 : (lose ? -0.1 : 0.0)) : -0.001;                // :408-427
                reward += (timeout & both) ? -2.0 : 0.0;                                     // :434-438
                done = over | timeout;
                info = (uint32_t)(landed * INV_INFO_LANDED_HIT | got * INV_INFO_GOT_HIT | (over & win) * INV_INFO_WIN | (over & lose) * INV_INFO_LOSE);
                e.prev1 = al1;
                e.prev2 = al2;
                e.ep_ret += reward;
            }
        }

        SSTAMP(2);                                  // tick + reward
        if (lane == 0) {
            if (o.reward) o.reward[idx] = (float)reward;
            if (o.reward64) o.reward64[idx] = reward;
            if (o.done) o.done[idx] = (uint8_t)done;
            if (o.info_flags) o.info_flags[idx] = (uint8_t)info;
            if (o.episode_steps) o.episode_steps[idx] = e.step_count;
            if (o.episode_return) o.episode_return[idx] = e.ep_ret;
        }

        SSTAMP(3);                                  // scalar outputs
        if (done && (a.flags & INV_STEP_AUTO_RESET)) {                                            // training.py:148-151
            State rs = a.s;
            if (!SINGLE) rs.mt = ka->s.mt;
            reset_env(d, rs, env, lane, g, L.scr, e);
#ifdef INV_STEP_STAMPS
            st_reset = 1;
#endif
        }
        SSTAMP(4);                                  // reset

        if (o.obs) render_obs(d, g, L.scr, tab, lane, 0, e, o.obs + idx * obs_stride, o.extra ? o.extra + idx * 4 : nullptr, obits, a.obs_wt != 0);
        if (o.obs_p2) render_obs(d, g, L.scr, tab, lane, 1, e, o.obs_p2 + idx * obs_stride, o.extra_p2 ? o.extra_p2 + idx * 4 : nullptr, obits, a.obs_wt != 0);
    }

    SSTAMP(5);                                      // observation render(s)
    if (SINGLE) {
        store_env(d, a.s, env, lane, g, e);
        if (err && lane == 0) atomicOr(a.s.err, err);
#ifdef INV_STEP_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SSTAMP(6);                                  // state store, all stores drained
        if (lane == 0 && env < kStampEnvs) {
            for (int q = 0; q < 7; ++q) g_step_stamps[env * 8 + q] = st_acc[q];
            g_step_stamps[env * 8 + 7] = (unsigned long long)st_reset;
        }
#endif
    } else {
        typedef const volatile StepArgs __attribute__((address_space(4))) *KernargPtr;
        const KernargPtr ka = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
        State fs;
        fs.head = ka->s.head; fs.grid = ka->s.grid; fs.bullets = ka->s.bullets; fs.mt = nullptr; fs.err = ka->s.err; fs.prox = nullptr;
        store_env(d, fs, env, lane, g, e);
        if (err && lane == 0) atomicOr(fs.err, err);
    }
}


// The same step with FOUR envs per wave (inv_step4.h): full steps only (all four phases), no mask.  WPB waves per block.
// The multi-step (inv_rollout) instantiations are asked for 4 waves per SIMD (<= 128 VGPRs; 15x10: 158 -> 128, no scratch): the
// launch's 4 096 waves are then all resident (16 per CU) instead of 3 072 + a second round of 1 024 -- the packed-observation
// rollout (compute-bound) 8.56 -> 7.07 us per step, the f32 rollouts (store-bound) unchanged (same-box A/B, round 3).
#ifndef INV_ROLLOUT_MIN_WAVES
#define INV_ROLLOUT_MIN_WAVES 4
#endif
// (the 32 x 32 instantiation needs ~140 registers: held to 128 by that bound it spilled 48 B/lane; it gets three waves per SIMD)
#ifndef INV_ROLLOUT_MIN_WAVES_32
#define INV_ROLLOUT_MIN_WAVES_32 3
#endif
template <class D, bool SINGLE, int WPB>
__global__ __launch_bounds__(kWave * WPB, SINGLE ? 1 : (D::W > 15 ? INV_ROLLOUT_MIN_WAVES_32 : INV_ROLLOUT_MIN_WAVES)) void k_step4(StepArgs a)
{
    using namespace inv4;
    using LL = Lds4<D>;
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (kWave - 1), wv = rfl((int)(threadIdx.x >> 6));
    const int l = lane & (kRowLanes - 1), r = lane >> 4;
    const int N = a.d.N;
    const int env0 = (xcd_block((int)blockIdx.x, (int)gridDim.x) * WPB + wv) * kRows;   // first env of this wave (uniform)
    if (env0 >= N) return;
#ifdef INV_STEP_STAMPS
    unsigned long long wst[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
    WSTAMP(0);                                                              // wave started
    const int env = env0 + r;
    const bool valid = env < N;
    const int envc = valid ? env : 0;
    uint32_t *wl = lds + wv * LL::wave_words;
    uint32_t *rowp = wl + kTabWords + r * LL::row_words;
    uint32_t *g = rowp + 8, *slots = g + LL::GWp, *reg = slots + 16, *mtscr = wl + kTabWords + kRows * LL::row_words;
    const float4 *tab = reinterpret_cast<const float4 *>(wl);
    init_nibble_table(reinterpret_cast<float *>(wl), lane);
    const D d(a.d);
    // the proximity table (env_wrappers.py:382) lives in the wave's MT scratch between resets: a per-row LDS read instead of a
    // vector load that would queue behind the previous step's observation stores in the multi-step kernel
    double *proxl = reinterpret_cast<double *>(mtscr);
    for (int q = lane; q <= D::W + D::H; q += kWave) proxl[q] = a.s.prox[q];

    Row e;
    load_rows<D>(a.s, env, l, valid, rowp, e);
    WSTAMP(1);                                                              // state in registers / LDS
    uint32_t err = 0u;
    const bool obits = (a.flags & INV_STEP_OBS_BITS) != 0u;
    const size_t obs_stride = obits ? (size_t)D::OW : (size_t)12 * D::WH;
    const bool auto_reset = (a.flags & INV_STEP_AUTO_RESET) != 0u;
    const uint32_t genv = a.env_offset + (uint32_t)envc;

    const int T = SINGLE ? 1 : a.T;
    for (int t = 0; t < T; ++t) {
        const size_t idx = (size_t)t * N + envc;
        const size_t idx0 = (size_t)t * N + env0;                            // wave-uniform: the four envs' action bytes are adjacent
        const int act1 = (int)((sload_u8x4(a.a1, idx0, (size_t)T * N) >> (8 * r)) & 255u);
        int act2;
        if (a.a2) {
            act2 = (int)((sload_u8x4(a.a2, idx0, (size_t)T * N) >> (8 * r)) & 255u);
        } else {
            act2 = dummy_action4<D>(g, lane, l, e, a.hard, a.opp_seed, genv);
            e.opp_ctr += 1u;
        }
        const bool bad = act1 > 12 || act2 > 12;                            // ValueError in the reference (env_wrappers.py:66): env untouched
        const bool ok = !bad;
        err |= bad ? ERR_BAD_ACTION : 0u;

        apply_action4<D, 0>(g, l, act1, ok, e, err);                        // core.py:510-525: P1 then P2
        apply_action4<D, 1>(g, l, act2, ok, e, err);
        reload4(e.al1, e.am1, e.rl1, ok);                                   // :528
        reload4(e.al2, e.am2, e.rl2, ok);
        const int tile_diff = update_bullets4<D>(g, slots, reg, lane, l, ok, e);   // :531

        // env_wrappers.py:333-442, f64 adds in the reference's order (every skipped term is +0.0: the running sum is never -0.0)
        double reward = 0.0;
        int done = 0;
        uint32_t info = 0u;
        {
            const int sc = e.step_count + 1;
            const int al1 = e.al1, al2 = e.al2, both = al1 & al2, over = both ^ 1;
            const int landed = e.prev2 & (al2 ^ 1), got = e.prev1 & (al1 ^ 1);
            const int win = al1 & (al2 ^ 1), lose = al2 & (al1 ^ 1);
            const int xeq = e.x1 == e.x2, yeq = e.y1 == e.y2, aligned = xeq | yeq;
            const int dist = abs(e.x1 - e.x2) + abs(e.y1 - e.y2);
            const int shot = (act1 - 1) & 3;
            const int aim_v = ((e.y1 < e.y2) & (shot == 2)) | ((e.y1 > e.y2) & (shot == 0));
            const int aim_h = ((e.x1 < e.x2) & (shot == 1)) | ((e.x1 > e.x2) & (shot == 3));
            const int aiming = both & aligned & (act1 >= 5) & (e.am1 > 0) & (xeq ? aim_v : aim_h);
            const int timeout = sc >= a.d.max_steps;
            const double prox = proxl[dist];                                 // dist <= W + H - 2: always in the table
            double rw = 0.0;
            rw += tile_diff > 0 ? (double)tile_diff * 0.01 : 0.0;             // :349-354
            rw += landed ? 1.0 : 0.0;                                         // :357-362
            rw += got ? -0.01 : 0.0;                                          // :364-369
            rw += (al1 & (e.am1 == 0)) ? -0.001 : 0.0;                        // :372-373
            rw += both ? prox : 0.0;                                          // :378-382
            rw += (both & aligned) ? 0.002 : 0.0;                             // :385-387
            rw += aiming ? 0.05 : 0.0;                                        // :391-405
            rw += over ? (win ? 10.0 : (lose ? -0.1 : 0.0)) : -0.001;         // :408-427
            rw += (timeout & both) ? -2.0 : 0.0;                              // :434-438
            if (ok) {
                reward = rw;
                done = over | timeout;
                info = (uint32_t)(landed * INV_INFO_LANDED_HIT | got * INV_INFO_GOT_HIT | (over & win) * INV_INFO_WIN | (over & lose) * INV_INFO_LOSE);
                e.step_count = sc;
                e.prev1 = al1;
                e.prev2 = al2;
                e.ep_ret += rw;
            }
        }
        if (l == 0 && valid) {
            if (a.o.reward) a.o.reward[idx] = (float)reward;
            if (a.o.reward64) a.o.reward64[idx] = reward;
            if (a.o.done) a.o.done[idx] = (uint8_t)done;
            if (a.o.info_flags) a.o.info_flags[idx] = (uint8_t)info;
            if (a.o.episode_steps) a.o.episode_steps[idx] = e.step_count;
            if (a.o.episode_return) a.o.episode_return[idx] = e.ep_ret;
        }

        WSTAMP(2);                                                          // tick + reward + scalar outputs issued
        // training.py:148-151: rows whose episode ended are reset one after the other by the whole wave (MT19937 of inv_device.h)
        uint64_t need = __ballot(auto_reset && done && valid && l == 0);
        while (need) {
            const int rl = __ffsll((unsigned long long)need) - 1;           // lane 16 * row
            need &= need - 1;
            const int rr = rl >> 4;
            Env et;
            et.mti = (int)rdlane((uint32_t)e.mti, rl);
            uint32_t *rg = wl + kTabWords + rr * LL::row_words + 8;
            reset_env(d, a.s, env0 + rr, lane, rg, mtscr, et);               // may use the MT scratch: restore the table
            for (int q = lane; q <= D::W + D::H; q += kWave) proxl[q] = a.s.prox[q];
            wave_sync();
            if (r == rr) {
                e.x1 = et.p[0].x; e.y1 = et.p[0].y; e.am1 = 6; e.rl1 = 0; e.al1 = 1;
                e.x2 = et.p[1].x; e.y2 = et.p[1].y; e.am2 = 6; e.rl2 = 0; e.al2 = 1;
                e.nb = 0; e.bw = 0u; e.step_count = 0; e.ep_ret = 0.0; e.prev1 = e.prev2 = 1; e.mti = et.mti;
            }
        }

        WSTAMP(3);                                                          // resets done
        const int nvalid = min(kRows, N - env0);                            // rows of this wave that are envs
        const size_t idxr = (size_t)t * N + env;                            // this row's slot (past the end for rows >= nvalid: never written)
        // the observation regions of a wave's envs are adjacent: render4 gets the (wave-uniform) address of the first one
        if (a.o.obs) render4<D>(g, reg, tab, lane, 0, e, valid, nvalid, LL::row_words, a.o.obs + idx0 * obs_stride, a.o.extra ? a.o.extra + idxr * 4 : nullptr, obits, a.obs_wt != 0);
        if (a.o.obs_p2) render4<D>(g, reg, tab, lane, 1, e, valid, nvalid, LL::row_words, a.o.obs_p2 + idx0 * obs_stride, a.o.extra_p2 ? a.o.extra_p2 + idxr * 4 : nullptr, obits, a.obs_wt != 0);
    }
    WSTAMP(4);                                                              // observation stores issued
    store_rows<D>(a.s, env, l, valid, rowp, e);
#ifdef INV_STEP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WSTAMP(5);                                                              // every store of this wave acknowledged
    if (lane == 0 && env0 / kRows < kStampEnvs) {
        for (int q = 0; q < 8; ++q) g_step_stamps[(env0 / kRows) * 8 + q] = wst[q];
    }
#endif
    // rows past N step a copy of env 0 on whatever bytes follow the action buffer: their state is never stored and their
    // errors are not errors
    if (__ballot(err != 0u && valid) != 0ull) {
        if (err && l == 0 && valid) atomicOr(a.s.err, err);
    }
}

template <class D>
__global__ __launch_bounds__(kWave * kWavesPerBlock) void k_reset(Dims dd, State s, const uint8_t *mask)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (kWave - 1), wv = rfl((int)(threadIdx.x >> 6));
    const int env = blockIdx.x * kWavesPerBlock + wv;
    if (env >= dd.N) return;
    if (mask && rfl((int)mask[env]) == 0) return;
    const D d(dd);
    const WaveLds<D> L(d, lds, wv);
    Env e;
    load_env(d, s, env, lane, L.g, e);
    reset_env(d, s, env, lane, L.g, L.scr, e);
    store_env(d, s, env, lane, L.g, e);
}

template <class D>
__global__ __launch_bounds__(kWave * kWavesPerBlock) void k_render(Dims dd, State s, int player, float *obs, float *extra, int as_bits, int obs_wt)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (kWave - 1), wv = rfl((int)(threadIdx.x >> 6));
    const int env = blockIdx.x * kWavesPerBlock + wv;
    if (env >= dd.N) return;
    const D d(dd);
    const WaveLds<D> L(d, lds, wv);
    init_nibble_table(reinterpret_cast<float *>(L.tabw), lane);
    Env e;
    load_env(d, s, env, lane, L.g, e);
    render_obs(d, L.g, L.scr, reinterpret_cast<const float4 *>(L.tabw), lane, player, e,
               obs + (size_t)env * (as_bits ? (size_t)d.OW : (size_t)12 * d.WH), extra ? extra + (size_t)env * 4 : nullptr, as_bits != 0, obs_wt != 0);
}

// dummy_opponent_policy (env_wrappers.py:69-170) as a function of its own: the P2 action id the device dummy takes at every env's
// NEXT step, from the env's current state and counter; nothing is written back (the counter advances only when a step consumes it).
template <class D>
__global__ __launch_bounds__(kWave * kWavesPerBlock) void k_dummy_actions(Dims dd, State s, int hard, uint32_t opp_seed, uint32_t env_offset, uint8_t *out)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (kWave - 1), wv = rfl((int)(threadIdx.x >> 6));
    const int env = blockIdx.x * kWavesPerBlock + wv;
    if (env >= dd.N) return;
    const D d(dd);
    const WaveLds<D> L(d, lds, wv);
    Env e;
    load_env(d, s, env, lane, L.g, e);
    const int act = rfl(dummy_action(d, L.g, lane, e, hard, opp_seed, env_offset + (uint32_t)env, e.opp_ctr));
    if (lane == 0) out[env] = (uint8_t)act;
}

// Packed observation bitstring -> f32 planes: float4 q of env i = nibble (q & 7) of word q >> 3 (the same
// expansion render_obs applies in LDS), for the consumers that want f32 (the PPO update's minibatches).
__global__ __launch_bounds__(256) void k_unpack_obs(const uint32_t *__restrict__ bits, long long total, int f4_per_env, int words_per_env,
                                                    float4 *__restrict__ obs)
{
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= total) return;
    const long long env = j / f4_per_env;
    const int q = (int)(j - env * f4_per_env);
    const uint32_t nib = (bits[env * words_per_env + (q >> 3)] >> ((q & 7) * 4)) & 15u;
    obs[j] = make_float4((nib & 1u) ? 1.0f : 0.0f, (nib & 2u) ? 1.0f : 0.0f, (nib & 4u) ? 1.0f : 0.0f, (nib & 8u) ? 1.0f : 0.0f);
}

// Pick the dims policy: the two configurations of BASELINE.json get fully folded code.
#define INV_DISPATCH_DIMS(dims, ...)                                                      \
    do {                                                                                  \
        if ((dims).W == 15 && (dims).H == 10 && (dims).K == 16) { using DT = StaticDims<15, 10, 16>; __VA_ARGS__; } \
        else if ((dims).W == 32 && (dims).H == 32 && (dims).K == 16) { using DT = StaticDims<32, 32, 16>; __VA_ARGS__; } \
        else { using DT = RuntimeDims; __VA_ARGS__; }                                     \
    } while (0)

// random.Random(seed): init_by_array(key) of CPython's _randommodule.c, one thread per env.
// key[env] = {low word, high word of abs(seed)}; key length 2 iff the high word is non-zero.
__global__ void k_seed(int N, uint32_t *mt, uint32_t *head, const uint32_t *key, const uint8_t *mask)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    if (mask && !mask[env]) return;
    uint32_t *m = mt + (size_t)env * kMtStride;
    const uint32_t k0 = key[2 * env], k1 = key[2 * env + 1];
    const int klen = k1 ? 2 : 1;
    uint32_t prev = 19650218u;                       // init_genrand(19650218)
    m[0] = prev;
    for (int i = 1; i < 624; ++i) {
        prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i;
        m[i] = prev;
    }
    int i = 1, j = 0;
    prev = m[0];
    for (int k = 624; k; --k) {
        const uint32_t cur = (m[i] ^ ((prev ^ (prev >> 30)) * 1664525u)) + (j ? k1 : k0) + (uint32_t)j;
        m[i] = cur;
        prev = cur;
        ++i; ++j;
        if (i >= 624) { m[0] = prev; i = 1; }
        if (j >= klen) j = 0;
    }
    for (int k = 623; k; --k) {
        const uint32_t cur = (m[i] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
        m[i] = cur;
        prev = cur;
        ++i;
        if (i >= 624) { m[0] = prev; i = 1; }
    }
    m[0] = 0x80000000u;
    head[(size_t)env * kHeadWords + HD_MTI] = 624u;
    head[(size_t)env * kHeadWords + HD_OPP] = 0u;
}

__global__ void k_get_state(Dims d, State s, inv_state_view v)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= d.N) return;
    const uint32_t *h = s.head + (size_t)env * kHeadWords;
    const uint32_t fl = h[HD_FLAGS];
    const int nb = (int)((fl >> 8) & 255u);
    if (v.players) {
        int32_t *p = v.players + (size_t)env * 10;
        for (int q = 0; q < 2; ++q) {
            const uint32_t w = h[q];
            p[5 * q + 0] = (int)(w & 255u); p[5 * q + 1] = (int)((w >> 8) & 255u);
            p[5 * q + 2] = (int)((w >> 16) & 255u); p[5 * q + 3] = (int)(w >> 24);
            p[5 * q + 4] = (int)((fl >> q) & 1u);
        }
    }
    if (v.nbullets) v.nbullets[env] = nb;
    if (v.bullets) for (int i = 0; i < d.K; ++i) v.bullets[(size_t)env * d.K + i] = i < nb ? s.bullets[(size_t)env * d.K + i] : (uint16_t)0;
    if (v.grid) for (int w = 0; w < d.GW; ++w) v.grid[(size_t)env * d.GW + w] = s.grid[(size_t)env * d.GW + w];
    if (v.step_count) v.step_count[env] = (int32_t)h[HD_STEP];
    if (v.episode_return) v.episode_return[env] = __longlong_as_double((long long)((uint64_t)h[HD_RET_LO] | ((uint64_t)h[HD_RET_HI] << 32)));
    if (v.prev_alive) { v.prev_alive[2 * env] = (uint8_t)((fl >> 2) & 1u); v.prev_alive[2 * env + 1] = (uint8_t)((fl >> 3) & 1u); }
    if (v.opp_ctr) v.opp_ctr[env] = h[HD_OPP];
    if (v.mt_index) v.mt_index[env] = (int32_t)h[HD_MTI];
    if (v.mt) for (int i = 0; i < 624; ++i) v.mt[(size_t)env * 624 + i] = s.mt[(size_t)env * kMtStride + i];
}

__global__ void k_set_state(Dims d, State s, inv_state_view v)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= d.N) return;
    uint32_t *h = s.head + (size_t)env * kHeadWords;
    uint32_t fl = h[HD_FLAGS];
    if (v.players) {
        const int32_t *p = v.players + (size_t)env * 10;
        for (int q = 0; q < 2; ++q) {
            h[q] = ((uint32_t)p[5 * q + 0] & 255u) | (((uint32_t)p[5 * q + 1] & 255u) << 8) |
                   (((uint32_t)p[5 * q + 2] & 255u) << 16) | (((uint32_t)p[5 * q + 3] & 255u) << 24);
            fl = (fl & ~(1u << q)) | ((p[5 * q + 4] ? 1u : 0u) << q);
        }
    }
    if (v.nbullets) {
        int nb = v.nbullets[env];
        if (nb > d.K) { nb = d.K; atomicOr(s.err, ERR_BULLET_OVERFLOW); }
        if (nb < 0) nb = 0;
        fl = (fl & ~0xff00u) | ((uint32_t)nb << 8);
    }
    if (v.prev_alive) fl = (fl & ~(FL_PREV1 | FL_PREV2)) | (v.prev_alive[2 * env] ? FL_PREV1 : 0u) | (v.prev_alive[2 * env + 1] ? FL_PREV2 : 0u);
    h[HD_FLAGS] = fl;
    if (v.bullets) for (int i = 0; i < d.K; ++i) s.bullets[(size_t)env * d.K + i] = v.bullets[(size_t)env * d.K + i];
    if (v.grid) {
        const uint32_t lastmask = (d.WH & 31) ? ((1u << (d.WH & 31)) - 1u) : 0xffffffffu;
        for (int w = 0; w < d.GW; ++w) s.grid[(size_t)env * d.GW + w] = v.grid[(size_t)env * d.GW + w] & (w == d.GW - 1 ? lastmask : 0xffffffffu);
    }
    if (v.step_count) h[HD_STEP] = (uint32_t)v.step_count[env];
    if (v.episode_return) {
        const uint64_t rb = (uint64_t)__double_as_longlong(v.episode_return[env]);
        h[HD_RET_LO] = (uint32_t)rb; h[HD_RET_HI] = (uint32_t)(rb >> 32);
    }
    if (v.opp_ctr) h[HD_OPP] = v.opp_ctr[env];
    if (v.mt_index) h[HD_MTI] = (uint32_t)v.mt_index[env];
    if (v.mt) for (int i = 0; i < 624; ++i) s.mt[(size_t)env * kMtStride + i] = v.mt[(size_t)env * 624 + i];
}

__global__ void k_init_head(int N, uint32_t *head)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    uint32_t *h = head + (size_t)env * kHeadWords;
    // a blank env: players at (1,1), 6 ammo, alive; MT unseeded (index 624 over a zero state)
    h[HD_P1] = 1u | (1u << 8) | (6u << 16);
    h[HD_P2] = 1u | (1u << 8) | (6u << 16);
    h[HD_FLAGS] = FL_ALIVE1 | FL_ALIVE2 | FL_PREV1 | FL_PREV2;
    h[HD_STEP] = 0u; h[HD_OPP] = 0u; h[HD_MTI] = 624u; h[HD_RET_LO] = 0u; h[HD_RET_HI] = 0u;
}

__global__ void k_uniform_actions(uint8_t *a, int n, uint32_t seed, uint32_t env_offset, uint32_t t)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const U4 x = philox4x32(t, env_offset + (uint32_t)i, 0u, kTagAction, seed, kKey1);
    a[i] = (uint8_t)__umulhi(x.x, 13u);
}

// =================================================================== host side

struct inv_handle {
    inv_config cfg;
    Dims d;
    State s;
    double *prox;
    uint32_t *keys;      // [N][2] staging for inv_seed, followed by N mask bytes
    uint32_t *keys_host; // pinned, same layout
    int lds_bytes;
    int step4_mode;      // full unmasked steps: 0 wave-per-env kernel, 1 auto (see launch_step), 2 four-envs-per-wave kernel
    int obs_store_plain; // inv_set_obs_store_policy(INV_OBS_STORE_PLAIN): never write the f32 observations through
};

static thread_local char g_err[512] = "";

int inv_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define fail inv_fail

int inv_cu_count()
{
    static int cus[64] = { 0 };                      // per device ordinal; 0 = not queried yet
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0)
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    return cus[dev];
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(INV_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

static inline int env_blocks(int n) { return (n + kWavesPerBlock - 1) / kWavesPerBlock; }

extern "C" {

#ifdef INV_STEP_STAMPS
int inv_step_debug_stamps(unsigned long long *out_host /* [16384][8] */)
{
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_step_stamps), sizeof(unsigned long long) * kStampEnvs * 8) == hipSuccess ? INV_OK : INV_ERR_HIP;
}
#endif

const char *inv_last_error(void) { return g_err; }
int inv_abi_version(void) { return INV_ABI_VERSION; }

int inv_create(const inv_config *cfg, inv_handle **out)
{
    if (!cfg || !out) return fail(INV_ERR_BAD_ARG, "inv_create: NULL argument");
    *out = nullptr;
    if (cfg->num_envs <= 0) return fail(INV_ERR_BAD_ARG, "inv_create: num_envs must be > 0");
    if (cfg->width < 3 || cfg->height < 3)
        return fail(INV_ERR_BAD_DIMS, "empty range for randrange() (%d, %d, %d)", 1, cfg->width < 3 ? cfg->width - 1 : cfg->height - 1, 0);
    if (cfg->width > 64 || cfg->height > 64) return fail(INV_ERR_BAD_DIMS, "inv_create: width/height must be <= 64");
    if (cfg->max_bullets < 1 || cfg->max_bullets > 64) return fail(INV_ERR_BAD_ARG, "inv_create: max_bullets must be 1..64");
    if (cfg->max_episode_steps < 1) return fail(INV_ERR_BAD_ARG, "inv_create: max_episode_steps must be >= 1");
    if (cfg->opponent != INV_OPP_EXTERNAL && cfg->opponent != INV_OPP_DUMMY) return fail(INV_ERR_BAD_ARG, "Unknown opponent_type: %d", cfg->opponent);

    HIP_TRY(hipSetDevice(cfg->device));
    inv_handle *h = new (std::nothrow) inv_handle();        // value-initialised: every pointer starts NULL
    if (!h) return fail(INV_ERR_BAD_ARG, "inv_create: out of host memory");
    // from here on a failing HIP call releases whatever was allocated so far (inv_destroy accepts a partial handle)
#undef HIP_TRY
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (void)inv_destroy(h);                                                                  \
            return fail(INV_ERR_HIP, "inv_create: %s: %s", #expr, hipGetErrorString(e_));          \
        }                                                                                          \
    } while (0)
    h->cfg = *cfg;
    Dims &d = h->d;
    d.N = cfg->num_envs; d.W = cfg->width; d.H = cfg->height; d.WH = d.W * d.H; d.K = cfg->max_bullets;
    d.GW = (d.WH + 31) / 32;
    d.OW = (12 * d.WH + 31) / 32;
    d.F4 = 3 * d.WH;
    d.scr_words = scr_words_for(d.GW, d.K, d.OW);     // table + grid + slots + scratch, 16-B multiple
    d.max_steps = cfg->max_episode_steps;
    h->lds_bytes = d.scr_words * 4 * kWavesPerBlock;
    h->step4_mode = 1;

    const size_t n = (size_t)d.N;
    HIP_TRY(hipMalloc(&h->s.head, n * kHeadWords * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&h->s.grid, n * d.GW * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&h->s.bullets, n * d.K * sizeof(uint16_t)));
    HIP_TRY(hipMalloc(&h->s.mt, n * kMtStride * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&h->s.err, sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&h->prox, (size_t)(d.W + d.H + 1) * sizeof(double)));
    HIP_TRY(hipMalloc(&h->keys, n * 2 * sizeof(uint32_t) + n));
    HIP_TRY(hipHostMalloc(&h->keys_host, n * 2 * sizeof(uint32_t) + n));
    HIP_TRY(hipMemset(h->s.grid, 0, n * d.GW * sizeof(uint32_t)));
    HIP_TRY(hipMemset(h->s.bullets, 0, n * d.K * sizeof(uint16_t)));
    HIP_TRY(hipMemset(h->s.mt, 0, n * kMtStride * sizeof(uint32_t)));
    HIP_TRY(hipMemset(h->s.err, 0, sizeof(uint32_t)));

    // env_wrappers.py:378-382: 0.002 * (1.0 - dist / max_dist), IEEE f64 on the host
    const int max_dist = d.W + d.H;
    double prox[129];
    for (int dist = 0; dist <= max_dist; ++dist) prox[dist] = 0.002 * (1.0 - (double)dist / (double)max_dist);
    HIP_TRY(hipMemcpy(h->prox, prox, (size_t)(max_dist + 1) * sizeof(double), hipMemcpyHostToDevice));
    h->s.prox = h->prox;

    hipLaunchKernelGGL(k_init_head, dim3((d.N + 255) / 256), dim3(256), 0, 0, d.N, h->s.head);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    *out = h;
    return INV_OK;
#undef HIP_TRY
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(INV_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)
}

int inv_destroy(inv_handle *h)
{
    if (!h) return INV_OK;
    void *dev[] = { h->s.head, h->s.grid, h->s.bullets, h->s.mt, h->s.err, h->prox, h->keys };
    for (void *p : dev)
        if (p) (void)hipFree(p);
    if (h->keys_host) (void)hipHostFree(h->keys_host);
    delete h;
    return INV_OK;
}

int inv_set_step_kernel(inv_handle *h, int32_t kind)
{
    if (!h) return fail(INV_ERR_BAD_ARG, "inv_set_step_kernel: NULL handle");
    if (kind != INV_STEP_KERNEL_AUTO && kind != INV_STEP_KERNEL_WAVE && kind != INV_STEP_KERNEL_ROW4)
        return fail(INV_ERR_BAD_ARG, "inv_set_step_kernel: unknown kind %d", (int)kind);
    const bool row4_ok = h->d.K == 16 && ((h->d.W == 15 && h->d.H == 10) || (h->d.W == 32 && h->d.H == 32));
    if (kind == INV_STEP_KERNEL_ROW4 && !row4_ok)
        return fail(INV_ERR_BAD_ARG, "inv_set_step_kernel: the four-envs-per-wave kernel exists for 15x10 and 32x32 boards with max_bullets = 16");
    h->step4_mode = kind == INV_STEP_KERNEL_WAVE ? 0 : (kind == INV_STEP_KERNEL_ROW4 ? 2 : 1);
    return INV_OK;
}

int inv_set_obs_store_policy(inv_handle *h, int32_t policy)
{
    if (!h) return fail(INV_ERR_BAD_ARG, "inv_set_obs_store_policy: NULL handle");
    if (policy != INV_OBS_STORE_AUTO && policy != INV_OBS_STORE_PLAIN)
        return fail(INV_ERR_BAD_ARG, "inv_set_obs_store_policy: unknown policy %d", (int)policy);
    h->obs_store_plain = policy == INV_OBS_STORE_PLAIN;
    return INV_OK;
}

int inv_get_config(const inv_handle *h, inv_config *cfg)
{
    if (!h || !cfg) return fail(INV_ERR_BAD_ARG, "inv_get_config: NULL argument");
    *cfg = h->cfg;
    return INV_OK;
}

int64_t inv_obs_floats_per_env(const inv_handle *h) { return h ? (int64_t)12 * h->d.WH : 0; }
int32_t inv_grid_words(const inv_handle *h) { return h ? h->d.GW : 0; }
int32_t inv_obs_words_per_env(const inv_handle *h) { return h ? h->d.OW : 0; }

int inv_seed_masked(inv_handle *h, const int64_t *seeds_host, const uint8_t *mask_host, void *stream)
{
    if (!h || !seeds_host) return fail(INV_ERR_BAD_ARG, "inv_seed: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)h->d.N;
    HIP_TRY(hipStreamSynchronize(st));              // keys_host may still feed an earlier copy
    uint8_t *mh = reinterpret_cast<uint8_t *>(h->keys_host + 2 * n);
    for (size_t i = 0; i < n; ++i) {
        const int64_t sd = seeds_host[i];
        const uint64_t a = sd < 0 ? (uint64_t)0 - (uint64_t)sd : (uint64_t)sd;   // abs(seed), random_seed()
        h->keys_host[2 * i] = (uint32_t)(a & 0xffffffffu);
        h->keys_host[2 * i + 1] = (uint32_t)(a >> 32);
        mh[i] = mask_host ? mask_host[i] : (uint8_t)1;
    }
    HIP_TRY(hipMemcpyAsync(h->keys, h->keys_host, n * 2 * sizeof(uint32_t) + n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_seed, dim3((h->d.N + 63) / 64), dim3(64), 0, st, h->d.N, h->s.mt, h->s.head, h->keys,
                       reinterpret_cast<const uint8_t *>(h->keys + 2 * n));
    HIP_TRY(hipGetLastError());
    return INV_OK;
}

int inv_seed(inv_handle *h, const int64_t *seeds_host, void *stream) { return inv_seed_masked(h, seeds_host, nullptr, stream); }

int inv_reset(inv_handle *h, const uint8_t *mask, void *stream)
{
    if (!h) return fail(INV_ERR_BAD_ARG, "inv_reset: NULL handle");
    INV_DISPATCH_DIMS(h->d, hipLaunchKernelGGL(k_reset<DT>, dim3(env_blocks(h->d.N)), dim3(kWave * kWavesPerBlock), h->lds_bytes,
                                               (hipStream_t)stream, h->d, h->s, mask));
    HIP_TRY(hipGetLastError());
    return INV_OK;
}

// Write-through observation stores only where every store instruction covers whole 128-B lines (inv_device.h::store_f4):
// `group4` = the launch runs k_step4 on the 15x10 board, which writes a wave's four 7 200-B regions as one 28 800-B stream.
static inline bool aligned128(const void *p) { return ((uintptr_t)p & 127u) == 0u; }
static int obs_write_through(const inv_handle *h, bool group4, const float *obs, const float *obs_p2)
{
    if (h->obs_store_plain || (!obs && !obs_p2)) return 0;
    if ((obs && !aligned128(obs)) || (obs_p2 && !aligned128(obs_p2))) return 0;
    if (group4) return h->d.N % inv4::kRows == 0;
    return h->d.WH % 8 == 0;                         // one region = 48 * WH bytes
}

static int launch_step(inv_handle *h, int T, const uint8_t *mask, const uint8_t *a1, const uint8_t *a2, uint32_t flags, const inv_step_out *out, void *stream)
{
    if (!h || !a1) return fail(INV_ERR_BAD_ARG, "inv_step: NULL handle or a1");
    if (!a2 && h->cfg.opponent != INV_OPP_DUMMY)
        return fail(INV_ERR_NO_OPPONENT, "opponent_policy required for selfplay mode");
    StepArgs a;
    memset(&a, 0, sizeof a);
    a.d = h->d; a.s = h->s; a.mask = mask; a.a1 = a1; a.a2 = a2;
    if (out) a.o = *out;
    if ((a.o.extra && !a.o.obs) || (a.o.extra_p2 && !a.o.obs_p2)) return fail(INV_ERR_BAD_ARG, "inv_step: extra without obs");
    a.flags = flags;
    a.opponent = h->cfg.opponent;
    a.hard = h->cfg.difficulty == INV_HARD;
    a.opp_seed = h->cfg.opp_seed;
    a.env_offset = (uint32_t)h->cfg.env_offset;
    a.T = T;
    // Full steps without a mask on the two static board sizes: four envs per wave (inv_step4.h) unless inv_set_step_kernel chose
    // the wave-per-env kernel (A/B runs; the parity tests go through both).
    const bool full = (flags & INV_STEP_DEFAULT) == INV_STEP_DEFAULT && !mask && h->d.K == 16;
    // AUTO: the four-env kernel wherever it measured equal or faster (profiles/r02_step_kernel_ab.json): every 15x10 step, and
    // the 32x32 board with packed observations; 32x32 with f32 planes (0.8 GB of stores per step, store-bound either way)
    // stays on the wave-per-env kernel, which was 4 % faster there.
    const bool bits = (flags & INV_STEP_OBS_BITS) != 0u;
    a.obs_wt = bits ? 0 : obs_write_through(h, false, a.o.obs, a.o.obs_p2);
    if (h->step4_mode && full && h->d.W == 15 && h->d.H == 10) {
        using DT = StaticDims<15, 10, 16>;
        a.obs_wt = bits ? 0 : obs_write_through(h, true, a.o.obs, a.o.obs_p2);
        constexpr int WPB = 4, lds = inv4::Lds4<DT>::wave_words * 4 * WPB;
        const int blocks = (h->d.N + inv4::kRows * WPB - 1) / (inv4::kRows * WPB);
        if (T == 1) hipLaunchKernelGGL((k_step4<DT, true, WPB>), dim3(blocks), dim3(kWave * WPB), lds, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((k_step4<DT, false, WPB>), dim3(blocks), dim3(kWave * WPB), lds, (hipStream_t)stream, a);
        HIP_TRY(hipGetLastError());
        return INV_OK;
    }
    if ((h->step4_mode == 2 || (h->step4_mode == 1 && bits)) && full && h->d.W == 32 && h->d.H == 32) {
        using DT = StaticDims<32, 32, 16>;
        constexpr int WPB = 2, lds = inv4::Lds4<DT>::wave_words * 4 * WPB;
        const int blocks = (h->d.N + inv4::kRows * WPB - 1) / (inv4::kRows * WPB);
        if (T == 1) hipLaunchKernelGGL((k_step4<DT, true, WPB>), dim3(blocks), dim3(kWave * WPB), lds, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((k_step4<DT, false, WPB>), dim3(blocks), dim3(kWave * WPB), lds, (hipStream_t)stream, a);
        HIP_TRY(hipGetLastError());
        return INV_OK;
    }
    if (T == 1)
        INV_DISPATCH_DIMS(h->d, hipLaunchKernelGGL((k_step<DT, true>), dim3(env_blocks(h->d.N)), dim3(kWave * kWavesPerBlock),
                                                   h->lds_bytes, (hipStream_t)stream, a));
    else
        INV_DISPATCH_DIMS(h->d, hipLaunchKernelGGL((k_step<DT, false>), dim3(env_blocks(h->d.N)), dim3(kWave * kWavesPerBlock),
                                                   h->lds_bytes, (hipStream_t)stream, a));
    HIP_TRY(hipGetLastError());
    return INV_OK;
}

int inv_step(inv_handle *h, const uint8_t *a1, const uint8_t *a2, uint32_t flags, const inv_step_out *out, void *stream)
{
    return launch_step(h, 1, nullptr, a1, a2, flags, out, stream);
}

int inv_step_masked(inv_handle *h, const uint8_t *mask, const uint8_t *a1, const uint8_t *a2, uint32_t flags,
                    const inv_step_out *out, void *stream)
{
    return launch_step(h, 1, mask, a1, a2, flags, out, stream);
}

int inv_rollout(inv_handle *h, int32_t T, const uint8_t *a1, const uint8_t *a2, uint32_t flags, const inv_step_out *out, void *stream)
{
    if (T < 1) return fail(INV_ERR_BAD_ARG, "inv_rollout: T must be >= 1");
    if (flags & ~(uint32_t)INV_STEP_OBS_BITS) return fail(INV_ERR_BAD_ARG, "inv_rollout: only INV_STEP_OBS_BITS may be passed");
    return launch_step(h, T, nullptr, a1, a2, INV_STEP_DEFAULT | INV_STEP_AUTO_RESET | flags, out, stream);
}

static int launch_render(inv_handle *h, int32_t player, float *obs, float *extra, int as_bits, void *stream)
{
    if (!h || !obs) return fail(INV_ERR_BAD_ARG, "inv_render_obs: NULL argument");
    if (player != 0 && player != 1) return fail(INV_ERR_BAD_ARG, "Invalid player ID: %d", player);   // core.py:198
    INV_DISPATCH_DIMS(h->d, hipLaunchKernelGGL(k_render<DT>, dim3(env_blocks(h->d.N)), dim3(kWave * kWavesPerBlock), h->lds_bytes,
                                               (hipStream_t)stream, h->d, h->s, player, obs, extra, as_bits, as_bits ? 0 : obs_write_through(h, false, obs, nullptr)));
    HIP_TRY(hipGetLastError());
    return INV_OK;
}

int inv_dummy_actions(inv_handle *h, int32_t difficulty, uint8_t *actions, void *stream)
{
    if (!h || !actions) return fail(INV_ERR_BAD_ARG, "inv_dummy_actions: NULL argument");
    if (difficulty != INV_EASY && difficulty != INV_HARD) return fail(INV_ERR_BAD_ARG, "inv_dummy_actions: difficulty must be INV_EASY or INV_HARD");
    INV_DISPATCH_DIMS(h->d, hipLaunchKernelGGL(k_dummy_actions<DT>, dim3(env_blocks(h->d.N)), dim3(kWave * kWavesPerBlock), h->lds_bytes,
                                               (hipStream_t)stream, h->d, h->s, difficulty == INV_HARD ? 1 : 0, h->cfg.opp_seed,
                                               (uint32_t)h->cfg.env_offset, actions));
    HIP_TRY(hipGetLastError());
    return INV_OK;
}

int inv_render_obs(inv_handle *h, int32_t player, float *obs, float *extra, void *stream)
{
    return launch_render(h, player, obs, extra, 0, stream);
}

int inv_render_obs_bits(inv_handle *h, int32_t player, uint32_t *bits, float *extra, void *stream)
{
    return launch_render(h, player, reinterpret_cast<float *>(bits), extra, 1, stream);
}

int inv_unpack_obs(const uint32_t *bits, int64_t n, int32_t floats_per_env, float *obs, void *stream)
{
    if (!bits || !obs || n < 0 || floats_per_env <= 0 || (floats_per_env & 3)) return fail(INV_ERR_BAD_ARG, "inv_unpack_obs: bad argument");
    if (n == 0) return INV_OK;
    const long long total = (long long)n * (floats_per_env / 4);
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(INV_ERR_BAD_ARG, "inv_unpack_obs: n too large for one launch");
    hipLaunchKernelGGL(k_unpack_obs, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, bits, total, floats_per_env / 4,
                       (floats_per_env + 31) / 32, reinterpret_cast<float4 *>(obs));
    HIP_TRY(hipGetLastError());
    return INV_OK;
}

int inv_get_state(inv_handle *h, const inv_state_view *dst, void *stream)
{
    if (!h || !dst) return fail(INV_ERR_BAD_ARG, "inv_get_state: NULL argument");
    hipLaunchKernelGGL(k_get_state, dim3((h->d.N + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->d, h->s, *dst);
    HIP_TRY(hipGetLastError());
    return INV_OK;
}

int inv_set_state(inv_handle *h, const inv_state_view *src, void *stream)
{
    if (!h || !src) return fail(INV_ERR_BAD_ARG, "inv_set_state: NULL argument");
    hipLaunchKernelGGL(k_set_state, dim3((h->d.N + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->d, h->s, *src);
    HIP_TRY(hipGetLastError());
    return INV_OK;
}

int inv_check(inv_handle *h, void *stream)
{
    if (!h) return fail(INV_ERR_BAD_ARG, "inv_check: NULL handle");
    hipStream_t st = (hipStream_t)stream;
    uint32_t e = 0u;
    HIP_TRY(hipMemcpyAsync(&e, h->s.err, sizeof e, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (e) {
        HIP_TRY(hipMemsetAsync(h->s.err, 0, sizeof e, st));
        if (e & ERR_BAD_ACTION) return fail(INV_ERR_BAD_ACTION, "Invalid action_id, must be 0-12");
        return fail(INV_ERR_BULLET_OVERFLOW, "more than max_bullets=%d live bullets in one env", h->d.K);
    }
    return INV_OK;
}

int inv_uniform_actions(uint8_t *a, int32_t n, uint32_t seed, int64_t env_offset, uint32_t t, void *stream)
{
    if (!a || n < 0) return fail(INV_ERR_BAD_ARG, "inv_uniform_actions: bad argument");
    if (n == 0) return INV_OK;
    hipLaunchKernelGGL(k_uniform_actions, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, n, seed, (uint32_t)env_offset, t);
    HIP_TRY(hipGetLastError());
    return INV_OK;
}

}  // extern "C"
