/*
 * inv_oracle.h -- CPU restatement of the reference's Inversus hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it, and there only as the checker / CPU yardstick.
 *
 * Parity status: PINNED.  Every function here is checked against golden
 * vectors produced by importing the Python reference in the build container
 * (tests/golden/gen_golden.py -> the tests/golden/ .npz fixtures) and against the
 * known-answer tests lifted from the reference's own tests/ directory.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * the reference checkout).  The style is deliberately the reference's own
 * (array-of-structs, ordered bullet list, tile groups in first-arrival order)
 * so that it is an independent check of the wave-parallel HIP kernels.
 */
#ifndef INV_ORACLE_H
#define INV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_DIM      64
#define ORC_MAX_BULLETS  256

/* game_types.py:8-11 -- tile colours; bit value used everywhere: 1 == WHITE */
enum { ORC_BLACK = 0, ORC_WHITE = 1 };
/* observation channel order of directions, env_wrappers.py:216-221 */
enum { ORC_UP = 0, ORC_RIGHT = 1, ORC_DOWN = 2, ORC_LEFT = 3 };
/* action types, game_types.py:22-27, numbered so that type == (id + 3) >> 2 */
enum { ORC_NONE = 0, ORC_MOVE = 1, ORC_SHOOT = 2, ORC_CHARGE = 3 };

typedef struct { int x, y, dir, owner; } orc_bullet;          /* game_types.py:65-71 (owner 0=P1, 1=P2) */
typedef struct { int x, y, ammo, max_ammo, reload, alive; } orc_player; /* game_types.py:52-62 */

typedef struct {
    int W, H;
    uint8_t grid[ORC_MAX_DIM * ORC_MAX_DIM];   /* [y*W + x], 1 == WHITE */
    orc_player p[2];
    orc_bullet bullets[ORC_MAX_BULLETS];
    int nb;
    /* core.py:41 -- random.Random == MT19937 */
    uint32_t mt[624];
    int mti;
    /* wrapper state, env_wrappers.py:261-270 */
    int max_episode_steps;
    int step_count;
    double episode_return;
    int prev_alive[2];
    /* counter for the device-style dummy opponent RNG (not in the reference) */
    uint32_t opp_ctr;
} orc_env;

typedef struct {
    int landed_hit, got_hit, win, lose, episode_steps;
    double episode_return;
} orc_info;

/* ---- MT19937 exactly as CPython's _randommodule.c uses it ---- */
void     orc_mt_seed(orc_env *e, int64_t seed);          /* random.Random(seed) for int seeds */
uint32_t orc_mt_u32(orc_env *e);                         /* genrand_uint32 */
int      orc_randbelow(orc_env *e, int n);               /* Random._randbelow_with_getrandbits */

/* ---- core.py ---- */
int  orc_init(orc_env *e, int W, int H, int max_episode_steps);  /* core.py:20-53 without the reset */
void orc_core_reset(orc_env *e);                         /* core.py:55-154 (seed=None branch: keep stream) */
int  orc_try_move(orc_env *e, int pid, int dir);         /* core.py:249-296 */
int  orc_spawn_bullet(orc_env *e, int pid, int dir);     /* core.py:298-326 */
int  orc_spawn_wide(orc_env *e, int pid, int dir);       /* core.py:328-381 */
void orc_reload(orc_env *e);                             /* core.py:383-397 */
void orc_update_bullets(orc_env *e);                     /* core.py:399-475 */
int  orc_round_over(const orc_env *e);                   /* core.py:477-481 */
int  orc_winner(const orc_env *e);                       /* core.py:483-495: 0 none/tie, 1 P1, 2 P2 */
int  orc_step_players(orc_env *e, int a1, int a2);       /* env_wrappers.py:20-66 + core.py:497-531; -1 on bad id */

/* ---- env_wrappers.py ---- */
void orc_build_obs(const orc_env *e, int pid, float *grid12, float *extra4);  /* :173-245 */
void orc_wrapper_reset(orc_env *e);                      /* :272-284 */
int  orc_wrapper_step(orc_env *e, int a1, int a2, double *reward, int *done, orc_info *info); /* :286-444 */

/* dummy opponent, env_wrappers.py:69-170, as a pure function of the state and the
 * random draws it would have consumed from the global `random` module:
 *   u[0] = random.random() at :96, u[1] at :105, u[2] at :123 (easy only);
 *   perm_a = result order of random.shuffle(directions) at :106 (only if u[1] < p_rand),
 *   swap_c = 1 if random.shuffle(candidates) at :138 swapped a 2-element list,
 *   perm_b = result order of random.shuffle(directions) at :155, applied to the list
 *            as it stands (already permuted by perm_a if that shuffle ran).
 * perm[i] = index into the list before the shuffle of the element placed at i. */
int orc_dummy_decide(const orc_env *e, int hard, const double u[3],
                     const int perm_a[4], int swap_c, const int perm_b[4]);

/* The counter RNGs the HIP kernels use: a lowbias32 counter hash for the on-device dummy
 * opponent's draws, Philox4x32-10 for synthetic action streams.  Not in the reference (which
 * uses the process-global `random`), so parity with the reference is statistical. */
void orc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                    uint32_t k0, uint32_t k1, uint32_t out[4]);
int  orc_dummy_device(orc_env *e, int hard, uint32_t seed, uint32_t global_env); /* consumes e->opp_ctr */
int  orc_uniform_action(uint32_t seed, uint32_t global_env, uint32_t t);         /* 0..12 */

/* ---- ppo_agent.py:127-157, strict f32 ---- */
void orc_gae_flat(const float *r, const float *v, const uint8_t *done, int64_t n,
                  float gamma, float gamma_lam, float last_value, float *adv, float *ret);
void orc_gae_per_env(const float *r, const float *v, const uint8_t *done, int T, int N,
                     float gamma, float gamma_lam, const float *last_value /* [N] or NULL */,
                     float *adv, float *ret);

/* ---- batch helpers (CPU baseline + parity drivers) ---- */
orc_env *orc_batch_create(int n, int W, int H, int max_episode_steps);
void     orc_batch_destroy(orc_env *envs);
orc_env *orc_batch_at(orc_env *envs, int i);
void     orc_batch_seed_reset(orc_env *envs, int n, const int64_t *seeds);
/* One vector step like MultiEnvRunner.step + the training loop's auto-reset
 * (training.py:128-151).  a2 == NULL -> device-style dummy opponent. */
void     orc_batch_step(orc_env *envs, int n, const uint8_t *a1, const uint8_t *a2,
                        int hard, uint32_t opp_seed, uint32_t env_offset, int auto_reset,
                        float *obs, float *extra, float *reward, double *reward64,
                        uint8_t *done, orc_info *info, int nthreads);

/* canonical state exchange (same packing as include/inversus_hip.h) */
void orc_get_grid_bits(const orc_env *e, uint32_t *words);
void orc_set_grid_bits(orc_env *e, const uint32_t *words);
uint16_t orc_pack_bullet(const orc_bullet *b);
void orc_unpack_bullet(uint16_t w, orc_bullet *b);

#ifdef __cplusplus
}
#endif
#endif
