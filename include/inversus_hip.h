/*
 * inversus_hip.h -- C ABI of libinversus_hip.so: the MI355X (gfx950) batched
 * Inversus environment, observation render, reward/termination and GAE scan.
 *
 * The reference (pure Python) has no FFI; its boundary for this path is the
 * duck-typed API of inversus_rl/env_wrappers.py (MultiEnvRunner :447-528,
 * SingleInversusRLEnv :248-444, build_observation :173-245, discrete_to_action
 * :20-66) over inversus/core.py (InversusEnv :17-544), plus
 * PPOAgent.compute_advantages (inversus_rl/ppo_agent.py:127-157).  Every entry
 * point below names the reference interface it replaces; INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in signatures (`stream` is a
 *     hipStream_t passed as void*; NULL = the default stream).
 *   - every data pointer is a DEVICE pointer unless the parameter says `_host`.
 *   - all launches are asynchronous on `stream`; nothing synchronises except
 *     inv_check() and the `_host` copies.
 *   - the uint8 inputs of inv_step / inv_step_masked / inv_rollout (a1, a2, mask) are read as
 *     aligned 32-bit words through the scalar cache: the ALLOCATION behind each of them must
 *     extend to the next 4-byte boundary past its last byte (hipMalloc and torch allocations
 *     always do; a hand-carved sub-buffer at the very end of an allocation of odd size must be
 *     padded).  The extra bytes are never interpreted; nothing beyond that boundary is read.
 *   - return value: INV_OK or a negative inv_status; inv_last_error() gives text.
 *   - env i of a handle has GLOBAL index env_offset + i (multi-GPU sharding: one
 *     handle per rank over a contiguous global range; results do not depend on
 *     how the range is cut).
 *
 * Canonical encodings (also used by oracle/ and tests/golden/)
 *   grid    u32 words [N][GW], GW = ceil(W*H/32); bit (y*W + x) set <=> WHITE
 *   player  i32 [N][2][5] = x, y, ammo, reload_counter, alive (P1 then P2)
 *   bullet  u16 [N][K] = x | y<<6 | dir<<12 | owner<<14 ; list order is state
 *           dir 0=UP 1=RIGHT 2=DOWN 3=LEFT (obs channel order), owner 0=P1 1=P2
 *   obs     f32 [N][12][H][W], extra f32 [N][4]  (env_wrappers.py:185-245)
 *   action  u8 0..12 (env_wrappers.py:24-37): type=(id+3)>>2, dir=(id-1)&3
 */
#ifndef INVERSUS_HIP_H
#define INVERSUS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INV_ABI_VERSION 8

typedef enum {
    INV_OK = 0,
    INV_ERR_BAD_ARG = -1,
    INV_ERR_BAD_DIMS = -2,        /* W or H < 3: "empty range for randrange" (core.py:69-70) */
    INV_ERR_HIP = -3,
    INV_ERR_BAD_ACTION = -4,      /* action id outside 0..12: ValueError (env_wrappers.py:66) */
    INV_ERR_BULLET_OVERFLOW = -5, /* more than max_bullets live bullets in one env */
    INV_ERR_NO_OPPONENT = -6      /* a2 == NULL on a handle without a device opponent (env_wrappers.py:309) */
} inv_status;

enum { INV_OPP_EXTERNAL = 0,      /* "selfplay": caller supplies a2 every step */
       INV_OPP_DUMMY = 1 };       /* scripted dummy computed on the device (env_wrappers.py:69-170) */
enum { INV_EASY = 0, INV_HARD = 1 };

/* inv_step flags */
enum {
    INV_PHASE_ACTIONS = 1u,       /* core.py:510-525 */
    INV_PHASE_RELOAD  = 2u,       /* core.py:528 */
    INV_PHASE_BULLETS = 4u,       /* core.py:531 */
    INV_PHASE_WRAPPER = 8u,       /* step_count, reward, done, info (env_wrappers.py:318-442) */
    INV_STEP_AUTO_RESET = 16u,    /* training.py:148-151: reset done envs, emit the reset obs */
    INV_STEP_OBS_BITS = 32u,      /* obs / obs_p2 receive the PACKED observation instead of f32 planes: they point at
                                     uint32_t[N][inv_obs_words_per_env] and bit c*H*W + y*W + x of an env's bitstring is set
                                     iff build_observation's grid[c][y][x] == 1.0 (env_wrappers.py:196-235; every entry of
                                     that array is 0.0 or 1.0, so the bitstring is lossless; pad bits are 0).  32x fewer
                                     bytes for on-device consumers: inv_policy_trunk_bits, rollout buffers, inv_unpack_obs */
    INV_STEP_DEFAULT  = 15u
};

/* info_flags bits (env_wrappers.py:360-369, 413-427) */
enum { INV_INFO_LANDED_HIT = 1, INV_INFO_GOT_HIT = 2, INV_INFO_WIN = 4, INV_INFO_LOSE = 8 };

enum { INV_GAE_REFERENCE_FLAT = 0,  /* one chain over the env-interleaved buffer (training.py:128-137 fill order) */
       INV_GAE_PER_ENV = 1 };       /* independent chain per env column of a [T][N] buffer */

typedef struct inv_handle inv_handle;

typedef struct {
    int32_t  num_envs;            /* envs in this handle (shard) */
    int32_t  width, height;       /* 3..64; reference default 15 x 10 (config.py:7-8) */
    int32_t  max_bullets;         /* K, 1..64; 16 covers every reachable state up to 64x64 */
    int32_t  max_episode_steps;   /* env_wrappers.py:251, default 500 */
    int32_t  opponent;            /* INV_OPP_* */
    int32_t  difficulty;          /* INV_EASY / INV_HARD */
    int32_t  device;              /* HIP device ordinal */
    uint32_t opp_seed;            /* key of the dummy's counter RNG */
    int64_t  env_offset;          /* global index of env 0 */
} inv_config;

/* Optional per-step outputs; any pointer may be NULL. */
typedef struct {
    float   *obs, *extra;         /* P1 observation of the state the caller continues from (see INV_STEP_OBS_BITS) */
    float   *obs_p2, *extra_p2;   /* same state from P2's perspective (self-play opponent input) */
    float   *reward;              /* f32(f64 reward), env_wrappers.py:525 */
    double  *reward64;            /* the f64 sum itself (parity checks) */
    uint8_t *done;
    uint8_t *info_flags;          /* INV_INFO_* */
    int32_t *episode_steps;       /* info["episode_steps"] */
    double  *episode_return;      /* info["episode_return"] */
} inv_step_out;

/* Canonical state view for inv_get_state / inv_set_state; NULL members are skipped. */
typedef struct {
    uint32_t *grid;               /* [N][GW] */
    int32_t  *players;            /* [N][2][5] */
    int32_t  *nbullets;           /* [N] */
    uint16_t *bullets;            /* [N][K] (entries >= nbullets read back as 0) */
    int32_t  *step_count;         /* [N] */
    double   *episode_return;     /* [N] */
    uint8_t  *prev_alive;         /* [N][2] */
    uint32_t *opp_ctr;            /* [N] */
    uint32_t *mt;                 /* [N][624] MT19937 words */
    int32_t  *mt_index;           /* [N] 0..624 */
} inv_state_view;

const char *inv_last_error(void);
int  inv_abi_version(void);

/* MultiEnvRunner.__init__ (env_wrappers.py:450-469) minus the seeding/reset. */
int  inv_create(const inv_config *cfg, inv_handle **out);
int  inv_destroy(inv_handle *h);
int  inv_get_config(const inv_handle *h, inv_config *cfg);
/* Which kernel runs FULL, unmasked steps (inv_step with INV_STEP_DEFAULT, inv_rollout).  Results are identical; this is a
 * speed knob and the hook the parity tests use to run every case through both.
 *   AUTO  four envs per wavefront (csrc/inv_step4.h) where it exists and measured faster or equal (15x10 boards; 32x32
 *         boards with packed observations; max_bullets = 16), else WAVE
 *   WAVE  one env per wavefront (csrc/inv_device.h); always used for partial-phase and masked steps
 *   ROW4  as AUTO, but an error where the four-env kernel does not exist */
enum { INV_STEP_KERNEL_AUTO = 0, INV_STEP_KERNEL_WAVE = 1, INV_STEP_KERNEL_ROW4 = 2 };
int  inv_set_step_kernel(inv_handle *h, int32_t kind);
/* Cache policy of the f32 observation stores (inv_step / inv_rollout / inv_render_obs).  Results are identical; speed only.
 *   AUTO   write-through (`sc1`) stores for launches whose every observation store instruction covers whole 128-B lines --
 *          the observation pointers are 128-B aligned and either the four-env kernel writes a wave's four 15x10 regions as one
 *          28 800-B stream (num_envs a multiple of 4) or one region is a multiple of 128 B (W*H a multiple of 8, e.g. 32x32) --
 *          plain stores otherwise: a partially written line costs a masked write at memory when it is written through
 *   PLAIN  never write through (the round-1/2 behaviour; A/B runs) */
enum { INV_OBS_STORE_AUTO = 0, INV_OBS_STORE_PLAIN = 1 };
int  inv_set_obs_store_policy(inv_handle *h, int32_t policy);
/* bytes of one env's obs / words per grid, for sizing caller buffers */
int64_t inv_obs_floats_per_env(const inv_handle *h);   /* 12*H*W */
int32_t inv_grid_words(const inv_handle *h);           /* GW */
int32_t inv_obs_words_per_env(const inv_handle *h);    /* ceil(12*H*W / 32): u32 words of one packed observation */

/* random.Random(seed) per env (core.py:41,58): CPython's init_by_array on the 32-bit
 * words of abs(seed).  seeds_host: N int64 in HOST memory.  Also zeroes opp_ctr. */
int  inv_seed(inv_handle *h, const int64_t *seeds_host, void *stream);
/* Same, only for envs whose mask_host byte is non-zero (SingleInversusRLEnv.reset(seed=...) on one env). */
int  inv_seed_masked(inv_handle *h, const int64_t *seeds_host, const uint8_t *mask_host, void *stream);

/* SingleInversusRLEnv.reset() (env_wrappers.py:272-284) -> InversusEnv.reset()
 * (core.py:55-154) on every env whose mask byte is non-zero (mask NULL = all):
 * continues each env's own MT stream. */
int  inv_reset(inv_handle *h, const uint8_t *mask, void *stream);

/* MultiEnvRunner.step (env_wrappers.py:485-528) = per env discrete_to_action +
 * InversusEnv.step_players (core.py:497-531) + reward/done/info (:318-442) +
 * build_observation (:173-245); with INV_STEP_AUTO_RESET also the training loop's
 * reset-and-patch (training.py:148-151).  a2 NULL -> device dummy opponent. */
int  inv_step(inv_handle *h, const uint8_t *a1, const uint8_t *a2, uint32_t flags,
              const inv_step_out *out, void *stream);
/* Same, stepping only envs whose (device) mask byte is non-zero; the others keep their state
 * and their output slots are not written (SingleInversusRLEnv.step on runner.envs[i]). */
int  inv_step_masked(inv_handle *h, const uint8_t *mask, const uint8_t *a1, const uint8_t *a2, uint32_t flags,
                     const inv_step_out *out, void *stream);

/* dummy_opponent_policy(env, difficulty) (env_wrappers.py:69-170) for every env of the handle, WITHOUT stepping it (ABI 8):
 * actions[i] (device, uint8_t[N]) = the P2 action id (0..8: the dummy never charges) the device dummy takes for env i's current
 * state -- what inv_step with a2 == NULL would apply next when the handle's difficulty is `difficulty` (INV_EASY / INV_HARD; like
 * the reference's argument it need not be the handle's own).  The reference draws from the global `random`; the device dummy's
 * draws are a counter hash keyed by (opp_seed, global env index, steps taken against the dummy so far), which this call reads
 * and does not advance: calling it twice without a step in between returns the same actions. */
int  inv_dummy_actions(inv_handle *h, int32_t difficulty, uint8_t *actions, void *stream);

/* build_observation(env, player) (env_wrappers.py:173-245); player 0 = P1, 1 = P2. */
int  inv_render_obs(inv_handle *h, int32_t player, float *obs, float *extra, void *stream);
/* Same observation as a packed bitstring, bits: uint32_t[N][inv_obs_words_per_env] (see INV_STEP_OBS_BITS). */
int  inv_render_obs_bits(inv_handle *h, int32_t player, uint32_t *bits, float *extra, void *stream);
/* Packed -> f32 planes for n observations of floats_per_env = 12*H*W floats each (words per env =
 * ceil(floats_per_env / 32)); needs no handle, so a learner can expand minibatches of stored rollouts. */
int  inv_unpack_obs(const uint32_t *bits, int64_t n, int32_t floats_per_env, float *obs, void *stream);

/* State poke / dump, the SoA form of the reference tests' direct field access
 * (tests/test_bullet_collision.py etc. set env.bullets / env.player1.x / _set_tile). */
int  inv_get_state(inv_handle *h, const inv_state_view *dst, void *stream);
int  inv_set_state(inv_handle *h, const inv_state_view *src, void *stream);

/* Synchronise `stream` and return (then clear) the sticky device error:
 * INV_OK, INV_ERR_BAD_ACTION or INV_ERR_BULLET_OVERFLOW. */
int  inv_check(inv_handle *h, void *stream);

/* PPOAgent.compute_advantages (ppo_agent.py:127-157), strict f32, no FMA.
 * r, v, adv, ret: f32 [T][N]; done u8 [T][N]; gamma_lam = f32(f64 gamma*lam).
 * last_value: f32[N] (per_env; NULL = 0) or f32[1] (flat; NULL = 0). */
int  inv_gae(const float *r, const float *v, const uint8_t *done, int32_t T, int32_t N,
             float gamma, float gamma_lam, const float *last_value, int32_t layout,
             float *adv, float *ret, void *stream);

/* Synthetic action stream (bench stand-in for a policy): a[i] = Philox(seed; t, env_offset+i) mod 13. */
int  inv_uniform_actions(uint8_t *a, int32_t n, uint32_t seed, int64_t env_offset, uint32_t t, void *stream);

/* T vector steps in ONE launch with pre-supplied actions a1[T][N] (a2[T][N] or NULL),
 * writing the rollout buffers obs[T][N][12][H][W] ... -- the persistent multi-step form
 * of inv_step (always auto-resets).  Outputs as inv_step_out with a leading T dim. */
int  inv_rollout(inv_handle *h, int32_t T, const uint8_t *a1, const uint8_t *a2, uint32_t flags /* 0 or INV_STEP_OBS_BITS */,
                 const inv_step_out *out, void *stream);

/* ---- fused policy trunk -------------------------------------------------
 * conv1..conv4 + LayerNorm([C,H,W]) + ReLU (+ residual) of InversusCNNPolicy.forward
 * (inversus_rl/policies.py:91-99) for the 10 x 15 board, bf16 MFMA / f32 accumulate, one workgroup
 * per sample.  Per layer l (Cin_pad = 32, 32, 64, 128; Cout = 32, 64, 128, 128):
 *   w    bf16 [Cout/16][9*Cin_pad/32][64][8]  A fragments of v_mfma_f32_16x16x32_bf16: lane = 16*g + r holds
 *                                             W[cout = 16*mt + r][k = 32*ks + 8*g + j], k = (3*ky+kx)*Cin_pad + c
 *   bias f32  [Cout]
 *   ln   fp16 [150][Cout/4][8]                LayerNorm affine per (pixel, 4 consecutive channels): {weight x4, bias x4}
 *                                             (fp16: 11 significant bits, 8x finer than the bf16 activations it scales)
 * obs: f32 [N][12][10][15] (what inv_step writes); feat: bf16 [N][150][128] channels-last, i.e.
 * feature index pixel*128 + channel (the caller permutes the first Linear's columns accordingly). */
typedef struct { const void *w; const float *bias; const void *ln; } inv_policy_layer;
typedef struct { inv_policy_layer layer[4]; } inv_policy_params;
int  inv_policy_trunk(const inv_policy_params *params, const float *obs, void *feat_bf16, int32_t N, void *stream);
/* Same trunk reading packed observations, bits: uint32_t[N][57] (INV_STEP_OBS_BITS layout of the 15 x 10 board):
 * 228 B instead of 7 200 B per sample, the next sample's words are fetched while the current one computes. */
int  inv_policy_trunk_bits(const inv_policy_params *params, const uint32_t *bits, void *feat_bf16, int32_t N, void *stream);

/* The trunk for boards whose activations do not fit LDS (BASELINE config C5: 32 x 32), same parameter packing with H*W pixels
 * (ln: fp16 [H*W][Cout/4][8]); obs: f32 [N][12][H][W] or packed uint32_t [N][ceil(12*H*W/32)] (obs_is_bits); feat: bf16
 * [N][H*W][128].  Per sample and layer the board is processed in bands of 4 rows; the layers' pre-norm outputs stream through
 * `scratch` (inv_policy_trunk_banded_scratch_bytes(H, W) bytes of device memory, reused by every launch on a stream; 0 =
 * board not supported).  Built for 32 x 32. */
int64_t inv_policy_trunk_banded_scratch_bytes(int32_t height, int32_t width);
/* The banded trunk's TRAINING path (BASELINE config C5 with the fused update; reference: PPOAgent.update over
 * policies.py:91-99 at H = W = 32).  The forward additionally keeps, per sample, the four layers' pre-norm outputs z
 * (bf16 channels-last [N][H*W][32 | 64 | 128 | 128]) and the LayerNorm statistics (stats: f32 [N][4][2] = rstd, -mean*rstd);
 * the backward then runs layer by layer (l = 3 .. 0) over the whole batch:
 *   inv_policy_band_ln_bwd(l)       da_l (for l = 3: d loss / d feat) -> dz_l (LayerNorm + ReLU backward; the mask is recomputed from z_l),
 *                                   optionally dy_l = da_l * [mask] and a_l = relu(LN(z_l)) (the `x` operand of inv_policy_band_wgrad(l + 1))
 *   inv_policy_band_conv_t(l)       l >= 1: da_{l-1} = convT(dz_l) (+ dz_3 for the residual); wt_packed =
 *                                   inv_policy_pack_conv(transposed = 1) of layer l, zero_bias: f32 zeros [128]
 *   inv_policy_band_act(l)          a_l = relu(LN(z_l)) as a tensor of its own (the training path takes it from inv_policy_band_ln_bwd)
 *   inv_policy_band_wgrad(l)        dW_l (f32 [Cout][9][CinP], ADDED) and dbias from x = a_{l-1} (l = 0: inv_policy_obs_to_x0_any) and dz_l
 *   inv_policy_band_param_grads(l)  per-slice partial sums [slices][2][H*W*C] of dy x^ and dy (LayerNorm weight / bias), recomputed from da_l, z_l
 * All tensors bf16 channels-last [N][H*W][C] unless noted.  Built for 32 x 32. */
typedef struct { void *z[4]; float *stats; } inv_policy_band_saved;
int  inv_policy_trunk_banded_train_fwd(const inv_policy_params *params, const void *obs, int32_t obs_is_bits, int32_t height, int32_t width,
                                       void *feat_bf16, const inv_policy_band_saved *saved, int32_t N, void *stream);
int  inv_policy_band_ln_bwd(int32_t layer, int32_t height, int32_t width, const void *da_bf16, const void *z_bf16, const float *stats,
                            const void *ln_packed, void *dy_bf16 /* may be NULL */, void *dz_bf16, void *act_bf16 /* may be NULL */,
                            int32_t N, void *stream);
/* Round 4: the same layer step in two tensor passes instead of four, with the LayerNorm parameter sums on the way (what the 32 x 32
 * training path calls): inv_policy_band_ln_sums writes sums f32 [N][2] (sum of dx^, sum of dx^ x^ per sample: pass A of
 * inv_policy_band_ln_bwd alone); inv_policy_band_ln_bwd2 turns (da, z, sums) into dz, the activation a = relu(LN(z)) (act may be
 * NULL) and partial f32 [slices][2][P * C] (per slice of samples n = s, s + slices, ...: sums of dy x^ and of dy -- the output of
 * inv_policy_band_param_grads).  dz, act and the partial sums are bit-identical to the four-pass path's. */
int  inv_policy_band_ln_sums(int32_t layer, int32_t height, int32_t width, const void *da_bf16, const void *z_bf16, const float *stats,
                             const void *ln_packed, float *sums, int32_t N, void *stream);
int  inv_policy_band_ln_bwd2(int32_t layer, int32_t height, int32_t width, const void *da_bf16, const void *z_bf16, const float *stats,
                             const void *ln_packed, const float *sums, void *dz_bf16, void *act_bf16, float *partial, int32_t slices,
                             int32_t N, void *stream);
int  inv_policy_band_conv_t(int32_t layer, int32_t height, int32_t width, const void *wt_packed, const float *zero_bias,
                            const void *dz_bf16, void *da_below_bf16, int32_t N, void *stream);
/* ABI 7: inv_policy_band_conv_t(l) that also leaves, per sample, the two LayerNorm-backward sums of layer l - 1 (sums_below f32 [N][2],
 * what inv_policy_band_ln_sums(l - 1) computes from da_{l-1}: here from the chunks of da_{l-1} as they are written, z_{l-1}, stats and
 * layer l - 1's packed LayerNorm parameters; same expressions, another order of the f32 additions) -- one read of da_{l-1} and one
 * launch less per layer (fused_train.py's banded backward; INVERSUS_BAND_FUSED_SUMS=0 keeps the separate pass). */
int  inv_policy_band_conv_t_sums(int32_t layer, int32_t height, int32_t width, const void *wt_packed, const float *zero_bias,
                                 const void *dz_bf16, void *da_below_bf16, const void *z_below_bf16, const float *stats,
                                 const void *ln_below_packed, float *sums_below, int32_t N, void *stream);
int  inv_policy_band_act(int32_t layer, int32_t height, int32_t width, const void *z_bf16, const float *stats, const void *ln_packed,
                         void *act_bf16, int32_t N, void *stream);
int  inv_policy_band_wgrad(int32_t layer, int32_t height, int32_t width, const void *x_bf16, const void *dz_bf16, float *dw, float *dbias,
                           int32_t N, void *stream);
int  inv_policy_band_param_grads(int32_t layer, const void *da_bf16, const void *z_bf16, const float *stats, const void *ln_packed,
                                 float *partial, int32_t slices, int32_t elems_per_sample, int32_t N, void *stream);
int  inv_policy_obs_to_x0_any(const void *obs, int32_t obs_is_bits, int32_t pixels, void *x0_bf16, int32_t N, void *stream);
int  inv_policy_trunk_banded(const inv_policy_params *params, const void *obs, int32_t obs_is_bits, int32_t height, int32_t width,
                             void *feat_bf16, void *scratch, int64_t scratch_bytes, int32_t N, void *stream);

/* ---- the same trunk for the PPO update: forward that keeps what the backward needs, and the
 * backward of LayerNorm/ReLU + the data gradient of the convolutions (ppo_agent.py:219-224 runs autograd over
 * policies.py:91-99).  All tensors bf16 channels-last [N][150][C_l], C = 32, 64, 128, 128:
 *   xhat[l]  normalised pre-activation (z - mean) * rstd of layer l        act[l]  ReLU output of layer l (l < 3; act 3 = feat)
 *   rstd     f32 [N][4]
 *   dy[l]    gradient wrt the LayerNorm output (after the ReLU mask): dln_w = sum_n dy * xhat, dln_b = sum_n dy
 *   dz[l]    gradient wrt the convolution output: dbias = sum dz, dW = sum_n sum_p dz[p] (x) input[p + tap]
 * The parameter gradients are reductions over the batch of these tensors (left to the caller's GEMM / reduction library).
 * wt[l - 1] (l = 1..3): the weights of layer l transposed for the data gradient, packed like inv_policy_layer.w for a
 * convolution with Cin <-> Cout swapped and the taps flipped: Wt[ci][8 - tap][co] = W[co][ci][tap]. */
typedef struct { void *xhat[4]; void *act[3]; float *rstd; } inv_policy_saved;
typedef struct { const void *wt[3]; const void *ln[4]; } inv_policy_bwd_params;
typedef struct { void *dy[4]; void *dz[4]; } inv_policy_grads;
/* saved == NULL: the same numerics (x^ rounded to bf16 before the affine) with nothing stored -- the rollout forward to use
 * when inv_policy_trunk_bwd-based updates follow, so that the update's first forward reproduces the rollout's features
 * BIT FOR BIT (tests/test_gpu_fused_train.py); it runs the inference kernel's epilogue, at the inference kernel's speed. */
int  inv_policy_trunk_train_fwd(const inv_policy_params *params, const void *obs, int32_t obs_is_bits, void *feat_bf16,
                                const inv_policy_saved *saved, int32_t N, void *stream);
/* grads->dy[3] may be NULL (ABI 8): the top layer's dy = [x^ g + b > 0] * dfeat is then not written, and
 * inv_policy_param_grads_all2 recomputes it from dfeat (every other member is required). */
int  inv_policy_trunk_bwd(const inv_policy_bwd_params *params, const inv_policy_saved *saved, const void *dfeat_bf16,
                          const inv_policy_grads *grads, int32_t N, void *stream);
/* Weight (and bias) gradient of convolution `layer` (0..3): dw[co][tap][ci] += sum_n sum_p dz[n][p][co] * x[n][p + tap][ci],
 * f32 [Cout][9][CinP], tap = 3*ky + kx; dbias[co] += sum_n sum_p dz[n][p][co] (f32 [Cout], may be NULL).  Both are
 * accumulated into: zero them first.  x: the layer's input, bf16 channels-last [N][150][CinP] with CinP = 16, 32, 64, 128
 * (layer 0: the 12 observation planes padded with 4 zero channels); dz: what inv_policy_trunk_bwd wrote.
 * MFMA over pixels through transposing LDS reads. */
int  inv_policy_wgrad(int32_t layer, const void *x_bf16, const void *dz_bf16, float *dw, float *dbias, int32_t N, void *stream);
/* The four layers' dw blocks ([Cout][9][CinP], back to back in `src`: 32x9x16, 64x9x32, 128x9x64, 128x9x128 floats) -> the module's
 * tensors dw[l] f32 [Cout][Cin][3][3] (assigned), one launch. */
typedef struct { float *dw[4]; } inv_policy_conv_grads;
int  inv_policy_wgrad_to_module(const float *src, const inv_policy_conv_grads *out, void *stream);
/* All four layers' LayerNorm weight / bias gradients in one launch, ADDED into the module's tensors gw[l], gb[l]: f32 [C_l][10][15]
 * (zeroed by the caller); dy from inv_policy_grads, x^ from inv_policy_saved; `slices` sample slices (64 is a good value). */
typedef struct { float *gw[4]; float *gb[4]; } inv_policy_ln_grads;
int  inv_policy_param_grads_all(const inv_policy_grads *grads, const inv_policy_saved *saved, const inv_policy_ln_grads *out,
                                int32_t slices, int32_t N, void *stream);
/* ABI 8: the same launch WITHOUT a dy tensor for the top layer.  inv_policy_trunk_bwd accepts grads->dy[3] == NULL and then does not
 * write it (0.63 GB per 16 384 samples on 15 x 10); this entry point, given dfeat_bf16 (what the backward kernel was given) and
 * ln_top_packed (inv_policy_bwd_params.ln[3]), recomputes dy_3 = [x^ g + b > 0] * dfeat per element -- the backward kernel's own
 * expression, so the sums are bit-identical to those from a stored dy[3].  dfeat_bf16 and ln_top_packed: both or neither (neither =
 * inv_policy_param_grads_all). */
int  inv_policy_param_grads_all2(const inv_policy_grads *grads, const inv_policy_saved *saved, const inv_policy_ln_grads *out,
                                 const void *dfeat_bf16, const void *ln_top_packed, int32_t slices, int32_t N, void *stream);
/* Batch reductions for the LayerNorm gradients of one layer, one pass over dy and xhat (bf16 [N][E], E = 150 * C):
 * partial[s][0][e] = sum dy*xhat, partial[s][1][e] = sum dy over the samples n = s (mod slices); f32 [slices][2][E], every
 * entry written.  Adding the slices gives the LayerNorm weight / bias gradients in [pixel][channel] order. */
int  inv_policy_param_grads(const void *dy_bf16, const void *xhat_bf16, float *partial, int32_t slices,
                            int32_t elems_per_sample, int32_t N, void *stream);

/* Host glue of the training path as kernels: the parameters change with every optimizer step, so the packed forms above are
 * rebuilt per minibatch.
 *   inv_policy_pack_conv  f32 [Cout][Cin][3][3] -> the bf16 A fragments of inv_policy_layer.w (transposed = 0; Cin padded to cin_pad)
 *                         or of inv_policy_bwd_params.wt (transposed = 1: Wt[ci][8 - tap][co] = W[co][ci][tap]; cin_pad unused)
 *   inv_policy_pack_ln    LayerNorm([C,H,W]) weight / bias f32 -> fp16 [H*W][C/4][weight x4 | bias x4] (inv_policy_layer.ln); pixels = H*W
 *   inv_policy_obs_to_x0  observations (f32 planes [N][12][10][15] or packed [N][57], obs_is_bits) -> bf16 channels-last
 *                         [N][150][16], the layer-0 `x` operand of inv_policy_wgrad */
int  inv_policy_pack_conv(const float *w, int32_t cout, int32_t cin, int32_t cin_pad, int32_t transposed, void *out_bf16, void *stream);
int  inv_policy_pack_ln(const float *weight, const float *bias, int32_t cout, int32_t pixels, void *out_f16, void *stream);
int  inv_policy_obs_to_x0(const void *obs, int32_t obs_is_bits, void *x0_bf16, int32_t N, void *stream);

/* ---- the first Linear of the policy heads ("fused heads GEMM [B x 19 204] . [19 204 x 512]") ------------
 * Reference: inversus_rl/policies.py:60-75 (fc_actor[0], fc_critic[0] = Linear(128 P + 4, 256)), :102-106 (features ++ extra), and
 * their autograd derivatives in ppo_agent.py:211-232.  bf16 operands, f32 accumulation, hand-written MFMA kernels (csrc/inv_fc1.hip);
 * feature index = p * 128 + c (the trunk kernels' channels-last order).  The 4 extra columns / bias / ReLU are inv_heads_fwd's.
 *   inv_fc1_fwd    pre = feat [B][K] @ wf [nout][K]^T, K = 128 P, nout = 256 nheads, written as inv_fc1_fwd_slabs(B, nout, K) f32
 *                  split-K partial sums pre_slabs [slabs][B][nout] (inv_heads_fwd adds them in slab order).
 *   inv_fc1_dfeat  dfeat bf16 [B][nfeat] = dpre [B][nhid] @ wft [nfeat][nhid]^T   (wft = inv_transpose_bf16(wf))
 *   inv_fc1_dw     weight gradient in the MODULE's layout: grad_w_actor / grad_w_critic f32 [256][128 P + 4] (overwritten), column
 *                  c * P + p (nn.Flatten's order) = sum_b dpre_t[j][b] feat[b][p * 128 + c], last 4 columns = dwe [nhid][4].
 *                  dpre_t bf16 [nhid][ld], ld = B rounded up to a multiple of 64, columns B .. ld - 1 ZERO (inv_heads_bwd writes it).
 *                  workspace: inv_fc1_dw_workspace_bytes(B, nhid, 128 P) bytes (stream-K segment sums).  nhid = 512 (256: actor only,
 *                  grad_w_critic unused). */
int32_t inv_fc1_fwd_slabs(int32_t B, int32_t nout, int32_t K);
int  inv_fc1_fwd(const void *feat_bf16, const void *wf_bf16, int32_t B, int32_t nout, int32_t K, float *pre_slabs, void *stream);
int  inv_fc1_dfeat(const void *dpre_bf16, const void *wft_bf16, int32_t B, int32_t nhid, int32_t nfeat, void *dfeat_bf16, void *stream);
int64_t inv_fc1_dw_workspace_bytes(int32_t B, int32_t nhid, int32_t nfeat);
int  inv_fc1_dw(const void *dpre_t_bf16, const void *feat_bf16, const float *dwe, int32_t B, int32_t nhid, int32_t pixels,
                float *grad_w_actor, float *grad_w_critic, void *workspace, void *stream);

/* ---- the policy heads after their first Linear ------------------------------------------
 * Reference: inversus_rl/policies.py:60-75 (fc_actor, fc_critic = Linear(19204, 256) ReLU Linear(256, 128) ReLU Linear(128, 13 | 1))
 * and :102-106.  The first Linear's big product  pre = features @ W1[:, :128 P]^T  (f32 [B][256 nheads], columns 0..255 actor,
 * 256..511 critic) is inv_fc1_fwd's; these entry points are everything after it, forward and backward, one kernel each:
 *     h = relu(pre + extra W1[:, 128 P:]^T + b1)      a2 = relu(W2 h + b2)      logits | value = W3 a2 + b3
 * h and a2 are rounded to bf16, W2 / W3 are bf16 MFMA operands, accumulation / biases / the extra columns are f32.
 * Packed operands (inv_heads_pack): MFMA A fragments of a [rows x cols] matrix padded to [mpad x kpad] (multiples of 16 / 32),
 *     out[(mt * (kpad / 32) + ks) * 64 + lane][e] = M[mt * 16 + (lane & 15)][ks * 32 + (lane >> 4) * 8 + e]     (bf16, zero padding)
 *   w2[h]  = W2 of head h            [128 x 256]          w2t[h] = its transpose [256 x 128]              (h = 0 actor, 1 critic)
 *   w3[h]  = W3 of head h padded to  [16 x 128]           w3t[h] = its transpose [128 x 32]  (13 | 1 columns used)
 *   we f32 [256 nheads][4] = W1[:, 128 P:],  b1 f32 [256 nheads],  b2[h] f32 [128],  b3[h] f32 [13 | 1].
 * w2t / w3t are only read by inv_heads_bwd. */
#define INV_HEADS_HID 256
#define INV_HEADS_MID 128
#define INV_HEADS_ACTIONS 13
#define INV_HEADS_PACK_MAX 8
typedef struct {
    const void *w2[2], *w2t[2], *w3[2], *w3t[2];
    const float *we, *b1, *b2[2], *b3[2];
} inv_heads_params;
/* one matrix to pack: out = fragments of M [rows x cols], M[r][k] = transposed ? src[k * ld + r] : src[r * ld + k] (src f32) */
typedef struct { const float *src; void *dst; int32_t rows, cols, ld, transposed, mpad, kpad; } inv_heads_pack_item;
typedef struct { inv_heads_pack_item item[INV_HEADS_PACK_MAX]; } inv_heads_pack_list;
int  inv_heads_pack(const inv_heads_pack_list *list, int32_t count, void *stream);                  /* one launch for `count` matrices */
/* First Linear, `rows` output rows: w f32 [rows][128 P + 4] (feature index c * P + p, nn.Flatten's order) -> wf bf16 [rows][128 P]
 * in the trunk kernels' channels-last order p * 128 + c, and we f32 [rows][4] (the way back is inv_fc1_dw's epilogue). */
int  inv_heads_pack_fc1(const float *w, int32_t rows, int32_t pixels, void *wf_bf16, float *we, void *stream);
/* bf16 [rows][cols] -> [cols][rows]: the K-contiguous copy of the first Linear's weights that inv_fc1_dfeat reads */
int  inv_transpose_bf16(const void *in_bf16, int32_t rows, int32_t cols, void *out_bf16, void *stream);
/* Forward.  pre_slabs f32 [nslabs][B][256 nheads] = inv_fc1_fwd's output (nslabs >= 1).  nheads = 2: actor and critic; nheads = 1:
 * actor only (value may be NULL) -- the frozen self-play opponent.  logits f32 [B][13], value f32 [B].  h_save bf16 [B][256 nheads]
 * and a2_save bf16 [B][128 nheads]: both or neither; what inv_heads_bwd needs.  For a fixed slab count the result for a sample does
 * not depend on B or on its position in the batch. */
int  inv_heads_fwd(const inv_heads_params *params, const float *pre_slabs, int32_t nslabs, const float *extra, float *logits, float *value,
                   void *h_save, void *a2_save, int32_t nheads, int32_t B, void *stream);
/* Backward (both heads).  In: dlogits f32 [B][13], dvalue f32 [B], the saved h / a2, extra f32 [B][4].  Out: dpre bf16 [B][512]
 * (= d loss / d pre, the operand of inv_fc1_dfeat), optionally (non-NULL) its transpose dpre_t bf16 [512][dpre_t_ld] (dpre_t_ld >= B, a
 * multiple of 8; inv_fc1_dw wants B rounded up to 64 with the tail zeroed by the caller), da2 bf16 [B][256] (with h: the operands
 * of inv_heads_dw2), and ADDED into the f32 buffers of `grads` (zero them first):
 * db1 [512], dwe [512][4], db2[h] [128], dw3[h] [13 | 1][128], db3[h] [13 | 1]. */
typedef struct { float *db1, *dwe, *db2[2], *dw3[2], *db3[2]; } inv_heads_grads;
int  inv_heads_bwd(const inv_heads_params *params, const float *dlogits, const float *dvalue, const void *h, const void *a2,
                   const float *extra, void *dpre_bf16, void *dpre_t_bf16, int32_t dpre_t_ld, void *da2_bf16, const inv_heads_grads *grads,
                   int32_t B, void *stream);
/* dW2[h] f32 [128][256] += da2[:, 128 h : 128 h + 128]^T h_saved[:, 256 h : 256 h + 256] (zero the two buffers first).
 * Determinism: the partial sums of the 256-sample chunks (B / 256 per head) are ADDED with f32 atomics, like db1 / dwe / db2 / dw3 / db3
 * of inv_heads_bwd: these gradients are equal between launches to f32 reassociation (~1e-7 relative), NOT bit for bit -- unlike
 * inv_fc1_fwd / _dfeat / _dw (fixed-order slab sums, no atomics) and unlike the torch.bmm + sum(0) form this kernel replaced. */
int  inv_heads_dw2(const void *da2_bf16, const void *h_bf16, int32_t B, float *dw2_actor, float *dw2_critic, void *stream);

/* ---- the scalar tail of the PPO update ------------------------------------------------------
 * inv_ppo_loss: forward AND backward of the reference's minibatch loss (inversus_rl/ppo_agent.py:211-224)
 *     loss = mean(-min(ratio A, clamp(ratio, 1-c, 1+c) A)) + value_coef mean((v - R)^2) - entropy_coef mean(entropy)
 * over B samples with 13 logits each, one kernel: dlogits f32 [B][13] and dvalues f32 [B] receive d loss / d logits and
 * d loss / d values (what autograd returns for the same expression, including the 1/B of the means); partial_sums f32
 * [inv_ppo_loss_blocks(B)][3] receives per-block sums of the policy term, the squared value error and the entropy (add
 * them and divide by B for the three means the reference reports, ppo_agent.py:234-236).  actions: int64 [B]. */
int32_t inv_ppo_loss_blocks(int32_t B);
int  inv_ppo_loss(const float *logits, const float *values, const int64_t *actions, const float *old_log_probs,
                  const float *advantages, const float *returns, int32_t B, float clip_ratio, float value_coef,
                  float entropy_coef, float *dlogits, float *dvalues, float *partial_sums, void *stream);
/* inv_sample_actions: Categorical(logits).sample() and .log_prob(sample) of PPOAgent.act (inversus_rl/ppo_agent.py:96-100) for B
 * samples of 13 logits, one kernel.  The log-softmax uses the expressions of inv_ppo_loss, so the update's ratio
 * exp(new_logp - old_logp) is exactly 1 on identical logits.  Sampling: inverse CDF at one Philox4x32-10 uniform per sample,
 * counter = sample index, key / counter words from seed[0..1] (two int64 in DEVICE memory -- e.g. drawn from the framework's
 * generator, so that its seeding controls the stream without a host round trip).  Outputs (each may be NULL, at least one action
 * form required): actions int64 [B], actions_u8 uint8 [B] (the form inv_step reads), log_prob f32 [B]. */
int  inv_sample_actions(const float *logits, int32_t B, const int64_t *seed, int64_t *actions, uint8_t *actions_u8, float *log_prob,
                        void *stream);
/* inv_clip_adam_step: torch.nn.utils.clip_grad_norm_(params, max_grad_norm) followed by torch.optim.Adam.step()
 * (ppo_agent.py:227-232; no weight decay, no amsgrad) over ONE flat f32 buffer of n parameters with its gradient and
 * moment buffers (16-byte aligned), two launches, no host synchronisation:
 *     total = |grad_scale| * ||grads||_2;  coef = min(1, max_grad_norm / (total + 1e-6))   (max_grad_norm <= 0: no clipping)
 *     g = coef * grad_scale * grads;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
 *     params -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps)
 * grad_scale multiplies the gradient first (1/world_size after a SUM all-reduce; 1 otherwise).  `grads` is not modified.
 * step: 1 for the first call.  workspace: inv_clip_adam_workspace_bytes() bytes of device memory; after the call its last
 * 4 bytes hold `total` (the pre-clip norm clip_grad_norm_ returns) as f32 at byte offset workspace_bytes - 8. */
int64_t inv_clip_adam_workspace_bytes(void);
int  inv_clip_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                        float beta1, float beta2, float eps, int32_t step, float max_grad_norm, float grad_scale,
                        void *workspace, void *stream);

#ifdef __cplusplus
}
#endif
#endif
