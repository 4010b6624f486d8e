/*
 * inv_oracle.c -- CPU restatement of the reference's Inversus hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see inv_oracle.h).  Parity status: PINNED by
 * the tests/golden/ .npz fixtures (generated from the Python reference by
 * tests/golden/gen_golden.py) and by the reference's own known-answer tests.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; no fast-math -- the
 * reward is an f64 sum in a fixed order and GAE is strict f32).
 */
#include "inv_oracle.h"

#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* MT19937 as driven by CPython (Modules/_randommodule.c; the reference uses  */
/* it through random.Random at core.py:41,58,60,69-70,86-87)                  */
/* ------------------------------------------------------------------------ */

static void mt_init_genrand(uint32_t *mt, uint32_t s)
{
    mt[0] = s;
    for (int i = 1; i < 624; i++)
        mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}

static void mt_init_by_array(uint32_t *mt, const uint32_t *key, int key_len)
{
    mt_init_genrand(mt, 19650218u);
    int i = 1, j = 0;
    int k = 624 > key_len ? 624 : key_len;
    for (; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
        if (j >= key_len) j = 0;
    }
    for (k = 623; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
}

/* random.Random(seed) with an int seed: key = 32-bit little-endian words of abs(seed),
 * at least one word (random_seed() in _randommodule.c). */
void orc_mt_seed(orc_env *e, int64_t seed)
{
    uint64_t a = seed < 0 ? (uint64_t)0 - (uint64_t)seed : (uint64_t)seed;
    uint32_t key[2];
    int len = 1;
    key[0] = (uint32_t)(a & 0xffffffffu);
    key[1] = (uint32_t)(a >> 32);
    if (key[1]) len = 2;
    mt_init_by_array(e->mt, key, len);
    e->mti = 624;
}

uint32_t orc_mt_u32(orc_env *e)
{
    static const uint32_t mag01[2] = { 0u, 0x9908b0dfu };
    uint32_t *mt = e->mt, y;
    if (e->mti >= 624) {
        int kk;
        for (kk = 0; kk < 624 - 397; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1u];
        }
        for (; kk < 623; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1u];
        }
        y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ mag01[y & 1u];
        e->mti = 0;
    }
    y = mt[e->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* Random._randbelow_with_getrandbits: k = n.bit_length(); r = getrandbits(k) until r < n;
 * getrandbits(k<=32) = genrand_uint32() >> (32 - k). */
int orc_randbelow(orc_env *e, int n)
{
    int k = 0;
    for (unsigned t = (unsigned)n; t; t >>= 1) k++;
    uint32_t r = orc_mt_u32(e) >> (32 - k);
    while (r >= (uint32_t)n) r = orc_mt_u32(e) >> (32 - k);
    return (int)r;
}

/* ------------------------------------------------------------------------ */
/* core.py                                                                    */
/* ------------------------------------------------------------------------ */

static int in_bounds(const orc_env *e, int x, int y)            /* core.py:222-224 */
{
    return 0 <= x && x < e->W && 0 <= y && y < e->H;
}

static const int DX[4] = { 0, 1, 0, -1 };   /* UP, RIGHT, DOWN, LEFT */
static const int DY[4] = { -1, 0, 1, 0 };

/* P1's colour is BLACK, P2's is WHITE (config.py:9,11; core.py:76,128). */
static int player_color(int pid) { return pid == 0 ? ORC_BLACK : ORC_WHITE; }

static void paint_plus(orc_env *e, int cx, int cy, int color)
{
    const int px[5] = { cx, cx + 1, cx - 1, cx, cx };
    const int py[5] = { cy, cy, cy, cy + 1, cy - 1 };
    for (int i = 0; i < 5; i++)
        if (in_bounds(e, px[i], py[i])) e->grid[py[i] * e->W + px[i]] = (uint8_t)color;
}

int orc_init(orc_env *e, int W, int H, int max_episode_steps)
{
    if (W < 3 || H < 3 || W > ORC_MAX_DIM || H > ORC_MAX_DIM) return -1;  /* randint(1, W-2) empty range, core.py:69-70 */
    memset(e, 0, sizeof(*e));
    e->W = W; e->H = H;
    e->max_episode_steps = max_episode_steps;
    e->mti = 624;
    e->prev_alive[0] = e->prev_alive[1] = 1;
    return 0;
}

void orc_core_reset(orc_env *e)
{
    /* config.py:20-56 make_initial_grid: all player colour (BLACK), WHITE plus at (1,1) */
    memset(e->grid, ORC_BLACK, (size_t)(e->W * e->H));
    if (in_bounds(e, 1, 1)) paint_plus(e, 1, 1, ORC_WHITE);

    /* core.py:69-70 */
    int p1x = 1 + orc_randbelow(e, e->W - 2);
    int p1y = 1 + orc_randbelow(e, e->H - 2);
    /* core.py:85-90 -- the last try is kept even if it is too close */
    int p2x = 0, p2y = 0;
    for (int t = 0; t < 20; t++) {
        p2x = 1 + orc_randbelow(e, e->W - 2);
        p2y = 1 + orc_randbelow(e, e->H - 2);
        int dist = abs(p2x - p1x) + abs(p2y - p1y);
        if (dist > 4) break;
    }
    paint_plus(e, p2x, p2y, ORC_WHITE);   /* core.py:96-108 */
    paint_plus(e, p1x, p1y, ORC_WHITE);   /* core.py:112-121 */
    paint_plus(e, p2x, p2y, ORC_BLACK);   /* core.py:136-146 (wins overlaps) */

    e->p[0] = (orc_player){ p1x, p1y, 6, 6, 0, 1 };
    e->p[1] = (orc_player){ p2x, p2y, 6, 6, 0, 1 };
    e->nb = 0;                             /* core.py:154 */
}

int orc_try_move(orc_env *e, int pid, int dir)
{
    orc_player *p = &e->p[pid];
    if (!p->alive) return 0;
    int nx = p->x + DX[dir], ny = p->y + DY[dir];
    if (!in_bounds(e, nx, ny)) return 0;
    if (e->grid[ny * e->W + nx] == player_color(pid)) return 0;   /* core.py:238-247 */
    p->x = nx; p->y = ny;
    return 1;
}

int orc_spawn_bullet(orc_env *e, int pid, int dir)
{
    orc_player *p = &e->p[pid];
    if (!p->alive) return 0;
    if (p->ammo <= 0) return 0;
    p->ammo -= 1;
    if (e->nb < ORC_MAX_BULLETS) e->bullets[e->nb++] = (orc_bullet){ p->x, p->y, dir, pid };
    return 1;
}

int orc_spawn_wide(orc_env *e, int pid, int dir)
{
    orc_player *p = &e->p[pid];
    if (!p->alive) return 0;
    if (p->ammo < 3) return 0;
    p->ammo -= 3;
    int xs[3], ys[3];
    if (dir == ORC_UP || dir == ORC_DOWN) {            /* core.py:357-363 */
        xs[0] = p->x; xs[1] = p->x - 1; xs[2] = p->x + 1;
        ys[0] = ys[1] = ys[2] = p->y;
    } else {                                           /* core.py:364-370 */
        xs[0] = xs[1] = xs[2] = p->x;
        ys[0] = p->y; ys[1] = p->y - 1; ys[2] = p->y + 1;
    }
    int spawned = 0;
    for (int i = 0; i < 3; i++)
        if (in_bounds(e, xs[i], ys[i])) {
            if (e->nb < ORC_MAX_BULLETS) e->bullets[e->nb++] = (orc_bullet){ xs[i], ys[i], dir, pid };
            spawned++;
        }
    return spawned > 0;
}

void orc_reload(orc_env *e)
{
    for (int i = 0; i < 2; i++) {
        orc_player *p = &e->p[i];
        if (!p->alive) continue;
        if (p->ammo < p->max_ammo) {
            p->reload += 1;
            if (p->reload >= 30) { p->ammo += 1; p->reload = 0; }
        }
    }
}

void orc_update_bullets(orc_env *e)
{
    /* phase 1 (core.py:412-435): move, drop out-of-bounds, group by target tile in
     * first-arrival order (Python dicts keep insertion order). */
    static _Thread_local int tile_of_group[ORC_MAX_BULLETS];
    static _Thread_local int first_of_group[ORC_MAX_BULLETS];
    static _Thread_local int owners_of_group[ORC_MAX_BULLETS];   /* bit0: P1 present, bit1: P2 present */
    static _Thread_local orc_bullet moved[ORC_MAX_BULLETS];
    int ngroups = 0, nmoved = 0;

    for (int i = 0; i < e->nb; i++) {
        orc_bullet b = e->bullets[i];
        int nx = b.x + DX[b.dir], ny = b.y + DY[b.dir];
        if (!in_bounds(e, nx, ny)) continue;
        b.x = nx; b.y = ny;
        moved[nmoved] = b;
        int tile = ny * e->W + nx, g;
        for (g = 0; g < ngroups; g++) if (tile_of_group[g] == tile) break;
        if (g == ngroups) {
            tile_of_group[g] = tile; first_of_group[g] = nmoved; owners_of_group[g] = 0;
            ngroups++;
        }
        owners_of_group[g] |= 1 << b.owner;
        nmoved++;
    }

    /* phase 2 (core.py:440-473) */
    int nnew = 0;
    for (int g = 0; g < ngroups; g++) {
        if (owners_of_group[g] == 3) continue;                 /* mixed owners: all vanish, no flip, no hit */
        orc_bullet b = moved[first_of_group[g]];              /* bullets_here[0] */
        int tile = tile_of_group[g];
        if (e->grid[tile] == player_color(b.owner))           /* flip only the owner's colour, :458-461 */
            e->grid[tile] = (uint8_t)(1 - e->grid[tile]);
        for (int pid = 0; pid < 2; pid++)                     /* :464-470 */
            if (e->p[pid].alive && b.owner != pid && b.x == e->p[pid].x && b.y == e->p[pid].y)
                e->p[pid].alive = 0;
        e->bullets[nnew++] = b;                               /* :473 */
    }
    e->nb = nnew;
}

int orc_round_over(const orc_env *e) { return !(e->p[0].alive && e->p[1].alive); }

int orc_winner(const orc_env *e)
{
    if (!orc_round_over(e)) return 0;
    if (!e->p[0].alive && e->p[1].alive) return 2;
    if (!e->p[1].alive && e->p[0].alive) return 1;
    return 0;
}

static void apply_action(orc_env *e, int pid, int a)
{
    if (!e->p[pid].alive || a == 0) return;
    int type = (a + 3) >> 2, dir = (a - 1) & 3;       /* env_wrappers.py:20-66 */
    if (type == ORC_MOVE) orc_try_move(e, pid, dir);
    else if (type == ORC_SHOOT) orc_spawn_bullet(e, pid, dir);
    else orc_spawn_wide(e, pid, dir);
}

int orc_step_players(orc_env *e, int a1, int a2)
{
    if (a1 < 0 || a1 > 12 || a2 < 0 || a2 > 12) return -1;    /* ValueError, env_wrappers.py:66 */
    apply_action(e, 0, a1);       /* core.py:510-516 */
    apply_action(e, 1, a2);       /* core.py:519-525 */
    orc_reload(e);                /* :528 */
    orc_update_bullets(e);        /* :531 */
    return 0;
}

/* ------------------------------------------------------------------------ */
/* env_wrappers.py                                                            */
/* ------------------------------------------------------------------------ */

void orc_build_obs(const orc_env *e, int pid, float *g, float *extra)
{
    const int W = e->W, H = e->H, WH = W * H;
    memset(g, 0, sizeof(float) * 12u * (size_t)WH);
    const orc_player *me = &e->p[pid], *en = &e->p[1 - pid];
    for (int i = 0; i < WH; i++) {                                   /* :200-205, absolute colours */
        if (e->grid[i] == ORC_BLACK) g[0 * WH + i] = 1.0f; else g[1 * WH + i] = 1.0f;
    }
    if (me->alive) g[2 * WH + me->y * W + me->x] = 1.0f;              /* :208-209 */
    if (en->alive) g[3 * WH + en->y * W + en->x] = 1.0f;              /* :212-213 */
    for (int i = 0; i < e->nb; i++) {                                 /* :224-235 */
        const orc_bullet *b = &e->bullets[i];
        if (!in_bounds(e, b->x, b->y)) continue;
        int base = (b->owner == pid) ? 4 : 8;
        g[(base + b->dir) * WH + b->y * W + b->x] = 1.0f;
    }
    extra[0] = me->alive ? (float)((double)me->ammo / 6.0) : 0.0f;    /* :238-243, f64 divide then f32 */
    extra[1] = en->alive ? (float)((double)en->ammo / 6.0) : 0.0f;
    extra[2] = me->alive ? 1.0f : 0.0f;
    extra[3] = en->alive ? 1.0f : 0.0f;
}

void orc_wrapper_reset(orc_env *e)
{
    orc_core_reset(e);
    e->step_count = 0;
    e->episode_return = 0.0;
    e->prev_alive[0] = e->prev_alive[1] = 1;
}

static int count_white(const orc_env *e)
{
    int c = 0;
    for (int i = 0; i < e->W * e->H; i++) c += e->grid[i];
    return c;
}

int orc_wrapper_step(orc_env *e, int a1, int a2, double *reward_out, int *done_out, orc_info *info)
{
    if (a1 < 0 || a1 > 12 || a2 < 0 || a2 > 12) return -1;
    const int prev1 = e->prev_alive[0], prev2 = e->prev_alive[1];     /* :319-320 */
    const int prev_walk = count_white(e);                             /* :328-329 (P1 walks WHITE) */

    orc_step_players(e, a1, a2);                                      /* :332 */
    e->step_count += 1;

    const orc_player *p1 = &e->p[0], *p2 = &e->p[1];
    double reward = 0.0;
    int done = 0;

    int tile_diff = count_white(e) - prev_walk;                       /* :349-354 */
    if (tile_diff > 0) reward += tile_diff * 0.01;

    info->landed_hit = (prev2 && !p2->alive);                         /* :357-362 */
    if (info->landed_hit) reward += 1.0;
    info->got_hit = (prev1 && !p1->alive);                            /* :364-369 */
    if (info->got_hit) reward -= 0.01;

    if (p1->alive && p1->ammo == 0) reward -= 0.001;                  /* :372-373 */

    if (p1->alive && p2->alive) {                                     /* :377-405 */
        int dist = abs(p1->x - p2->x) + abs(p1->y - p2->y);
        int max_dist = e->W + e->H;
        reward += 0.002 * (1.0 - (double)dist / (double)max_dist);
        int aligned = (p1->x == p2->x) || (p1->y == p2->y);
        if (aligned) reward += 0.002;
        if (5 <= a1 && a1 <= 12 && aligned && p1->ammo > 0) {
            int shot = (a1 - 1) & 3, aiming = 0;
            if (p1->x == p2->x) {
                if (p1->y < p2->y && shot == ORC_DOWN) aiming = 1;
                if (p1->y > p2->y && shot == ORC_UP) aiming = 1;
            } else if (p1->y == p2->y) {
                if (p1->x < p2->x && shot == ORC_RIGHT) aiming = 1;
                if (p1->x > p2->x && shot == ORC_LEFT) aiming = 1;
            }
            if (aiming) reward += 0.05;
        }
    }

    info->win = info->lose = 0;
    if (orc_round_over(e)) {                                          /* :408-422 */
        done = 1;
        int w = orc_winner(e);
        if (w == 1) { reward += 10.0; info->win = 1; }
        else if (w == 2) { reward -= 0.1; info->lose = 1; }
    } else {
        reward -= 0.001;                                              /* :423-427 */
    }

    e->prev_alive[0] = p1->alive;                                     /* :430-431 */
    e->prev_alive[1] = p2->alive;

    if (e->step_count >= e->max_episode_steps) {                      /* :434-438 */
        done = 1;
        if (!orc_round_over(e)) reward -= 2.0;
    }

    e->episode_return += reward;                                      /* :440-442 */
    info->episode_steps = e->step_count;
    info->episode_return = e->episode_return;
    *reward_out = reward;
    *done_out = done;
    return 0;
}

/* dummy opponent, env_wrappers.py:69-170 */
static int p2_walkable(const orc_env *e, int dir)
{
    int nx = e->p[1].x + DX[dir], ny = e->p[1].y + DY[dir];
    return in_bounds(e, nx, ny) && e->grid[ny * e->W + nx] != ORC_WHITE;   /* tile != p2.color */
}

int orc_dummy_decide(const orc_env *e, int hard, const double u[3],
                     const int perm_a[4], int swap_c, const int perm_b[4])
{
    const orc_player *p1 = &e->p[0], *p2 = &e->p[1];
    if (!p2->alive) return 0;                                         /* :77-78 */
    const double shoot_prob = hard ? 0.2 : 0.0;                       /* :81-89 */
    const double random_move_prob = hard ? 0.05 : 0.0;
    const double move_prob = 0.001;                                   /* only read in easy mode, :122-124 */

    int x_al = (p2->x == p1->x), y_al = (p2->y == p1->y);
    if (u[0] < shoot_prob && p2->ammo > 0 && (x_al || y_al)) {        /* :96-101 */
        if (x_al) return 5 + (p1->y < p2->y ? ORC_UP : ORC_DOWN);
        return 5 + (p1->x < p2->x ? ORC_LEFT : ORC_RIGHT);
    }

    int dirs[4] = { ORC_UP, ORC_DOWN, ORC_LEFT, ORC_RIGHT };          /* :104 */
    if (u[1] < random_move_prob) {                                    /* :105-119 */
        int t[4];
        for (int i = 0; i < 4; i++) t[i] = dirs[perm_a[i]];
        memcpy(dirs, t, sizeof t);
        if (p2_walkable(e, dirs[0])) return 1 + dirs[0];
    }

    if (!hard) {                                                      /* :122-124 */
        if (u[2] > move_prob) return 0;
    }

    int dx = p1->x - p2->x, dy = p1->y - p2->y;                       /* :127-136 */
    int cand[2], nc = 0;
    if (dx != 0) cand[nc++] = dx > 0 ? ORC_RIGHT : ORC_LEFT;
    if (dy != 0) cand[nc++] = dy > 0 ? ORC_DOWN : ORC_UP;
    if (nc == 2 && swap_c) { int t = cand[0]; cand[0] = cand[1]; cand[1] = t; }   /* :138 */
    for (int i = 0; i < nc; i++)                                      /* :140-152 */
        if (p2_walkable(e, cand[i])) return 1 + cand[i];

    int t[4];                                                         /* :155-168 */
    for (int i = 0; i < 4; i++) t[i] = dirs[perm_b[i]];
    for (int i = 0; i < 4; i++)
        if (p2_walkable(e, t[i])) return 1 + t[i];
    return 0;                                                         /* :170 */
}

/* ------------------------------------------------------------------------ */
/* Philox4x32-10 counter RNG (device dummy opponent + synthetic actions)      */
/* ------------------------------------------------------------------------ */

void orc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                    uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

/* random.shuffle's loop (for i in reversed(range(1, n)): j = randbelow(i+1); swap) with
 * randbelow(m) := mulhi(word, m) on successive words. */
static void fy_perm4(const uint32_t w[3], int perm[4])
{
    for (int i = 0; i < 4; i++) perm[i] = i;
    for (int i = 3, k = 0; i >= 1; i--, k++) {
        int j = (int)mulhi32(w[k], (uint32_t)(i + 1));
        int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
}

#define ORC_TAG_DUMMY  0x44554d59u   /* "DUMY" */
#define ORC_TAG_ACTION 0x4143544eu   /* "ACTN" */
#define ORC_KEY1       0x1234abcdu

/* Draw words of the device dummy (see csrc/inv_device.h): key = lowbias32(genv * 0x9E3779B9 + seed),
 * word(j) = lowbias32(key + (ctr * 16 + j) * 0x85EBCA6B); j = 0,1,2: u0,u1,u2; 3: candidate swap;
 * 4..6: shuffle A; 8..10: shuffle B.  The oracle evaluates every word eagerly and feeds the pinned
 * decision function; the kernel generates them lazily -- same values. */
static uint32_t lowbias32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

int orc_dummy_device(orc_env *e, int hard, uint32_t seed, uint32_t global_env)
{
    uint32_t ctr = e->opp_ctr++;
    uint32_t key = lowbias32(global_env * 0x9E3779B9u + seed);
    uint32_t w[11];
    for (uint32_t j = 0; j < 11; j++) w[j] = lowbias32(key + (ctr * 16u + j) * 0x85EBCA6Bu);
    double u[3] = { w[0] * (1.0 / 4294967296.0), w[1] * (1.0 / 4294967296.0), w[2] * (1.0 / 4294967296.0) };
    uint32_t wa[3] = { w[4], w[5], w[6] };
    uint32_t wb[3] = { w[8], w[9], w[10] };
    int perm_a[4], perm_b[4];
    fy_perm4(wa, perm_a);
    fy_perm4(wb, perm_b);
    int swap_c = mulhi32(w[3], 2u) == 0u;   /* n=2 shuffle: i=1, j=randbelow(2), swap iff j==0 */
    return orc_dummy_decide(e, hard, u, perm_a, swap_c, perm_b);
}

int orc_uniform_action(uint32_t seed, uint32_t global_env, uint32_t t)
{
    uint32_t x[4];
    orc_philox4x32(t, global_env, 0u, ORC_TAG_ACTION, seed, ORC_KEY1, x);
    return (int)mulhi32(x[0], 13u);
}

/* ------------------------------------------------------------------------ */
/* GAE, ppo_agent.py:127-157.  NumPy >= 2 scalar rules: every op is f32,      */
/* sequential, no FMA: d = (r + g*V[k+1]) - V[k]; A = d + gl*A_next.          */
/* ------------------------------------------------------------------------ */

void orc_gae_flat(const float *r, const float *v, const uint8_t *done, int64_t n,
                  float gamma, float gamma_lam, float last_value, float *adv, float *ret)
{
    float last_gae = 0.0f;
    for (int64_t t = n - 1; t >= 0; t--) {
        float vnext = (t + 1 < n) ? v[t + 1] : last_value;
        float delta;
        if (done[t]) {
            delta = r[t] - v[t];
            last_gae = delta;
        } else {
            float gv = gamma * vnext;
            float s = r[t] + gv;
            delta = s - v[t];
            float gl = gamma_lam * last_gae;
            last_gae = delta + gl;
        }
        adv[t] = last_gae;
    }
    for (int64_t t = 0; t < n; t++) ret[t] = adv[t] + v[t];
}

/* Per-env columns of a [T][N] buffer: the reference's scan applied to each env's own
 * trajectory (its num_envs == 1 behaviour). */
void orc_gae_per_env(const float *r, const float *v, const uint8_t *done, int T, int N,
                     float gamma, float gamma_lam, const float *last_value, float *adv, float *ret)
{
    for (int i = 0; i < N; i++) {
        float last_gae = 0.0f;
        for (int t = T - 1; t >= 0; t--) {
            int64_t k = (int64_t)t * N + i;
            float vnext = (t + 1 < T) ? v[k + N] : (last_value ? last_value[i] : 0.0f);
            float delta;
            if (done[k]) {
                delta = r[k] - v[k];
                last_gae = delta;
            } else {
                float gv = gamma * vnext;
                float s = r[k] + gv;
                delta = s - v[k];
                float gl = gamma_lam * last_gae;
                last_gae = delta + gl;
            }
            adv[k] = last_gae;
            ret[k] = last_gae + v[k];
        }
    }
}

/* ------------------------------------------------------------------------ */
/* batch helpers                                                              */
/* ------------------------------------------------------------------------ */

orc_env *orc_batch_create(int n, int W, int H, int max_episode_steps)
{
    orc_env *envs = (orc_env *)malloc(sizeof(orc_env) * (size_t)n);
    if (!envs) return NULL;
    for (int i = 0; i < n; i++)
        if (orc_init(&envs[i], W, H, max_episode_steps) != 0) { free(envs); return NULL; }
    return envs;
}

void orc_batch_destroy(orc_env *envs) { free(envs); }
orc_env *orc_batch_at(orc_env *envs, int i) { return &envs[i]; }

void orc_batch_seed_reset(orc_env *envs, int n, const int64_t *seeds)
{
    for (int i = 0; i < n; i++) {
        orc_mt_seed(&envs[i], seeds[i]);
        envs[i].opp_ctr = 0;
        orc_wrapper_reset(&envs[i]);
    }
}

void orc_batch_step(orc_env *envs, int n, const uint8_t *a1, const uint8_t *a2,
                    int hard, uint32_t opp_seed, uint32_t env_offset, int auto_reset,
                    float *obs, float *extra, float *reward, double *reward64,
                    uint8_t *done, orc_info *info, int nthreads)
{
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int i = 0; i < n; i++) {
        orc_env *e = &envs[i];
        int b2 = a2 ? (int)a2[i] : orc_dummy_device(e, hard, opp_seed, env_offset + (uint32_t)i);
        double rw = 0.0; int dn = 0; orc_info inf;
        memset(&inf, 0, sizeof inf);
        orc_wrapper_step(e, (int)a1[i], b2, &rw, &dn, &inf);
        if (reward) reward[i] = (float)rw;          /* np.array(..., dtype=float32), env_wrappers.py:525 */
        if (reward64) reward64[i] = rw;
        if (done) done[i] = (uint8_t)dn;
        if (info) info[i] = inf;
        if (dn && auto_reset) orc_wrapper_reset(e); /* training.py:148-151 */
        if (obs) {
            size_t stride = 12u * (size_t)e->W * (size_t)e->H;
            orc_build_obs(e, 0, obs + stride * (size_t)i, extra + 4u * (size_t)i);
        }
    }
}

/* canonical packing shared with include/inversus_hip.h */
void orc_get_grid_bits(const orc_env *e, uint32_t *words)
{
    int WH = e->W * e->H, nw = (WH + 31) / 32;
    memset(words, 0, sizeof(uint32_t) * (size_t)nw);
    for (int i = 0; i < WH; i++) if (e->grid[i]) words[i >> 5] |= 1u << (i & 31);
}

void orc_set_grid_bits(orc_env *e, const uint32_t *words)
{
    int WH = e->W * e->H;
    for (int i = 0; i < WH; i++) e->grid[i] = (uint8_t)((words[i >> 5] >> (i & 31)) & 1u);
}

uint16_t orc_pack_bullet(const orc_bullet *b)
{
    return (uint16_t)((b->x & 63) | ((b->y & 63) << 6) | ((b->dir & 3) << 12) | ((b->owner & 1) << 14));
}

void orc_unpack_bullet(uint16_t w, orc_bullet *b)
{
    b->x = w & 63; b->y = (w >> 6) & 63; b->dir = (w >> 12) & 3; b->owner = (w >> 14) & 1;
}
