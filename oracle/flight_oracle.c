/*
 * oracle/flight_oracle.c -- TEST INFRASTRUCTURE ONLY (see flight_oracle.h).
 *
 * Plain-C fp64 restatement of the reference environment path.  Reference
 * citations are relative to /root/reference.  Build with -ffp-contract=off:
 * CPython/NumPy scalar arithmetic never fuses multiply-add.
 *
 * Floating-point fidelity.  The reference's wall test is structurally knife-edged: an
 * agent clamped to exactly y = 50.0 that leaves the wall and returns by mirrored moves
 * lands on 50 +- 1 ulp, and `y > 50` then decides OUT_PUNISH and the yaw reflection
 * (golden trace easy_n1_am0_s5_a4, step 192).  The index-table shortcut of SURVEY.md
 * Appendix A is therefore NOT faithful.  This restatement reproduces the reference's
 * float operations one for one instead:
 *   - yaw is an fp64 accumulated exactly as flight_env_easy.py:259-266,281-284 does
 *     (+= pi/18, -= 2*pi, pi - yaw, 3*pi - yaw), cos/sin are libm's on that double
 *     (numpy's scalar cos/sin are libm's: verified in the build container);
 *   - `**2` is libm pow(x, 2.0) (CPython float_pow and numpy scalar power both end
 *     there); orc_set_exact_pow(0) switches to x*x for the timed cpu_baseline leg.
 * On the same libm this makes positions bit-identical to the reference (tests assert
 * atol = 0), and every integer (reward, flags, counts, draw order and values) exact.
 */
#include "flight_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * NumPy legacy RandomState (numpy/random/_mt19937 + legacy-distributions).
 * Classic block form: twist all 624 words when the block is exhausted.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t key[624];
    int pos;
    int has_gauss;
    double gauss;
    uint64_t words;
} np_rng;

static void np_seed(np_rng *r, uint32_t s) {
    /* np.random.seed(int) == init_genrand(s) */
    r->key[0] = s;
    for (int i = 1; i < 624; i++)
        r->key[i] = 1812433253u * (r->key[i - 1] ^ (r->key[i - 1] >> 30)) + (uint32_t)i;
    r->pos = 624;
    r->has_gauss = 0;
    r->gauss = 0.0;
    r->words = 0;
}

static void np_twist(np_rng *r) {
    uint32_t *mt = r->key;
    int k;
    uint32_t y;
    for (k = 0; k < 624 - 397; k++) {
        y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
        mt[k] = mt[k + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; k < 623; k++) {
        y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
        mt[k] = mt[k + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    r->pos = 0;
}

static uint32_t np_u32(np_rng *r) {
    if (r->pos == 624) np_twist(r);
    uint32_t y = r->key[r->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    r->words++;
    return y;
}

static double np_rand(np_rng *r) {
    /* random_sample: (a >> 5, b >> 6) -> 53 bits */
    uint32_t a = np_u32(r) >> 5, b = np_u32(r) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

static double np_randn(np_rng *r) {
    /* legacy_gauss: polar Box-Muller with one cached value */
    if (r->has_gauss) {
        double t = r->gauss;
        r->has_gauss = 0;
        r->gauss = 0.0;
        return t;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * np_rand(r) - 1.0;
        x2 = 2.0 * np_rand(r) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    r->gauss = f * x1;
    r->has_gauss = 1;
    return f * x2;
}

/* ------------------------------------------------------------------------------------------------ */
static double (*volatile libm_pow)(double, double) = pow; /* volatile: keep gcc from folding pow(x,2) to x*x */
static int g_exact_pow = 1;
void orc_set_exact_pow(int on) { g_exact_pow = on; }
static inline double sq(double v) { return g_exact_pow ? libm_pow(v, 2.0) : v * v; }

/* Trig of the accumulated heading.  Mode 0 (default): libm sin/cos -- what the reference calls, bit-pinned by
 * the golden traces.  Mode 1: correctly rounded via the double-double table (the arithmetic the HIP kernel
 * uses; glibc's sin/cos are misrounded by 1 ulp for ~0.2 % of the reachable headings).
 * Mode 1 with orc_set_exact_pow(0) is the configuration the HIP path must match bit for bit. */
static int g_trig_mode = 0;
void orc_set_trig_mode(int mode) { g_trig_mode = mode; }
static const double k_trig[37][7] = {
#include "trig_table.inc"
};
static void heading_trig(double yaw, double *s, double *c) {
    if (g_trig_mode == 0) {
        *s = sin(yaw);
        *c = cos(yaw);
        return;
    }
    int k = (int)(yaw * 5.729577951308232 + 0.5); /* 18/pi */
    k = k < 0 ? 0 : (k > 36 ? 36 : k);
    const double *r = k_trig[k];
    double t = yaw - r[0];
    double dh = t - r[1];
    double bb = dh - t;
    double err = (t - (dh - bb)) + ((-r[1]) - bb);
    double dl = err - r[2];
    if (fabs(dh) > 1e-6) {
        *s = sin(yaw);
        *c = cos(yaw);
        return;
    }
    *s = r[3] + ((r[4] + dh * (r[5] - 0.5 * dh * r[3])) + dl * r[5]);
    *c = r[5] + ((r[6] - dh * (r[3] + 0.5 * dh * r[5])) - dl * r[3]);
}

struct orc_env {
    orc_config c;
    double ax[ORC_MAX_AGENTS], ay[ORC_MAX_AGENTS];
    double yaw[ORC_MAX_AGENTS]; /* accumulated float yaw, exactly as the reference carries it */
    int out[ORC_MAX_AGENTS];
    double tx[ORC_MAX_TARGETS], ty[ORC_MAX_TARGETS];
    int found[ORC_MAX_TARGETS];
    int target_find, win, time_step, curr_reward, newly_mask;
    long total_reward;
    double *prob; /* [map][map] */
    np_rng rng;
    double *dlog;
    int64_t dlog_cap, dlog_n;
};

static double env_rand(orc_env *e) {
    double u = np_rand(&e->rng);
    if (e->dlog && e->dlog_n < e->dlog_cap) e->dlog[e->dlog_n] = u;
    e->dlog_n++;
    return u;
}

orc_env *orc_create(const orc_config *cfg) {
    if (cfg->n_agents < 1 || cfg->n_agents > ORC_MAX_AGENTS) return NULL;
    if (cfg->n_targets < 1 || cfg->n_targets > ORC_MAX_TARGETS) return NULL;
    if (cfg->map_size < 1 || cfg->map_size > ORC_MAP_MAX) return NULL;
    orc_env *e = (orc_env *)calloc(1, sizeof(orc_env));
    e->c = *cfg;
    /* flight_env.py:53 -- the map exists from construction */
    e->prob = (double *)malloc(sizeof(double) * cfg->map_size * cfg->map_size);
    for (int i = 0; i < cfg->map_size * cfg->map_size; i++) e->prob[i] = 0.5;
    np_seed(&e->rng, 0);
    return e;
}

void orc_destroy(orc_env *e) {
    if (!e) return;
    free(e->prob);
    free(e);
}

void orc_seed(orc_env *e, uint32_t seed) { np_seed(&e->rng, seed); }

/* flight_env.py:294-303 : fraction of the 4 cell corners strictly inside ANY agent's sensor disc */
static int corners_in_view(const orc_env *e, int i, int j) {
    const double R2 = (double)(e->c.view_range * e->c.view_range);
    int cnt = 0;
    for (int c = 0; c < 4; c++) {
        /* corner order (i,j),(i+1,j),(i,j+1),(i+1,j+1) -- order is irrelevant for the count */
        double x = (double)(i + (c & 1)), y = (double)(j + (c >> 1));
        for (int a = 0; a < e->c.n_agents; a++) {
            if (sq(x - e->ax[a]) + sq(y - e->ay[a]) < R2) {
                cnt++;
                break;
            }
        }
    }
    return cnt;
}

/* flight_env.py:275-292 */
static void update_prob_map(orc_env *e, int newly_mask) {
    const int L = e->c.map_size;
    int fx[ORC_MAX_TARGETS], fy[ORC_MAX_TARGETS], nf = 0;
    for (int j = 0; j < e->c.n_targets; j++)
        if (newly_mask & (1 << j)) {
            /* min(int(x), L-1): int() truncates toward zero; negative coords give <= 0 */
            int ix = (int)e->tx[j], iy = (int)e->ty[j];
            fx[nf] = ix < L - 1 ? ix : L - 1;
            fy[nf] = iy < L - 1 ? iy : L - 1;
            nf++;
        }
    const double q = 1.0 - e->c.detect_prob; /* (1-self.detect_prob) = 0.09999999999999998 */
    for (int i = 0; i < L; i++)
        for (int j = 0; j < L; j++) {
            int cnt = corners_in_view(e, i, j);
            if (cnt == 0) continue;
            int hit = 0;
            for (int k = 0; k < nf; k++)
                if (fx[k] == i && fy[k] == j) hit = 1;
            double *p = &e->prob[i * L + j];
            if (hit)
                *p = 1.0;
            else {
                double percent = (double)cnt / 4.0, pv = *p;
                *p = percent * q * pv / (q * pv + (1.0 - pv));
            }
        }
}

/* flight_env_easy.py:223-253 / flight_env.py:232-266 : detection pass + reward */
static void update_obs(orc_env *e) {
    const double R2 = (double)(e->c.view_range * e->c.view_range);
    int r = -1; /* MOVE_COST */
    int newly = 0;
    for (int i = 0; i < e->c.n_agents; i++) {
        double x = e->ax[i], y = e->ay[i];
        for (int j = 0; j < e->c.n_targets; j++) {
            if (sq(e->tx[j] - x) + sq(e->ty[j] - y) <= R2) {
                double u = env_rand(e); /* drawn for every in-range pair, found or not (quirk Q4) */
                if (!e->found[j] && u <= e->c.detect_prob) {
                    e->found[j] = 1;
                    r += 10; /* FIND_ONE_TGT */
                    e->target_find++;
                    newly |= 1 << j;
                    if (e->target_find == e->c.n_targets && !e->win) {
                        r += 100; /* FIND_ALL_TGT */
                        e->win = 1;
                    }
                }
            }
        }
        if (e->out[i]) r += -1; /* OUT_PUNISH */
    }
    e->curr_reward = r;
    e->newly_mask = newly;
    if (e->c.variant == 1) update_prob_map(e, newly);
}

/* flight_env_easy.py:79-182 / flight_env.py:83-191 */
void orc_reset(orc_env *e, int init) {
    const orc_config *c = &e->c;
    const double L = (double)c->map_size;
    if (c->variant == 1 && init)
        for (int i = 0; i < c->map_size * c->map_size; i++) e->prob[i] = 0.5;
    e->time_step = 0;
    e->target_find = 0;
    e->total_reward = 0;
    e->curr_reward = 0;
    e->win = 0;
    if (c->target_mode == 0) {
        const double a = L / 10.0;
        for (int j = 0; j < c->n_targets; j++) {
            double x = a * c->cx[j], y = a * c->cy[j];
            if (!c->deter[j]) {
                double ddx = a * c->dx[j], ddy = a * c->dy[j];
                double g1 = np_randn(&e->rng);
                double delta_x = ddx * 2.0 * (g1 - 0.5); /* (dx*2)*(randn-0.5), left to right */
                double g2 = np_randn(&e->rng);
                double delta_y = ddy * 2.0 * (g2 - 0.5);
                x += delta_x;
                y += delta_y;
            }
            e->tx[j] = x;
            e->ty[j] = y;
            e->found[j] = 0;
        }
    } else {
        for (int j = 0; j < c->n_targets; j++) {
            e->tx[j] = L * env_rand(e);
            e->ty[j] = L * env_rand(e);
            e->found[j] = 0;
        }
    }
    for (int i = 0; i < c->n_agents; i++) {
        /* i*map_size/(n-1): integer product first, then true division */
        double s = c->n_agents != 1 ? (double)(i * c->map_size) / (double)(c->n_agents - 1) : L / 2.0;
        switch (c->agent_mode) {
        case 0: e->ax[i] = s; e->ay[i] = 0.0; e->yaw[i] = M_PI / 2.0; break;
        case 1: e->ax[i] = s; e->ay[i] = L / 2.0; e->yaw[i] = M_PI / 2.0; break;
        case 2: e->ax[i] = 0.0; e->ay[i] = s; e->yaw[i] = 0.0; break;
        default: e->ax[i] = L; e->ay[i] = s; e->yaw[i] = M_PI; break;
        }
        e->out[i] = 0;
    }
    update_obs(e); /* reset runs one detection pass; its reward is discarded (quirk Q3) */
}

/* flight_env_easy.py:293-301 : repulsion on agent `idx`, evaluated at its PRE-move position against the
 * current list (lower-index agents already moved, quirk Q7) */
static void potential_force(const orc_env *e, int idx, double *fx, double *fy) {
    const double x = e->ax[idx], y = e->ay[idx];
    const double F2 = e->c.force_dist * e->c.force_dist;
    const double k = e->c.safe_dist * e->c.force_factor * e->c.velocity;
    double sx = 0.0, sy = 0.0;
    for (int j = 0; j < e->c.n_agents; j++) {
        if (j == idx) continue;
        double xa = e->ax[j], ya = e->ay[j];
        double d2 = sq(xa - x) + sq(ya - y);
        if (d2 < F2 && (xa != x || ya != y)) {
            double den = sq(x - xa) + sq(y - ya);
            sx += k * (x - xa) / den;
            sy += k * (y - ya) / den;
        }
    }
    *fx = sx;
    *fy = sy;
}

/* flight_env_easy.py:255-291 / flight_env.py:305-345 */
static void agent_step(orc_env *e, const int32_t *act) {
    const double L = (double)e->c.map_size;
    for (int i = 0; i < e->c.n_agents; i++) {
        double yaw = e->yaw[i];
        int a = act[i];
        if (a == 1) yaw += M_PI / 18.0;       /* dyaw = [0, np.pi/18, -np.pi/18] */
        else if (a == 2) yaw += -(M_PI / 18.0);
        if (yaw > 2.0 * M_PI) yaw -= 2.0 * M_PI;
        else if (yaw < 0.0) yaw += 2.0 * M_PI;
        double sn, cs;
        heading_trig(yaw, &sn, &cs);
        double x = e->ax[i] + e->c.velocity * cs;
        double y = e->ay[i] + e->c.velocity * sn;
        double fx, fy;
        potential_force(e, i, &fx, &fy);
        x += fx;
        y += fy;
        int hit = e->c.variant == 1 ? (x < 0.0 || x >= L || y < 0.0 || y >= L)  /* flight_env.py:328 */
                                    : (x < 0.0 || x > L || y < 0.0 || y > L);     /* flight_env_easy.py:278 */
        if (hit) {
            x = fmin(fmax(x, 0.0), L);
            y = fmin(fmax(y, 0.0), L);
            yaw = (yaw <= M_PI) ? M_PI - yaw : 3.0 * M_PI - yaw;
            e->out[i] = 1;
        } else
            e->out[i] = 0;
        e->ax[i] = x;
        e->ay[i] = y;
        e->yaw[i] = yaw;
    }
}

/* flight_env_easy.py:303-314 */
int orc_step(orc_env *e, const int32_t *actions, int32_t *reward, int32_t *terminated, int32_t *win) {
    for (int i = 0; i < e->c.n_agents; i++)
        if (actions[i] < 0 || actions[i] > 2) return -1;
    agent_step(e, actions);
    update_obs(e);
    e->total_reward += e->curr_reward;
    e->time_step += 1;
    int term = (e->target_find >= e->c.n_targets) || (e->time_step >= e->c.time_limit);
    if (reward) *reward = e->curr_reward;
    if (terminated) *terminated = term;
    if (win) *win = e->win;
    return 0;
}

static void agent_feats(const orc_env *e, int i, double *o) {
    const double L = (double)e->c.map_size;
    o[0] = (e->ax[i] - 0.5 * L) / (L / 2.0);
    o[1] = (e->ay[i] - 0.5 * L) / (L / 2.0);
    heading_trig(e->yaw[i], &o[3], &o[2]);
}

/* flight_env_easy.py:218-221 ; flight_env.py:223-230 (map first, 4 features last) */
void orc_get_obs(const orc_env *e, double *out) {
    const int n = e->c.n_agents, cells = e->c.map_size * e->c.map_size;
    if (e->c.variant == 0) {
        for (int i = 0; i < n; i++) agent_feats(e, i, out + 4 * i);
    } else {
        for (int i = 0; i < n; i++) {
            memcpy(out + (size_t)i * (cells + 4), e->prob, sizeof(double) * cells);
            agent_feats(e, i, out + (size_t)i * (cells + 4) + cells);
        }
    }
}

/* flight_env_easy.py:190-216 */
void orc_get_state(const orc_env *e, double *out) {
    const int n = e->c.n_agents, m = e->c.n_targets;
    const double L = (double)e->c.map_size;
    for (int i = 0; i < n; i++) agent_feats(e, i, out + 4 * i);
    for (int j = 0; j < m; j++) {
        out[4 * n + 3 * j + 0] = (e->tx[j] - 0.5 * L) / (L / 2.0);
        out[4 * n + 3 * j + 1] = (e->ty[j] - 0.5 * L) / (L / 2.0);
        out[4 * n + 3 * j + 2] = e->found[j] ? 1.0 : 0.0;
    }
}

void orc_get_agents(const orc_env *e, double *pos_xy, double *yaw, int32_t *out_flag) {
    for (int i = 0; i < e->c.n_agents; i++) {
        if (pos_xy) { pos_xy[2 * i] = e->ax[i]; pos_xy[2 * i + 1] = e->ay[i]; }
        if (yaw) yaw[i] = e->yaw[i];
        if (out_flag) out_flag[i] = e->out[i];
    }
}

void orc_set_agents(orc_env *e, const double *pos_xy, const double *yaw) {
    for (int i = 0; i < e->c.n_agents; i++) {
        if (pos_xy) { e->ax[i] = pos_xy[2 * i]; e->ay[i] = pos_xy[2 * i + 1]; }
        if (yaw) e->yaw[i] = yaw[i];
    }
}

void orc_get_targets(const orc_env *e, double *pos_xy, int32_t *found) {
    for (int j = 0; j < e->c.n_targets; j++) {
        if (pos_xy) { pos_xy[2 * j] = e->tx[j]; pos_xy[2 * j + 1] = e->ty[j]; }
        if (found) found[j] = e->found[j];
    }
}

void orc_set_targets(orc_env *e, const double *pos_xy, const int32_t *found) {
    for (int j = 0; j < e->c.n_targets; j++) {
        if (pos_xy) { e->tx[j] = pos_xy[2 * j]; e->ty[j] = pos_xy[2 * j + 1]; }
        if (found) e->found[j] = found[j] ? 1 : 0;
    }
    if (found) {
        int c = 0;
        for (int j = 0; j < e->c.n_targets; j++) c += e->found[j];
        e->target_find = c;
    }
}

void orc_get_counters(const orc_env *e, int32_t *o) {
    o[0] = e->target_find; o[1] = e->win; o[2] = e->time_step;
    o[3] = (int32_t)e->total_reward; o[4] = e->curr_reward; o[5] = e->newly_mask;
}

void orc_get_prob_map(const orc_env *e, double *out) {
    memcpy(out, e->prob, sizeof(double) * e->c.map_size * e->c.map_size);
}
void orc_set_prob_map(orc_env *e, const double *in) {
    memcpy(e->prob, in, sizeof(double) * e->c.map_size * e->c.map_size);
}
uint64_t orc_words_consumed(const orc_env *e) { return e->rng.words; }
void orc_set_draw_log(orc_env *e, double *buf, int64_t cap) { e->dlog = buf; e->dlog_cap = cap; e->dlog_n = 0; }
int64_t orc_draw_log_count(const orc_env *e) { return e->dlog_n; }
void orc_clear_draw_log(orc_env *e) { e->dlog_n = 0; }
uint32_t orc_rng_u32(orc_env *e) { return np_u32(&e->rng); }
double orc_rng_rand(orc_env *e) { return np_rand(&e->rng); }
double orc_rng_randn(orc_env *e) { return np_randn(&e->rng); }

/* ------------------------------------------------------------------------------------------------
 * batch driver (cpu_baseline leg of bench.py; also the oracle side of batched parity tests)
 * ---------------------------------------------------------------------------------------------- */
struct orc_batch {
    orc_config c;
    int64_t n;
    orc_env **envs;
};

orc_batch *orc_batch_create(const orc_config *cfg, int64_t batch, const uint32_t *seeds) {
    orc_batch *b = (orc_batch *)calloc(1, sizeof(orc_batch));
    b->c = *cfg;
    b->n = batch;
    b->envs = (orc_env **)calloc((size_t)batch, sizeof(orc_env *));
    for (int64_t i = 0; i < batch; i++) {
        b->envs[i] = orc_create(cfg);
        if (!b->envs[i]) { orc_batch_destroy(b); return NULL; }
        orc_seed(b->envs[i], seeds ? seeds[i] : (uint32_t)i);
    }
    return b;
}

void orc_batch_destroy(orc_batch *b) {
    if (!b) return;
    for (int64_t i = 0; i < b->n; i++) orc_destroy(b->envs[i]);
    free(b->envs);
    free(b);
}

orc_env *orc_batch_env(orc_batch *b, int64_t i) { return b->envs[i]; }

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_batch_reset(orc_batch *b, int init, const uint8_t *mask, int threads) {
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < b->n; i++)
        if (!mask || mask[i]) orc_reset(b->envs[i], init);
}

static void emit_f32(const orc_env *e, float *obs, float *state) {
    const int n = e->c.n_agents, m = e->c.n_targets, cells = e->c.map_size * e->c.map_size;
    double tmp[4 * ORC_MAX_AGENTS + 3 * ORC_MAX_TARGETS];
    if (state) {
        orc_get_state(e, tmp);
        for (int k = 0; k < 4 * n + 3 * m; k++) state[k] = (float)tmp[k];
    }
    if (obs) {
        if (e->c.variant == 0) {
            for (int i = 0; i < n; i++) {
                agent_feats(e, i, tmp);
                for (int k = 0; k < 4; k++) obs[4 * i + k] = (float)tmp[k];
            }
        } else {
            for (int i = 0; i < n; i++) {
                float *row = obs + (size_t)i * (cells + 4);
                for (int k = 0; k < cells; k++) row[k] = (float)e->prob[k];
                agent_feats(e, i, tmp);
                for (int k = 0; k < 4; k++) row[cells + k] = (float)tmp[k];
            }
        }
    }
}

void orc_batch_step(orc_batch *b, const int32_t *actions, float *reward, uint8_t *terminated, uint8_t *win,
                    float *obs, float *state, int auto_reset, int freeze_done, int threads) {
    const int n = b->c.n_agents, m = b->c.n_targets;
    const size_t obs_w = (size_t)n * (b->c.variant == 0 ? 4 : b->c.map_size * b->c.map_size + 4);
    const size_t st_w = (size_t)(4 * n + 3 * m);
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < b->n; i++) {
        orc_env *e = b->envs[i];
        int done = (e->target_find >= m) || (e->time_step >= e->c.time_limit);
        int32_t r = 0, t = 1, w = e->win;
        if (done && auto_reset) {
            orc_reset(e, 0);
            done = 0;
        }
        if (!(done && freeze_done)) orc_step(e, actions + (size_t)i * n, &r, &t, &w);
        if (reward) reward[i] = (float)r;
        if (terminated) terminated[i] = (uint8_t)t;
        if (win) win[i] = (uint8_t)w;
        emit_f32(e, obs ? obs + (size_t)i * obs_w : NULL, state ? state + (size_t)i * st_w : NULL);
    }
}

/* T steps of every env inside ONE parallel region (bench.py's cpu_baseline leg): each thread takes a contiguous
 * chunk of envs and walks every env of it through all T steps (env-major, so an env's state stays in that core's
 * cache), writing the same per-step outputs as T orc_batch_step calls: actions [T][B][n]; reward/terminated/win
 * [T][B]; obs [T][B][n][obs_w]; state [T][B][4n+3m] (any output may be NULL). */
void orc_batch_rollout(orc_batch *b, const int32_t *actions, int T, float *reward, uint8_t *terminated, uint8_t *win,
                       float *obs, float *state, int auto_reset, int freeze_done, int threads) {
    orc_batch_rollout_rep(b, actions, T, 1, reward, terminated, win, obs, state, auto_reset, freeze_done, threads);
}

/* the same with the T-step action table walked `repeat` times inside the one parallel region (outputs overwritten each
 * time): the timed baseline uses it so that the fork/join is paid once per repeat * T steps */
void orc_batch_rollout_rep(orc_batch *b, const int32_t *actions, int T, int repeat, float *reward, uint8_t *terminated,
                           uint8_t *win, float *obs, float *state, int auto_reset, int freeze_done, int threads) {
    const int n = b->c.n_agents, m = b->c.n_targets;
    const size_t obs_w = (size_t)n * (b->c.variant == 0 ? 4 : b->c.map_size * b->c.map_size + 4);
    const size_t st_w = (size_t)(4 * n + 3 * m);
    const size_t B = (size_t)b->n;
    /* dynamic, 64 envs at a time: the envs are independent, so the schedule cannot change a result.  64 envs = whole cache lines
     * of every per-step output (64 terminated / win bytes, 256 reward bytes), so no two threads write one line -- chunks of 4
     * were a third slower than a static split for that reason -- and a thread the host deschedules costs the region its
     * current chunk instead of its whole static share */
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < b->n; i++) {
        orc_env *e = b->envs[i];
        for (int rs = 0; rs < repeat * T; rs++) {
            const int s = rs % T;
            int done = (e->target_find >= m) || (e->time_step >= e->c.time_limit);
            int32_t r = 0, t = 1, w = e->win;
            if (done && auto_reset) {
                orc_reset(e, 0);
                done = 0;
            }
            if (!(done && freeze_done)) orc_step(e, actions + ((size_t)s * B + (size_t)i) * n, &r, &t, &w);
            const size_t slot = (size_t)s * B + (size_t)i;
            if (reward) reward[slot] = (float)r;
            if (terminated) terminated[slot] = (uint8_t)t;
            if (win) win[slot] = (uint8_t)w;
            emit_f32(e, obs ? obs + slot * obs_w : NULL, state ? state + slot * st_w : NULL);
        }
    }
}
