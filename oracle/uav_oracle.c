/*
 * uav_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see uav_oracle.h).
 *
 * fp64 scalar restatement of the reference step path.  Every function names
 * the reference lines it follows (paths relative to /root/reference/src).
 * Deliberately written the slow, literal way (per-UAV observation lists,
 * sequential in-place pose updates) so that it can be read side by side with
 * the Python; the batched / fused formulation lives in the HIP kernels.
 *
 * Parity status: PINNED by tests/golden/ (npz) (tests/test_oracle_golden.py).
 */
#include "uav_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.141592653589793   /* math.pi */
#define ORC_E  2.718281828459045   /* math.e  */

static double dist3(double ax, double ay, double az, double bx, double by, double bz)
{
    /* uav.py:53-59 (__distance): sqrt((x1-x2)**2 + (y1-y2)**2); 3-D adds z. */
    return sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by) + (az - bz) * (az - bz));
}

static double dist2(double ax, double ay, double bx, double by)
{
    /* uav.py:61-71 (distance, static). */
    return sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by));
}

static double py_fmod_pos(double a, double m)
{
    /* Python float %: result has the sign of the divisor (uav.py:97). */
    double r = fmod(a, m);
    if (r != 0.0 && ((r < 0.0) != (m < 0.0))) r += m;
    return r;
}

static double clip_and_normalize(double val, double floor_, double ceil_, int choice)
{
    /* utils/data_util.py:43-56. */
    double mid;
    if (val < floor_) val = floor_;
    if (val > ceil_) val = ceil_;
    mid = (floor_ + ceil_) / 2.0;
    if (choice == -1) return (val - floor_) / (ceil_ - floor_) - 1.0;
    if (choice == 0)  return (val - floor_) / (ceil_ - floor_);
    return (val - mid) / (mid - floor_);
}

static void note_margin(double *m, double d, double thr)
{
    double a = fabs(d - thr);
    if (a < *m) *m = a;
}

/* ---- PMINetwork eval-mode forward, PMINet.py:41-72 ------------------------ */
static double bn_eval(double x, const double *bn, int H, int h, double eps)
{
    /* BatchNorm1d in eval(): (x - running_mean) / sqrt(running_var + eps) * weight + bias */
    return (x - bn[2 * H + h]) / sqrt(bn[3 * H + h] + eps) * bn[0 * H + h] + bn[1 * H + h];
}

static double pmi_forward(const orc_pmi *p, const double x[12], double *scratch)
{
    int H = p->hidden, h, k;
    double *cat = scratch;           /* [3H] */
    double *hid = scratch + 3 * H;   /* [H]  */
    double out;
    for (h = 0; h < H; ++h) {        /* fc_comm + bn_comm + relu, PMINet.py:50-51 */
        double s = p->b_comm[h];
        for (k = 0; k < 5; ++k) s += p->w_comm[h * 5 + k] * x[k];
        s = bn_eval(s, p->bn_comm, H, h, p->bn_eps);
        cat[h] = s > 0.0 ? s : 0.0;
    }
    for (h = 0; h < H; ++h) {        /* fc_obs, PMINet.py:52-53 */
        double s = p->b_obs[h];
        for (k = 0; k < 4; ++k) s += p->w_obs[h * 4 + k] * x[5 + k];
        s = bn_eval(s, p->bn_obs, H, h, p->bn_eps);
        cat[H + h] = s > 0.0 ? s : 0.0;
    }
    for (h = 0; h < H; ++h) {        /* fc_boundary_state, PMINet.py:54-55 */
        double s = p->b_bs[h];
        for (k = 0; k < 3; ++k) s += p->w_bs[h * 3 + k] * x[9 + k];
        s = bn_eval(s, p->bn_bs, H, h, p->bn_eps);
        cat[2 * H + h] = s > 0.0 ? s : 0.0;
    }
    for (h = 0; h < H; ++h) {        /* fc1 + bn1 + relu, PMINet.py:58-60 */
        double s = p->b_fc1[h];
        for (k = 0; k < 3 * H; ++k) s += p->w_fc1[h * 3 * H + k] * cat[k];
        s = bn_eval(s, p->bn_fc1, H, h, p->bn_eps);
        hid[h] = s > 0.0 ? s : 0.0;
    }
    out = p->b_fc2[0];               /* fc2, PMINet.py:61 */
    for (h = 0; h < H; ++h) out += p->w_fc2[h] * hid[h];
    return (double)(float)out;       /* .item() of an fp32 tensor */
}

/* ---- one environment ------------------------------------------------------ */
typedef struct {
    double *obs_t;   /* [N][M][4] target_observation rows   (uav.py:101-122) */
    int    *cnt_t;   /* [N] */
    double *obs_u;   /* [N][N][5] uav_communication rows    (uav.py:124-147) */
    int    *cnt_u;   /* [N] */
    double *local;   /* [N][12] get_local_state             (uav.py:156-190) */
    double *rawr;    /* [N] uav.raw_reward */
    double *pmi_scratch;
    double *twall;   /* [M] test aid: each target's distance to the nearest wall test of its move */
} env_scratch;

static void step_one_env(const orc_config *c, const orc_pmi *pmi, env_scratch *s,
                         double *ux, double *uy, double *uz, double *uh, int32_t *ua,
                         double *tx, double *ty, double *tz, double *th,
                         const int32_t *act,
                         double *obs, double *reward, double *tt, double *bp, double *dup,
                         double *raw, int32_t *covered, double *margin, double *margin_row)
{
    const int N = c->n_uav, M = c->m_targets;
    const int three_d = (c->dim == 3);
    const int na_total = c->na * (c->nc > 0 ? c->nc : 1);
    double mg = INFINITY;
    int i, j, k;

    /* -- targets: TARGET.update_position, agent/target.py:27-60 ------------- */
    for (k = 0; k < M; ++k) {
        /* target.py:34 draws a dead random number; heading update is commented out. */
        tx[k] += c->dt * c->t_v_max * cos(th[k]);
        ty[k] += c->dt * c->t_v_max * sin(th[k]);
        note_margin(&mg, ty[k], 0.0);
        note_margin(&mg, ty[k], c->y_max);
        note_margin(&mg, tx[k], 0.0);
        note_margin(&mg, tx[k], c->x_max);
        s->twall[k] = INFINITY;
        note_margin(&s->twall[k], ty[k], 0.0);
        note_margin(&s->twall[k], ty[k], c->y_max);
        note_margin(&s->twall[k], tx[k], 0.0);
        note_margin(&s->twall[k], tx[k], c->x_max);
        if (0.0 > ty[k] || ty[k] > c->y_max) {          /* target.py:52-53 */
            th[k] = -th[k];
        } else if (tx[k] < 0.0 || tx[k] > c->x_max) {   /* target.py:54-58 */
            if (th[k] > 0.0) th[k] = ORC_PI - th[k];
            else             th[k] = -ORC_PI - th[k];
        }
    }

    /* -- sequential UAV sweep, environment.py:133-138 ----------------------- */
    for (i = 0; i < N; ++i) {
        /* UAV.update_position, uav.py:83-99 (+ discrete_action :73-81) */
        int a_idx = act[i];
        int a_turn = a_idx % c->na;
        int a_climb = a_idx / c->na;
        double w = (2.0 * (a_turn + 1) - c->na - 1) * c->u_h_max / (c->na - 1);
        double cg = 1.0, sg = 0.0;
        if (three_d && c->nc > 1) {
            double g = (2.0 * a_climb - (c->nc - 1)) * c->u_g_max / (c->nc - 1);
            cg = cos(g);
            sg = sin(g);
        }
        ua[i] = a_idx;
        ux[i] += c->dt * c->u_v_max * cg * cos(uh[i]);
        uy[i] += c->dt * c->u_v_max * cg * sin(uh[i]);
        if (three_d) uz[i] += c->dt * c->u_v_max * sg;
        uh[i] += c->dt * w;
        uh[i] = py_fmod_pos(uh[i] + ORC_PI, 2.0 * ORC_PI) - ORC_PI;
        /* margin_row[i] (test aid): the knife edges that can change THIS UAV's observation row, its three reward terms
         * and its raw reward -- its own range tests against targets (dp) and peers (dc on the sequential view, 2 dp on
         * the post-move poses) and the wall tests of the targets it observes or nearly observes (a mirrored heading
         * enters the row, uav.py:116-117).  The neighbour test of the cooperative reward and the strict coverage test
         * couple UAVs of the environment and stay in the per-environment margin only. */
        if (margin_row) margin_row[i] = INFINITY;

        /* UAV.observe_target, uav.py:101-122 (relative=True) */
        s->cnt_t[i] = 0;
        for (k = 0; k < M; ++k) {
            double d = three_d ? dist3(ux[i], uy[i], uz[i], tx[k], ty[k], tz[k])
                               : dist2(ux[i], uy[i], tx[k], ty[k]);
            note_margin(&mg, d, c->dp);
            if (margin_row) {
                note_margin(&margin_row[i], d, c->dp);
                if (d <= c->dp + 1.0 && s->twall[k] < margin_row[i]) margin_row[i] = s->twall[k];
            }
            if (d <= c->dp) {
                double *row = s->obs_t + ((size_t)i * M + s->cnt_t[i]) * 4;
                row[0] = (tx[k] - ux[i]) / c->dp;
                row[1] = (ty[k] - uy[i]) / c->dp;
                row[2] = cos(th[k]) * c->t_v_max / c->u_v_max - cos(uh[i]);
                row[3] = sin(th[k]) * c->t_v_max / c->u_v_max - sin(uh[i]);
                s->cnt_t[i]++;
            }
        }
        /* UAV.observe_uav, uav.py:124-147: peers j<i already moved, j>i not yet. */
        s->cnt_u[i] = 0;
        for (j = 0; j < N; ++j) {
            double d;
            if (j == i) continue;   /* `uav != self` */
            d = three_d ? dist3(ux[i], uy[i], uz[i], ux[j], uy[j], uz[j])
                        : dist2(ux[i], uy[i], ux[j], uy[j]);
            note_margin(&mg, d, c->dc);
            if (margin_row) note_margin(&margin_row[i], d, c->dc);
            if (d <= c->dc) {
                double *row = s->obs_u + ((size_t)i * N + s->cnt_u[i]) * 5;
                row[0] = (ux[j] - ux[i]) / c->dc;
                row[1] = (uy[j] - uy[i]) / c->dc;
                row[2] = cos(uh[j]) - cos(uh[i]);
                row[3] = sin(uh[j]) - sin(uh[i]);
                row[4] = (double)(ua[j] - ua[i]) / na_total;
                s->cnt_u[i]++;
            }
        }
    }

    /* -- get_local_state for every UAV, uav.py:156-190 ---------------------- */
    for (i = 0; i < N; ++i) {
        double *ls = s->local + (size_t)i * 12;
        int r, q;
        if (s->cnt_u[i] > 0) {
            double acc[5] = {0, 0, 0, 0, 0};
            for (r = 0; r < s->cnt_u[i]; ++r) {
                const double *row = s->obs_u + ((size_t)i * N + r) * 5;
                /* uav.py:165: distance between the NORMALISED offsets and the ABSOLUTE pose */
                double d = fmin(dist2(row[0], row[1], ux[i], uy[i]), 1.0);
                for (q = 0; q < 5; ++q) acc[q] += row[q] / d;
            }
            for (q = 0; q < 5; ++q) ls[q] = acc[q] / s->cnt_u[i];
        } else {
            for (q = 0; q < 5; ++q) ls[q] = -1.0;
        }
        if (s->cnt_t[i] > 0) {
            double acc[4] = {0, 0, 0, 0};
            for (r = 0; r < s->cnt_t[i]; ++r) {
                const double *row = s->obs_t + ((size_t)i * M + r) * 4;
                double d = fmin(dist2(row[0], row[1], ux[i], uy[i]), 1.0);   /* uav.py:179 */
                for (q = 0; q < 4; ++q) acc[q] += row[q] / d;
            }
            for (q = 0; q < 4; ++q) ls[5 + q] = acc[q] / s->cnt_t[i];
        } else {
            for (q = 0; q < 4; ++q) ls[5 + q] = -1.0;
        }
        ls[9]  = ux[i] / c->dc;                 /* uav.py:154 */
        ls[10] = uy[i] / c->dc;
        ls[11] = (double)ua[i] / na_total;
    }

    /* -- calculate_rewards loop 1, environment.py:200-220 ------------------- */
    for (i = 0; i < N; ++i) {
        double track = 0.0, punish = 0.0, bpun, d_bdr;
        double t_n, d_n, b_n;
        int inside;
        /* uav.py:199-212, called with the TARGET list (uav.py:257) */
        for (k = 0; k < M; ++k) {
            double d = three_d ? dist3(ux[i], uy[i], uz[i], tx[k], ty[k], tz[k])
                               : dist2(ux[i], uy[i], tx[k], ty[k]);
            if (d <= c->dp) track += 1.0 + (c->dp - d) / c->dp;
        }
        /* uav.py:231-250 */
        d_bdr = fmin(fmin(ux[i], c->x_max - ux[i]), fmin(uy[i], c->y_max - uy[i]));
        inside = (0.0 <= ux[i] && ux[i] <= c->x_max && 0.0 <= uy[i] && uy[i] <= c->y_max);
        if (three_d) {
            d_bdr = fmin(d_bdr, fmin(uz[i], c->z_max - uz[i]));
            inside = inside && (0.0 <= uz[i] && uz[i] <= c->z_max);
        }
        if (inside) bpun = (d_bdr < c->dp) ? -0.5 * (c->dp - d_bdr) / c->dp : 0.0;
        else        bpun = -0.5;
        /* uav.py:214-229, radio = 2, all poses are post-move here */
        for (j = 0; j < N; ++j) {
            double d;
            if (j == i) continue;
            d = three_d ? dist3(ux[i], uy[i], uz[i], ux[j], uy[j], uz[j])
                        : dist2(ux[i], uy[i], ux[j], uy[j]);
            note_margin(&mg, d, 2.0 * c->dp);
            if (margin_row) note_margin(&margin_row[i], d, 2.0 * c->dp);
            note_margin(&mg, d, c->dp);   /* neighbour test of the cooperative reward */
            if (d <= 2.0 * c->dp)
                punish += -0.5 * exp((2.0 * c->dp - d) / (2.0 * c->dp));
        }
        /* environment.py:207-211 */
        t_n = clip_and_normalize(track, 0.0, 2.0 * c->norm_m_targets, 0);
        d_n = clip_and_normalize(punish, -ORC_E / 2.0 * c->norm_n_uav, 0.0, -1);
        b_n = clip_and_normalize(bpun, -0.5, 0.0, -1);
        if (tt)  tt[i] = t_n;
        if (bp)  bp[i] = b_n;
        if (dup) dup[i] = d_n;
        s->rawr[i] = c->alpha * t_n + c->beta * b_n + c->gamma * d_n;   /* :219 */
        if (raw) raw[i] = s->rawr[i];
    }

    /* -- calculate_rewards loop 2, environment.py:222-226 ------------------- */
    for (i = 0; i < N; ++i) {
        double a = c->cooperative, r;
        if (a == 0.0) {                                   /* uav.py:270 / :300 */
            r = s->rawr[i];
        } else if (pmi) {                                 /* uav.py:262-291 */
            double nb_r[1024], nb_s[1024];
            int cnt = 0;
            for (j = 0; j < N; ++j) {
                double d, in[12];
                int q;
                if (j == i) continue;
                d = three_d ? dist3(ux[i], uy[i], uz[i], ux[j], uy[j], uz[j])
                            : dist2(ux[i], uy[i], ux[j], uy[j]);
                if (d > c->dp) continue;
                for (q = 0; q < 12; ++q)   /* _input = la * other_uav_la, fed as fp32 (PMINet.py:66) */
                    in[q] = (double)(float)(s->local[(size_t)i * 12 + q] * s->local[(size_t)j * 12 + q]);
                nb_r[cnt] = s->rawr[j];
                nb_s[cnt] = pmi_forward(pmi, in, s->pmi_scratch);
                cnt++;
            }
            if (cnt) {                                    /* scipy softmax on fp32, uav.py:286-288 */
                double mx = nb_s[0], den = 0.0, acc = 0.0;
                for (j = 1; j < cnt; ++j) if (nb_s[j] > mx) mx = nb_s[j];
                for (j = 0; j < cnt; ++j) den += exp(nb_s[j] - mx);
                for (j = 0; j < cnt; ++j) acc += nb_r[j] * (double)(float)(exp(nb_s[j] - mx) / den);
                r = (1.0 - a) * s->rawr[i] + a * acc;
            } else {
                r = (1.0 - a) * s->rawr[i];               /* uav.py:290 */
            }
        } else {                                          /* uav.py:293-310 */
            double sum = 0.0;
            int cnt = 0;
            for (j = 0; j < N; ++j) {
                double d;
                if (j == i) continue;
                d = three_d ? dist3(ux[i], uy[i], uz[i], ux[j], uy[j], uz[j])
                            : dist2(ux[i], uy[i], ux[j], uy[j]);
                if (d <= c->dp) { sum += s->rawr[j]; cnt++; }
            }
            /* `X if len(nb) else 0` binds over the whole sum (uav.py:308-309) */
            r = cnt ? (1.0 - a) * s->rawr[i] + a * sum / cnt : 0.0;
        }
        if (reward) reward[i] = clip_and_normalize(r, -1.0, 1.0, 1);   /* :225 */
    }

    /* -- get_states, environment.py:144 ------------------------------------- */
    if (obs) memcpy(obs, s->local, sizeof(double) * 12 * (size_t)N);

    /* -- calculate_covered_target, environment.py:246-253 (strict <) ------- */
    {
        int cov = 0;
        for (k = 0; k < M; ++k) {
            for (i = 0; i < N; ++i) {
                double d = three_d ? dist3(ux[i], uy[i], uz[i], tx[k], ty[k], tz[k])
                                   : dist2(ux[i], uy[i], tx[k], ty[k]);
                if (d < c->dp) { cov++; break; }
            }
        }
        if (covered) *covered = cov;
    }
    if (margin) *margin = mg;
}

static int scratch_alloc(env_scratch *s, int N, int M, int H)
{
    s->obs_t = (double *)malloc(sizeof(double) * 4 * (size_t)N * (M > 0 ? M : 1));
    s->cnt_t = (int *)malloc(sizeof(int) * N);
    s->obs_u = (double *)malloc(sizeof(double) * 5 * (size_t)N * N);
    s->cnt_u = (int *)malloc(sizeof(int) * N);
    s->local = (double *)malloc(sizeof(double) * 12 * (size_t)N);
    s->rawr  = (double *)malloc(sizeof(double) * N);
    s->pmi_scratch = (double *)malloc(sizeof(double) * 4 * (size_t)(H > 0 ? H : 1));
    s->twall = (double *)malloc(sizeof(double) * (size_t)(M > 0 ? M : 1));
    return (s->obs_t && s->cnt_t && s->obs_u && s->cnt_u && s->local && s->rawr && s->pmi_scratch && s->twall) ? 0 : -1;
}

static void scratch_free(env_scratch *s)
{
    free(s->obs_t); free(s->cnt_t); free(s->obs_u); free(s->cnt_u);
    free(s->local); free(s->rawr); free(s->pmi_scratch); free(s->twall);
}

int orc_step_rows(const orc_config *cfg,
                  double *ux, double *uy, double *uz, double *uh, int32_t *ua,
                  double *tx, double *ty, double *tz, double *th,
                  const int32_t *actions, const orc_pmi *pmi,
                  double *obs, double *reward, double *terms, double *raw,
                  int32_t *covered, double *margin, double *margin_row, int n_threads)
{
    const int B = cfg->n_envs, N = cfg->n_uav, M = cfg->m_targets;
    const size_t BN = (size_t)B * N;
    int fail = 0;
    if (N < 1 || N > 1024 || M < 0 || cfg->na < 2) return -1;
    if (cfg->dim == 3 && (!uz || (M > 0 && !tz))) return -2;
    if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
        env_scratch s;
        int ok = scratch_alloc(&s, N, M, pmi ? pmi->hidden : 0) == 0;
        long b;
        if (!ok) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            fail = 1;
        }
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (b = 0; b < B; ++b) {
            size_t un = (size_t)b * N, tm = (size_t)b * M;
            if (!ok) continue;
            step_one_env(cfg, pmi, &s,
                         ux + un, uy + un, uz ? uz + un : NULL, uh + un, ua + un,
                         tx + tm, ty + tm, tz ? tz + tm : NULL, th + tm,
                         actions + un,
                         obs ? obs + un * 12 : NULL,
                         reward ? reward + un : NULL,
                         terms ? terms + 0 * BN + un : NULL,
                         terms ? terms + 1 * BN + un : NULL,
                         terms ? terms + 2 * BN + un : NULL,
                         raw ? raw + un : NULL,
                         covered ? covered + b : NULL,
                         margin ? margin + b : NULL,
                         margin_row ? margin_row + un : NULL);
        }
        scratch_free(&s);
    }
    return fail ? -3 : 0;
}

int orc_step(const orc_config *cfg,
             double *ux, double *uy, double *uz, double *uh, int32_t *ua,
             double *tx, double *ty, double *tz, double *th,
             const int32_t *actions, const orc_pmi *pmi,
             double *obs, double *reward, double *terms, double *raw,
             int32_t *covered, double *margin, int n_threads)
{
    return orc_step_rows(cfg, ux, uy, uz, uh, ua, tx, ty, tz, th, actions, pmi, obs, reward, terms, raw, covered, margin, NULL,
                         n_threads);
}

int orc_reset_obs(const orc_config *cfg, const double *ux, const double *uy,
                  const int32_t *ua, double *obs)
{
    const size_t BN = (size_t)cfg->n_envs * cfg->n_uav;
    const int na_total = cfg->na * (cfg->nc > 0 ? cfg->nc : 1);
    size_t g;
    int q;
    for (g = 0; g < BN; ++g) {
        double *ls = obs + g * 12;
        for (q = 0; q < 9; ++q) ls[q] = -1.0;            /* uav.py:174,186 */
        ls[9]  = ux[g] / cfg->dc;
        ls[10] = uy[g] / cfg->dc;
        ls[11] = (double)ua[g] / na_total;
    }
    return 0;
}

/* ---- Philox4x32-10 (Salmon et al., SC'11), our reset stream --------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    int r;
    for (r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static float u01f(uint32_t r) { return (float)(r >> 8) * 0x1.0p-24f; }

int orc_reset_philox(const orc_config *cfg, uint64_t seed, uint32_t episode,
                     int64_t env_offset,
                     double *ux, double *uy, double *uz, double *uh, int32_t *ua,
                     double *tx, double *ty, double *tz, double *th)
{
    const int B = cfg->n_envs, N = cfg->n_uav, M = cfg->m_targets;
    const uint32_t na_total = (uint32_t)(cfg->na * (cfg->nc > 0 ? cfg->nc : 1));
    const float pi_f = (float)ORC_PI, two_pi_f = 2.0f * (float)ORC_PI;
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    long b;
    int i, k;
    for (b = 0; b < B; ++b) {
        uint64_t gid = (uint64_t)(env_offset + b);
        for (i = 0; i < N; ++i) {
            size_t g = (size_t)b * N + i;
            uint32_t ctr[4] = {(uint32_t)gid, episode, (uint32_t)i, 0x55415631u ^ (uint32_t)(gid >> 32)};
            uint32_t r[4];
            orc_philox4x32_10(ctr, key, r);
            /* layout of environment.py:105-107: x_i = i*x_max/(N+1) (1-based i), y = y_max/2 */
            ux[g] = (double)(float)((double)(i + 1) * cfg->x_max / (double)(N + 1));
            uy[g] = (double)(float)(cfg->y_max / 2.0);
            if (uz) uz[g] = (double)(float)(cfg->z_max / 2.0);
            uh[g] = (double)fmaf(u01f(r[0]), two_pi_f, -pi_f);
            ua[g] = (int32_t)(((uint64_t)r[1] * na_total) >> 32);
        }
        for (k = 0; k < M; ++k) {
            size_t g = (size_t)b * M + k;
            uint32_t ctr[4] = {(uint32_t)gid, episode, (uint32_t)(N + k), 0x55415631u ^ (uint32_t)(gid >> 32)};
            uint32_t r[4];
            orc_philox4x32_10(ctr, key, r);
            tx[g] = (double)(u01f(r[0]) * (float)cfg->x_max);
            ty[g] = (double)(u01f(r[1]) * (float)cfg->y_max);
            th[g] = (double)fmaf(u01f(r[2]), two_pi_f, -pi_f);
            if (tz) tz[g] = (double)(u01f(r[3]) * (float)cfg->z_max);
        }
    }
    return 0;
}

/* ---- greedy baseline policy, uav.py:324-369 -------------------------------------------------- */
int orc_greedy_actions(const orc_config *cfg, uint64_t seed, int64_t env_offset, const int32_t *step_count,
                       const double *ux, const double *uy, const double *uh,
                       const double *tx, const double *ty,
                       int32_t *actions, double *mg_score, double *mg_angle, double *mg_dist,
                       int force_argmax, double *best_angle_out, int32_t *branch_out)
{
    const int B = cfg->n_envs, N = cfg->n_uav, M = cfg->m_targets, na = cfg->na;
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    long b;
    if (cfg->dim != 2) return -1;
    for (b = 0; b < B; ++b) {
        const double *x = ux + (size_t)b * N, *y = uy + (size_t)b * N, *h = uh + (size_t)b * N;
        const double *gx = tx + (size_t)b * M, *gy = ty + (size_t)b * M;
        const uint64_t gid = (uint64_t)(env_offset + b);
        double md = INFINITY;
        int i, j, k, a;
        for (i = 0; i < N; ++i) {
            uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)step_count[b], (uint32_t)i, 0x47524459u ^ (uint32_t)(gid >> 32)};
            uint32_t r[4];
            double best = -INFINITY, second = -INFINITY, best_angle = 0.0;
            double ms = INFINITY, ma = INFINITY;                    /* this UAV's margins */
            orc_philox4x32_10(ctr, key, r);
            if (mg_score) mg_score[(size_t)b * N + i] = INFINITY;
            if (mg_angle) mg_angle[(size_t)b * N + i] = INFINITY;
            if (best_angle_out) best_angle_out[(size_t)b * N + i] = NAN;
            if (branch_out) branch_out[(size_t)b * N + i] = 0;
            if (!force_argmax && u01f(r[0]) < 0.25f) {              /* uav.py:338-339 */
                actions[(size_t)b * N + i] = (int32_t)(((uint64_t)r[1] * (uint32_t)na) >> 32);
                continue;
            }
            for (k = 0; k < M; ++k) {                               /* uav.py:341-362 */
                const double d_t = dist2(x[i], y[i], gx[k], gy[k]);
                double pen = 0.0, score;
                for (j = 0; j < N; ++j) {
                    if (x[j] != x[i] || y[j] != y[i]) {             /* `(uav_x, uav_y) != (self.x, self.y)` */
                        const double d = dist2(x[j], y[j], gx[k], gy[k]);
                        note_margin(&md, d, cfg->dc);
                        if (d < cfg->dc) pen += 0.8;
                    }
                }
                score = 1.0 / d_t - pen;
                if (score > best) {
                    second = best;
                    best = score;
                    best_angle = atan2(gy[k] - y[i], gx[k] - x[i]) - h[i];
                } else if (score > second) {
                    second = score;
                }
            }
            if (best - second < ms) ms = best - second;
            if (best_angle_out) best_angle_out[(size_t)b * N + i] = best_angle;   /* what uav.py:362 leaves behind */
            const int straight = !force_argmax && u01f(r[2]) < 0.3f; /* uav.py:365-366 */
            if (branch_out) branch_out[(size_t)b * N + i] = straight ? 1 : 2;
            if (straight) best_angle = 0.0;
            {   /* find_closest_a_idx, defined here: nearest turn rate of uav.py:73-81 to the wrapped angle */
                const double ang = py_fmod_pos(best_angle + ORC_PI, 2.0 * ORC_PI) - ORC_PI;
                double bestd = INFINITY, secondd = INFINITY;
                int besta = 0;
                for (a = 0; a < na; ++a) {
                    const double w = cfg->dt * (2.0 * (a + 1) - na - 1) * cfg->u_h_max / (na - 1);
                    const double dd = fabs(ang - w);
                    if (dd < bestd) { secondd = bestd; bestd = dd; besta = a; }
                    else if (dd < secondd) secondd = dd;
                }
                /* angle 0 sits exactly between the two middle turn rates: an exact tie, lowest index wins */
                if (!straight && (secondd - bestd) * 0.5 < ma) ma = (secondd - bestd) * 0.5;
                actions[(size_t)b * N + i] = besta;
            }
            if (mg_score) mg_score[(size_t)b * N + i] = ms;
            if (mg_angle) mg_angle[(size_t)b * N + i] = ma;
        }
        if (mg_dist) mg_dist[b] = md;
    }
    return 0;
}

/* ---- the learner's shared actor: FnnPolicyNet.forward (reference src/models/actor_critic.py:85-98:
 * Linear(12,H) - ReLU - Linear(H,A) - softmax) and ActorCritic.take_action (actor_critic.py:138-148:
 * Categorical(probs).sample()).  torch draws from its own generator; the restated draw is the inverse
 * CDF of the same probabilities at the Philox uniform keyed by (seed, global env, step_count, uav) --
 * the definition include/uavtrack.h gives uavtrack_actor_actions.  mode 1 = argmax (lowest index on ties).
 * w1 [H][12], b1 [H], w2 [A][H], b2 [A] (torch layouts, fp64).  probs [B][N][A] (nullable);
 * margin [B] (nullable) = min over the env's UAVs of the distance from the uniform to the nearest CDF
 * boundary (mode 0) / of the gap between the two largest probabilities (mode 1). */
int orc_actor_actions(const orc_config *cfg, uint64_t seed, int64_t env_offset, const int32_t *step_count,
                      const double *obs, const double *w1, const double *b1, const double *w2, const double *b2,
                      int32_t hidden, int32_t mode, int32_t *actions, double *probs, double *margin)
{
    const int B = cfg->n_envs, N = cfg->n_uav, A = cfg->na * cfg->nc, H = hidden;
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    long b;
    if (A < 1 || A > 64 || H < 1) return -1;
    for (b = 0; b < B; ++b) {
        const uint64_t gid = (uint64_t)(env_offset + b);
        double mg = INFINITY;
        int i, j, k, h;
        for (i = 0; i < N; ++i) {
            const double *x = obs + ((size_t)b * N + i) * 12;
            double lg[64], pr[64], m = -INFINITY, S = 0.0;
            for (j = 0; j < A; ++j) lg[j] = b2[j];
            for (h = 0; h < H; ++h) {
                double a = b1[h];
                for (k = 0; k < 12; ++k) a += w1[(size_t)h * 12 + k] * x[k];
                if (a < 0.0) a = 0.0;
                for (j = 0; j < A; ++j) lg[j] += w2[(size_t)j * H + h] * a;
            }
            for (j = 0; j < A; ++j) if (lg[j] > m) m = lg[j];
            for (j = 0; j < A; ++j) { pr[j] = exp(lg[j] - m); S += pr[j]; }
            for (j = 0; j < A; ++j) {
                pr[j] /= S;
                if (probs) probs[((size_t)b * N + i) * A + j] = pr[j];
            }
            if (mode == 1) {
                int best = 0;
                double second = -INFINITY;
                for (j = 1; j < A; ++j) {
                    if (pr[j] > pr[best]) { second = pr[best]; best = j; }
                    else if (pr[j] > second) second = pr[j];
                }
                if (A > 1 && pr[best] - second < mg) mg = pr[best] - second;
                actions[(size_t)b * N + i] = best;
            } else {
                /* one Philox block per four consecutive steps, word = step & 3 (include/uavtrack.h) */
                uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)step_count[b] >> 2, (uint32_t)i, 0x4143544Fu ^ (uint32_t)(gid >> 32)};
                uint32_t r[4];
                double u, c = 0.0;
                int pick = A - 1;
                orc_philox4x32_10(ctr, key, r);
                u = (double)u01f(r[(uint32_t)step_count[b] & 3u]);
                for (j = 0; j < A; ++j) {
                    c += pr[j];
                    if (j < A - 1 && fabs(c - u) < mg) mg = fabs(c - u);
                    if (c > u) { pick = j; break; }
                }
                for (++j; j < A - 1; ++j) { c += pr[j]; if (fabs(c - u) < mg) mg = fabs(c - u); }
                actions[(size_t)b * N + i] = pick;
            }
        }
        if (margin) margin[b] = mg;
    }
    return 0;
}
