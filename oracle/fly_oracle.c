/*
 * fly_oracle.c — TEST INFRASTRUCTURE (see fly_oracle.h for scope and parity status).
 * Scalar fp32 CPU restatement of petim0/fly_bProject's hot path, one function per reference
 * site, every op rounded separately (build with -ffp-contract=off) in the order the
 * reference's torch expressions evaluate.
 */
#include "fly_oracle.h"
#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* cpu_baseline leg: bound the OpenMP team to the host share of one GPU */
void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---- K3 physics, build-defined (two precisions) ---------------------------------------- */
#define REAL float
#define PHYS_NAME phys_f32
#define SIN sinf
#define COS cosf
#define SQRT sqrtf
#include "fly_physics.inc"
#undef REAL
#undef PHYS_NAME
#undef SIN
#undef COS
#undef SQRT
#define REAL double
#define PHYS_NAME phys_f64
#define SIN sin
#define COS cos
#define SQRT sqrt
#include "fly_physics.inc"
#undef REAL
#undef PHYS_NAME
#undef SIN
#undef COS
#undef SQRT

void orc_physics_step(const OrcConfig* c, float* root, float* dof_pos, float* dof_vel,
                      const float* targets, float* contact, int64_t n)
{ phys_f32(c, root, dof_pos, dof_vel, targets, contact, n); }

void orc_physics_step_f64(const OrcConfig* c, double* root, double* dof_pos, double* dof_vel,
                          const double* targets, double* contact, int64_t n)
{ phys_f64(c, root, dof_pos, dof_vel, targets, contact, n); }

/* ---- K1: fly.py:626-657; isaacgym scale(x,lo,hi) = 0.5*(x+1.0)*(hi-lo)+lo ---------------- */
void orc_scale_actions(const OrcConfig* c, const float* actions, float* targets, int64_t n)
{
    for (int64_t e = 0; e < n; ++e)
        for (int j = 0; j < ORC_NDOF; ++j) {
            float a = actions[e * ORC_NDOF + j];
            float t = 0.5f * (a + 1.0f);
            t = t * (c->dof_hi[j] - c->dof_lo[j]);
            targets[e * ORC_NDOF + j] = t + c->dof_lo[j];
        }
}

/* ---- K2: fly.py:446-480 -------------------------------------------------------------------- */
int64_t orc_reset_masked(const OrcConfig* c, float* root, float* dof_pos, float* dof_vel,
                         float* pot, float* prev_pot, int64_t* reset, int64_t* progress, int64_t n)
{
    int64_t cnt = 0;
    for (int64_t e = 0; e < n; ++e) {
        if (reset[e] == 0) continue;
        ++cnt;
        for (int j = 0; j < ORC_NDOF; ++j) {          /* fly.py:454-455 */
            dof_pos[e * ORC_NDOF + j] = c->dof_pose[j];
            dof_vel[e * ORC_NDOF + j] = 0.0f;
        }
        float* r = root + e * 13;                       /* fly.py:459, :344-346 */
        memset(r, 0, 13 * sizeof(float));
        r[2] = c->start_height;
        r[6] = 1.0f;
        /* fly.py:470-473: to_target = targets - origin_pos, z := 0; -norm/dt */
        float tx = c->target[0] - 0.0f, ty = c->target[1] - 0.0f;
        float nrm = sqrtf(tx * tx + ty * ty + 0.0f);
        float p = -nrm / c->dt;
        prev_pot[e] = p;
        pot[e] = p;
        reset[e] = 0;                                   /* fly.py:476-477 */
        progress[e] = 0;
    }
    return cnt;
}

/* ---- helper formulas (restated from the public isaacgym.torch_utils definitions) ----------- */
static void quat_mul_f(const float* a, const float* b, float* o)
{
    float x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3];
    float x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
    float ww = (z1 + x1) * (x2 + y2);
    float yy = (w1 - y1) * (w2 + z2);
    float zz = (w1 + y1) * (w2 - z2);
    float xx = ww + yy + zz;
    float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
    o[3] = qq - ww + (z1 - y1) * (y2 - z2);
    o[0] = qq - xx + (x1 + w1) * (x2 + w2);
    o[1] = qq - yy + (w1 - x1) * (y2 + z2);
    o[2] = qq - zz + (z1 + y1) * (w2 - x2);
}

/* quat_rotate / quat_rotate_inverse: a +- b + c */
static void quat_rot_f(const float* q, const float* v, float sign, float* o)
{
    float qw = q[3];
    float s = 2.0f * (qw * qw) - 1.0f;
    float cx = q[1] * v[2] - q[2] * v[1];
    float cy = q[2] * v[0] - q[0] * v[2];
    float cz = q[0] * v[1] - q[1] * v[0];
    float dot = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
    float a0 = v[0] * s, a1 = v[1] * s, a2 = v[2] * s;
    float b0 = cx * qw * 2.0f, b1 = cy * qw * 2.0f, b2 = cz * qw * 2.0f;
    float c0 = q[0] * dot * 2.0f, c1 = q[1] * dot * 2.0f, c2 = q[2] * dot * 2.0f;
    o[0] = (a0 + sign * b0) + c0;
    o[1] = (a1 + sign * b1) + c1;
    o[2] = (a2 + sign * b2) + c2;
}

static float py_mod_f(float a, float b)
{   /* torch.remainder semantics */
    float m = fmodf(a, b);
    if (m != 0.0f && ((b < 0.0f) != (m < 0.0f))) m += b;
    return m;
}

/* ---- K4: fly.py:771-805 (+ compute_heading_and_up, compute_rot, unscale) ------------------- */
void orc_pack_obs(const OrcConfig* c, const float* root, const float* dof_pos, const float* dof_vel,
                  const float* targets, const float* contact, float* pot, float* prev_pot,
                  float* obs, float* up_vec, float* heading_vec, int64_t n)
{
    const float two_pi = (float)(2.0 * M_PI);
    const float inv_start[4] = { -0.0f, -0.0f, -0.0f, 1.0f };  /* fly.py:129, :217 */
    const float b0[3] = { 1.0f, 0.0f, 0.0f }, b1[3] = { 0.0f, 0.0f, 1.0f }; /* fly.py:125-132 */
    _Pragma("omp parallel for schedule(static)")
    for (int64_t e = 0; e < n; ++e) {
        const float* r = root + e * 13;
        float* o = obs + e * ORC_NOBS;
        float tt[3] = { c->target[0] - r[0], c->target[1] - r[1], 0.0f };    /* :783-784 */
        prev_pot[e] = pot[e];                                                  /* :786 */
        float nrm = sqrtf(tt[0] * tt[0] + tt[1] * tt[1] + tt[2] * tt[2]);
        pot[e] = -nrm / c->dt;                                                 /* :787 */
        float dn = nrm < 1e-9f ? 1e-9f : nrm;                                  /* normalize() */
        float td[3] = { tt[0] / dn, tt[1] / dn, tt[2] / dn };
        float tq[4];
        quat_mul_f(r + 3, inv_start, tq);
        float up[3], hd[3];
        quat_rot_f(tq, b1, 1.0f, up);
        quat_rot_f(tq, b0, 1.0f, hd);
        float up_proj = up[2];
        float heading_proj = hd[0] * td[0] + hd[1] * td[1] + hd[2] * td[2];
        float vl[3], wl[3];
        quat_rot_f(tq, r + 7, -1.0f, vl);
        quat_rot_f(tq, r + 10, -1.0f, wl);
        /* get_euler_xyz */
        float x = tq[0], y = tq[1], z = tq[2], w = tq[3];
        float sinr = 2.0f * (w * x + y * z);
        float cosr = w * w - x * x - y * y + z * z;
        float roll = atan2f(sinr, cosr);
        float sinp = 2.0f * (w * y - z * x);
        float pitch;
        if (fabsf(sinp) >= 1.0f) {
            float sgn = (sinp > 0.0f) ? 1.0f : ((sinp < 0.0f) ? -1.0f : 0.0f);
            pitch = fabsf((float)(M_PI / 2.0)) * sgn;
        } else pitch = asinf(sinp);
        float siny = 2.0f * (w * z + x * y);
        float cosy = w * w + x * x - y * y - z * z;
        float yaw = atan2f(siny, cosy);
        roll = py_mod_f(roll, two_pi); pitch = py_mod_f(pitch, two_pi); yaw = py_mod_f(yaw, two_pi);
        float walk = atan2f(c->target[2] - r[2], c->target[0] - r[0]);        /* compute_rot */
        float ang = walk - yaw;
        o[0] = r[2];
        o[1] = vl[0]; o[2] = vl[1]; o[3] = vl[2];
        o[4] = wl[0]; o[5] = wl[1]; o[6] = wl[2];
        o[7] = yaw; o[8] = roll; o[9] = ang; o[10] = up_proj; o[11] = heading_proj;
        for (int j = 0; j < ORC_NDOF; ++j) {
            float p = dof_pos[e * ORC_NDOF + j];
            float hi = c->dof_hi[j], lo = c->dof_lo[j];
            o[12 + j] = ((2.0f * p - hi) - lo) / (hi - lo);                    /* unscale */
            o[30 + j] = dof_vel[e * ORC_NDOF + j] * c->dof_vel_scale;
            o[48 + j] = targets[e * ORC_NDOF + j];
        }
        o[66] = pitch;
        for (int k = 0; k < ORC_NLEG; ++k) {                                   /* :797 */
            const float* f = contact + (e * ORC_NCON + ORC_NABD + k) * 3;
            float s = (f[0] + f[1]) + f[2];
            o[67 + k] = (s > 0.0f) ? 1.0f : 0.0f;
        }
        if (up_vec) { up_vec[3 * e] = up[0]; up_vec[3 * e + 1] = up[1]; up_vec[3 * e + 2] = up[2]; }
        if (heading_vec) { heading_vec[3 * e] = hd[0]; heading_vec[3 * e + 1] = hd[1]; heading_vec[3 * e + 2] = hd[2]; }
    }
}

/* ---- K5: fly.py:413-443, :685-768 ---------------------------------------------------------- */
void orc_pack_reward(const OrcConfig* c, const float* obs, const float* targets, const float* root,
                     const float* contact, const float* pot, const float* prev_pot,
                     int64_t* progress, float* reward, int64_t* reset, int64_t n)
{
    const float uw = c->up_weight, hw = c->heading_weight;
    _Pragma("omp parallel for schedule(static)")
    for (int64_t e = 0; e < n; ++e) {
        const float* o = obs + e * ORC_NOBS;
        const float* act = targets + e * ORC_NDOF;
        const float* q = root + e * 13 + 3;
        if (progress[e] == 0) progress[e] = 1;                                 /* :415-416 */
        float z = o[0];
        float heading_r = (o[11] > 0.8f) ? hw : hw * o[11] / 0.8f;             /* :715-716 */
        float up_r = 0.0f;                                                     /* :719-721 */
        if (z > 1.4f) up_r = up_r + uw;
        if (z < 2.1f) up_r = up_r - uw;
        float ori = q[2] * q[2] + q[3] * q[3];                                 /* :728 */
        float orient_r = (ori > 0.98f) ? (0.0f + uw) : 0.0f;
        float actions_cost = 0.0f, elec = 0.0f;                                /* :732-733 */
        int64_t lim = 0;                                                       /* :736-737 */
        for (int j = 0; j < ORC_NDOF; ++j) {
            actions_cost += act[j] * act[j];
            elec += fabsf(act[j] - o[48 + j]);
            if (o[48 + j] > c->dof_hi[j] * 0.9f) ++lim;
            if (o[48 + j] < c->dof_lo[j] * 0.9f) ++lim;
        }
        float alive = 0.5f;                                                    /* :740 */
        float progress_r = pot[e] - prev_pot[e];                               /* :741 */
        int64_t touching = 0;                                                  /* :744 */
        for (int k = 0; k < ORC_NLEG; ++k) {
            const float* f = contact + (e * ORC_NCON + ORC_NABD + k) * 3;
            if ((f[0] + f[1]) + f[2] > 0.0f) ++touching;
        }
        float leg_r = (float)touching * 0.1f;
        float abd = 0.0f;                                                      /* :756 */
        for (int k = 0; k < ORC_NABD; ++k) {
            const float* f = contact + (e * ORC_NCON + k) * 3;
            abd += (f[0] + f[1]) + f[2];
        }
        float total;
        if (c->reward_mode == 0) {                                             /* :750 */
            total = alive + up_r * orient_r;
            total = total - c->energy_cost_scale * elec;
            total = total - (float)lim * c->joints_at_limit_cost_scale;
            total = total + leg_r;
        } else {                                                               /* :747-748 */
            total = progress_r * 2.0f + alive;
            total = total + up_r * orient_r;
            total = total + heading_r;
            total = total - c->actions_cost_scale * actions_cost;
            total = total - c->energy_cost_scale * elec;
            total = total - (float)lim * c->joints_at_limit_cost_scale;
        }
        int64_t rs = reset[e];
        if (z < c->termination_height) { total = c->death_cost; }              /* :753-756 */
        if (z > c->termination_height_up) { total = c->death_cost; }
        if (ori < 0.5f) { total = c->death_cost; }
        if (abd > 0.0f) { total = c->death_cost; }
        if (z < c->termination_height) rs = 1;                                 /* :759-766 */
        if (z > c->termination_height_up) rs = 1;
        if (progress[e] >= (int64_t)c->max_episode_length - 1) rs = 1;
        if (ori < 0.5f) rs = 1;
        if (abd > 0.0f) rs = 1;
        reward[e] = total;
        reset[e] = rs;
    }
}

/* ---- f4: the reference viewer's P-key dump of the reward terms, fly.py:504-546 ---------------
 * terms [n][9] = heading_reward, alive_reward, up_reward, orient_reward, actions_cost, electricity_cost,
 * dof_at_limit_cost (already times joints_at_limit_cost_scale, :531), progress_reward, leg_reward.
 * NOT the reward's own terms: the dump's up_reward has no "- up_weight below 2.1" branch (:512-513 vs :719-721)
 * and its orientation threshold is 0.92 (:518), the reward's is 0.98 (:728). */
void orc_reward_terms(const OrcConfig* c, const float* obs, const float* targets, const float* root,
                      const float* contact, const float* pot, const float* prev_pot, float* terms, int64_t n)
{
    const float uw = c->up_weight, hw = c->heading_weight;
    _Pragma("omp parallel for schedule(static)")
    for (int64_t e = 0; e < n; ++e) {
        const float* o = obs + e * ORC_NOBS;
        const float* act = targets + e * ORC_NDOF;
        const float* q = root + e * 13 + 3;
        float* t = terms + e * 9;
        t[0] = (o[11] > 0.8f) ? hw : hw * o[11] / 0.8f;                        /* :507-508 */
        t[1] = 0.5f;                                                           /* :510 */
        t[2] = (o[0] > 1.4f) ? (0.0f + uw) : 0.0f;                             /* :512-513 */
        t[3] = (q[2] * q[2] + q[3] * q[3] > 0.92f) ? (0.0f + uw) : 0.0f;       /* :517-518 */
        float ac = 0.0f, elec = 0.0f;
        int64_t lim = 0;
        for (int j = 0; j < ORC_NDOF; ++j) {
            ac += act[j] * act[j];                                             /* :521 */
            elec += fabsf(act[j] - o[48 + j]);                                 /* :524-525 */
            if (o[48 + j] > c->dof_hi[j] * 0.9f) ++lim;                        /* :529 */
            if (o[48 + j] < c->dof_lo[j] * 0.9f) ++lim;                        /* :530 */
        }
        t[4] = ac; t[5] = elec;
        t[6] = (float)lim * c->joints_at_limit_cost_scale;                     /* :531 */
        t[7] = pot[e] - prev_pot[e];                                           /* :532 */
        int64_t touching = 0;                                                  /* :544 */
        for (int k = 0; k < ORC_NLEG; ++k) {
            const float* f = contact + (e * ORC_NCON + ORC_NABD + k) * 3;
            if ((f[0] + f[1]) + f[2] > 0.0f) ++touching;
        }
        t[8] = (float)touching * 0.1f;
    }
}

/* ---- a1: Fly.step orchestration, fly.py:624-681 (flyLowGrav.py swaps reset/simulate) ------- */
void orc_env_step(const OrcConfig* c, const float* actions, float* root, float* dof_pos,
                  float* dof_vel, float* targets, float* contact, float* pot, float* prev_pot,
                  float* obs, float* reward, int64_t* reset, int64_t* progress, int64_t n)
{
    orc_scale_actions(c, actions, targets, n);                                 /* :626-657 */
    if (!c->reset_after_sim)
        orc_reset_masked(c, root, dof_pos, dof_vel, pot, prev_pot, reset, progress, n); /* :660 */
    orc_physics_step(c, root, dof_pos, dof_vel, targets, contact, n);          /* :663 */
    if (c->reset_after_sim)
        orc_reset_masked(c, root, dof_pos, dof_vel, pot, prev_pot, reset, progress, n);
    orc_pack_obs(c, root, dof_pos, dof_vel, targets, contact, pot, prev_pot, obs, 0, 0, n); /* :676 */
    for (int64_t e = 0; e < n; ++e) progress[e] += 1;                          /* :678 */
    orc_pack_reward(c, obs, targets, root, contact, pot, prev_pot, progress, reward, reset, n); /* :681 */
}

/* ---- a8: ppo.py:213-220 with MultivariateNormal(mu, scale_tril=cholesky(diag(var))) --------- */
void orc_sample_logprob(const float* mu, const float* var, const float* eps, float* act,
                        float* logp, int64_t n)
{
    float L[ORC_NDOF], half_log_det = 0.0f;
    for (int j = 0; j < ORC_NDOF; ++j) { L[j] = sqrtf(var[j]); half_log_det += logf(L[j]); }
    const float klog2pi = (float)(ORC_NDOF * log(2.0 * M_PI));
    for (int64_t e = 0; e < n; ++e) {
        float M = 0.0f;
        for (int j = 0; j < ORC_NDOF; ++j) {
            float a = mu[e * ORC_NDOF + j] + L[j] * eps[e * ORC_NDOF + j];  /* rsample */
            float x = (a - mu[e * ORC_NDOF + j]) / L[j];                     /* mahalanobis */
            M += x * x;
            float cl = a < -1.0f ? -1.0f : (a > 1.0f ? 1.0f : a);            /* :220 */
            act[e * ORC_NDOF + j] = cl;
        }
        logp[e] = -0.5f * (klog2pi + M) - half_log_det;
    }
}

/* ---- a10: ppo.py:157-171 --------------------------------------------------------------------- */
void orc_td_gae(const float* reward, const float* v, const float* v_next, const float* done,
                float gamma, float lambda, int64_t T, int64_t N, float* target, float* adv,
                int mode_flags)
{
    const float gl = (float)((double)gamma * (double)lambda);  /* python double product, :167 */
    for (int64_t e = 0; e < N; ++e) {
        float a = 0.0f;
        for (int64_t t = T - 1; t >= 0; --t) {
            int64_t i = t * N + e;
            float d = (mode_flags & 1) ? done[i] : done[e];      /* Q1: [N] mask broadcast */
            float tg = reward[i] + gamma * v_next[i] * d;        /* :160 */
            float delta = tg - v[i];                             /* :161 */
            float carry = (mode_flags & 2) ? a * d : a;
            a = gl * carry + delta;                              /* :167 */
            target[i] = tg;
            adv[i] = a;
        }
    }
}

/* ---- a7: ppo.py:10-102.  w/b order: shared0, shared2, mean0, mean2, value0, value2 ---------- */
static float elu_f(float x) { return x > 0.0f ? x : expm1f(x); }

static void linear_f(const float* w, const float* b, const float* x, int in, int out, float* y, int act)
{
    for (int o = 0; o < out; ++o) {
        float s = 0.0f;
        for (int i = 0; i < in; ++i) s += x[i] * w[o * in + i];
        s += b[o];
        y[o] = act ? elu_f(s) : s;
    }
}

void orc_net_forward(const float* const* w, const float* const* b, const float* x, int64_t n,
                     int head, float* out)
{
    float h1[256], h2[128], h3[64];
    for (int64_t e = 0; e < n; ++e) {
        linear_f(w[0], b[0], x + e * ORC_NOBS, ORC_NOBS, 256, h1, 1);
        linear_f(w[1], b[1], h1, 256, 128, h2, 1);
        if (head == 0) {
            linear_f(w[2], b[2], h2, 128, 64, h3, 1);
            linear_f(w[3], b[3], h3, 64, ORC_NDOF, out + e * ORC_NDOF, 1);   /* ELU on the mean, ppo.py:30 */
        } else {
            linear_f(w[4], b[4], h2, 128, 64, h3, 1);
            linear_f(w[5], b[5], h3, 64, 1, out + e, 0);
        }
    }
}

/* ---- a13: UselessFiles/dqn.py:89-100 (act) and :64-85 (update's TD loss) -------------------- */
void orc_dqn_eps_greedy(const float* q, const float* coin_u, const float* rand_u, float epsilon, int A,
                        float* act_out, int64_t n)
{
    for (int64_t e = 0; e < n; ++e) {
        const float* row = q + e * A;
        int idx = 0;
        for (int a = 1; a < A; ++a) if (row[a] > row[idx]) idx = a;      /* first maximal entry, :95-96 */
        float true_act = (float)idx / (float)(A - 1);                      /* :97 */
        float coin = coin_u[e] < epsilon ? 1.0f : 0.0f;                    /* :90 */
        float act = coin * rand_u[e] + (1.0f - coin) * true_act;           /* :99 */
        act_out[e] = 2.0f * (act - 0.5f);                                  /* :100 */
    }
}

void orc_dqn_huber_td(const float* q_table, const float* act, const float* reward, const float* q_next,
                      const float* done, float discount, int A, int64_t B, float* dq, float* loss_out)
{
    double loss = 0.0;
    for (int64_t b = 0; b < B; ++b) {
        float a01 = 0.5f * (act[b] + 1.0f);
        int idx = (int)rintf(a01 * (float)(A - 1));                        /* torch.round, :71 */
        float mx = q_next[b * A];
        for (int a = 1; a < A; ++a) if (q_next[b * A + a] > mx) mx = q_next[b * A + a];   /* :75 */
        float target = reward[b] + discount * mx * done[b];                /* :77 */
        float d = q_table[b * A + idx] - target;
        float h = fabsf(d) < 1.0f ? 0.5f * d * d : fabsf(d) - 0.5f;        /* smooth_l1, :78 */
        loss += h;
        for (int a = 0; a < A; ++a) dq[b * A + a] = 0.0f;
        float g = d < -1.0f ? -1.0f : (d > 1.0f ? 1.0f : d);
        dq[b * A + idx] = g / (float)B;
    }
    *loss_out = (float)(loss / (double)B);
}
