// fly_env.hip — gfx950 kernels for the vectorised Fly environment step.
//
// Replaces, per launch, the torch op chains and Isaac Gym calls of the reference's Fly.step
// (fly.py:624-681): K1 action scale (:626-657), K2 masked reset (:446-480), K3 physics
// (:482-485, build-defined FlyDyn), K4 observation pack (:771-805),
// progress += 1 (:678), K5 reward/done pack (:685-768).
//
// Mapping (CDNA4, 64-wide waves): 8 lanes own one env, so a wave carries 8 envs and a
// 256-thread workgroup 32.  Inside an env's 8-lane half of a DPP row
//     lanes 0..5   one leg each: 3 PD joints, leg kinematics, tip contact        (contact pass A)
//     lanes 0..4   ALSO one abdomen contact point each                          (contact pass B)
//     lanes 6..7   idle in the contact passes (they still hold the replicated root state)
// The kernel is VALU-issue-bound (15 substeps x ~400 instructions per wave), so what counts is
// instructions per env: the contact code runs twice per substep, but a wave now advances 8 envs
// instead of 4 -- 1.5x fewer instructions per env-substep than one point per lane on 16 lanes
// (11 of 16 lanes busy, root integration replicated 16x).
// The body wrench is an all-reduce over the 8 lanes with three DPP adds per component
// (quad_perm xor1, quad_perm xor2, row_half_mirror); the tree is symmetric so all 8 lanes end
// with bit-identical sums and integrate the replicated root state identically.
// All role differences are selects, not branches: a wave never diverges inside the substep loop.
// Substeps run in registers; HBM is touched once on the way in and once on the way out.
// Observation rows (row-major [N][73], a GEMM operand for the policy) are assembled in an LDS
// tile and written as one contiguous 16x73 block per workgroup with 16-byte stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flyhip.h"

namespace {

constexpr int LANES_PER_ENV = 8;
constexpr int BLOCK = 256;
constexpr int ENVS_PER_BLOCK = BLOCK / LANES_PER_ENV;

enum : int { PH_SCALE = 1, PH_RESET = 2, PH_INTEGRATE = 4, PH_OBS = 8, PH_REWARD = 16, PH_PROGRESS = 32 };

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
}
// all-reduce over an env's 8 lanes (half a DPP row); every lane gets the same bits
__device__ __forceinline__ float row_sum(float x)
{
    x += dpp_f<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_f<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_f<0x141>(x);  // row_half_mirror
    return x;
}
__device__ __forceinline__ int row_sum_i(int x)
{
    x += dpp_i<0xB1>(x);
    x += dpp_i<0x4E>(x);
    x += dpp_i<0x141>(x);
    return x;
}
// value held by lane `src` (0..15) of this lane's row
__device__ __forceinline__ float row_get(float x, int src)
{
    return __shfl(x, src, LANES_PER_ENV);
}

__device__ __forceinline__ float dot3(float a0, float a1, float a2, float b0, float b1, float b2)
{
    return fmaf(a0, b0, fmaf(a1, b1, a2 * b2));
}
__device__ __forceinline__ float clampf(float x, float lim) { return fminf(fmaxf(x, -lim), lim); }

struct Root {
    float px, py, pz, qx, qy, qz, qw, vx, vy, vz, wx, wy, wz;
};

// ---- K4 helpers: separately rounded fp32 (the file is built with -ffp-contract=off; every fused
// multiply-add in the physics is an explicit fmaf, so all template instances round identically) --
__device__ __forceinline__ void quat_rot(const float q[4], float v0, float v1, float v2, float sign, float o[3])
{
    float qw = q[3];
    float s = 2.0f * (qw * qw) - 1.0f;
    float cx = q[1] * v2 - q[2] * v1;
    float cy = q[2] * v0 - q[0] * v2;
    float cz = q[0] * v1 - q[1] * v0;
    float dot = q[0] * v0 + q[1] * v1 + q[2] * v2;
    o[0] = (v0 * s + sign * (cx * qw * 2.0f)) + q[0] * dot * 2.0f;
    o[1] = (v1 * s + sign * (cy * qw * 2.0f)) + q[1] * dot * 2.0f;
    o[2] = (v2 * s + sign * (cz * qw * 2.0f)) + q[2] * dot * 2.0f;
}

__device__ __forceinline__ float py_mod(float a, float b)
{
    float m = fmodf(a, b);
    if (m != 0.0f && ((b < 0.0f) != (m < 0.0f))) m += b;
    return m;
}

struct RootObs {
    float vl[3], wl[3], yaw, roll, pitch, ang, up_proj, heading_proj, pot;
};

// fly.py:771-805 root-derived part (compute_heading_and_up, compute_rot, get_euler_xyz)
__device__ __forceinline__ RootObs root_obs(const FlyConfig* __restrict__ c, const Root& r)
{
    RootObs o;
    const float two_pi = 6.28318530717958647692f;
    float t0 = c->target[0] - r.px, t1 = c->target[1] - r.py, t2 = 0.0f;
    float nrm = sqrtf(t0 * t0 + t1 * t1 + t2 * t2);
    o.pot = -nrm / c->dt;
    float dn = nrm < 1e-9f ? 1e-9f : nrm;
    float d0 = t0 / dn, d1 = t1 / dn, d2 = t2 / dn;
    // quat_mul(torso_rotation, inv_start_rot), inv_start_rot = conj(0,0,0,1)
    float x1 = r.qx, y1 = r.qy, z1 = r.qz, w1 = r.qw;
    float x2 = -0.0f, y2 = -0.0f, z2 = -0.0f, w2 = 1.0f;
    float ww = (z1 + x1) * (x2 + y2);
    float yy = (w1 - y1) * (w2 + z2);
    float zz = (w1 + y1) * (w2 - z2);
    float xx = ww + yy + zz;
    float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
    float tq[4];
    tq[3] = qq - ww + (z1 - y1) * (y2 - z2);
    tq[0] = qq - xx + (x1 + w1) * (x2 + w2);
    tq[1] = qq - yy + (w1 - x1) * (y2 + z2);
    tq[2] = qq - zz + (z1 + y1) * (w2 - x2);
    float up[3], hd[3];
    quat_rot(tq, 0.0f, 0.0f, 1.0f, 1.0f, up);
    quat_rot(tq, 1.0f, 0.0f, 0.0f, 1.0f, hd);
    o.up_proj = up[2];
    o.heading_proj = hd[0] * d0 + hd[1] * d1 + hd[2] * d2;
    quat_rot(tq, r.vx, r.vy, r.vz, -1.0f, o.vl);
    quat_rot(tq, r.wx, r.wy, r.wz, -1.0f, o.wl);
    float x = tq[0], y = tq[1], z = tq[2], w = tq[3];
    float sinr = 2.0f * (w * x + y * z);
    float cosr = w * w - x * x - y * y + z * z;
    float roll = atan2f(sinr, cosr);
    float sinp = 2.0f * (w * y - z * x);
    float pitch;
    if (fabsf(sinp) >= 1.0f) {
        float sgn = (sinp > 0.0f) ? 1.0f : ((sinp < 0.0f) ? -1.0f : 0.0f);
        pitch = 1.57079632679489661923f * sgn;
    } else {
        pitch = asinf(sinp);
    }
    float siny = 2.0f * (w * z + x * y);
    float cosy = w * w + x * x - y * y - z * z;
    float yaw = atan2f(siny, cosy);
    o.roll = py_mod(roll, two_pi);
    o.pitch = py_mod(pitch, two_pi);
    o.yaw = py_mod(yaw, two_pi);
    float walk = atan2f(c->target[2] - r.pz, c->target[0] - r.px);
    o.ang = walk - o.yaw;
    return o;
}

// isaacgym scale(): 0.5*(x+1.0)*(hi-lo)+lo, fly.py:629
__device__ __forceinline__ float scale_action(float a, float lo, float hi)
{
    float t = 0.5f * (a + 1.0f);
    t = t * (hi - lo);
    return t + lo;
}

template <int PH>
__global__ __launch_bounds__(BLOCK) void fly_kernel(const FlyConfig* __restrict__ c,
                                                    const float* __restrict__ actions, FlyBuffers b)
{
    __shared__ __attribute__((aligned(16))) float obs_tile[ENVS_PER_BLOCK * FLY_NUM_OBS];

    const int n = c->num_envs;
    const int tid = threadIdx.x;
    const int sub = tid & (LANES_PER_ENV - 1);
    const int env_in_blk = tid / LANES_PER_ENV;
    const long e_raw = (long)blockIdx.x * ENVS_PER_BLOCK + env_in_blk;
    const bool valid = e_raw < n;
    const long e = valid ? e_raw : (long)(n - 1);   // tail lanes shadow the last env; stores are masked
    const bool is_leg = sub < FLY_NUM_LEGS;
    const bool is_abd = sub < FLY_NUM_ABDOMEN;             // the first five lanes also own an abdomen point
    const int leg = is_leg ? sub : 0;
    const int abd = is_abd ? sub : 0;
    const int j0 = 3 * leg;

    // per-lane tables
    float lo[3], hi[3], pose[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { lo[i] = c->dof_lo[j0 + i]; hi[i] = c->dof_hi[j0 + i]; pose[i] = c->dof_pose[j0 + i]; }

    // ---- load state (only what the enabled phases touch) ---------------------------------------
    constexpr bool NEED_ROOT = (PH & (PH_RESET | PH_INTEGRATE | PH_OBS | PH_REWARD)) != 0;
    constexpr bool NEED_DOF = (PH & (PH_RESET | PH_INTEGRATE | PH_OBS)) != 0;
    constexpr bool NEED_POT = (PH & (PH_RESET | PH_OBS | PH_REWARD)) != 0;
    constexpr bool NEED_FLAG = (PH & (PH_RESET | PH_REWARD)) != 0;
    constexpr bool NEED_PROG = (PH & (PH_RESET | PH_REWARD | PH_PROGRESS)) != 0;
    Root r = {};
    if (NEED_ROOT) {
        const float* rp = b.root + e * FLY_ROOT_DIM;
        r.px = rp[0]; r.py = rp[1]; r.pz = rp[2]; r.qx = rp[3]; r.qy = rp[4]; r.qz = rp[5]; r.qw = rp[6];
        r.vx = rp[7]; r.vy = rp[8]; r.vz = rp[9]; r.wx = rp[10]; r.wy = rp[11]; r.wz = rp[12];
    }
    float jq[3] = {0, 0, 0}, jqd[3] = {0, 0, 0}, jt[3] = {0, 0, 0};
    if (NEED_DOF && is_leg) {
        const float* dp = b.dof_state + e * (FLY_NUM_DOF * 2) + 2 * j0;
#pragma unroll
        for (int i = 0; i < 3; ++i) { jq[i] = dp[2 * i]; jqd[i] = dp[2 * i + 1]; }
    }
    float pot = 0.0f, prev_pot = 0.0f;
    if (NEED_POT) { pot = b.pot[e]; prev_pot = b.prev_pot[e]; }
    int rs = 0;
    if (NEED_FLAG) rs = (int)(b.reset[e] != 0);
    long progress = 0;
    if (NEED_PROG) progress = b.progress[e];
    float cf[3] = {0, 0, 0};   // this lane's leg-tip contact force (lanes 0..5)
    float cfa[3] = {0, 0, 0};  // this lane's abdomen-point contact force (lanes 0..4)

    // ---- K1: targets -------------------------------------------------------------------------
    if (PH & PH_SCALE) {
        if (is_leg) {
            const float* ap = actions + e * FLY_NUM_DOF + j0;
#pragma unroll
            for (int i = 0; i < 3; ++i) jt[i] = scale_action(ap[i], lo[i], hi[i]);
        }
    } else if ((PH & (PH_INTEGRATE | PH_OBS | PH_REWARD)) && is_leg) {
        const float* tp = b.targets + e * FLY_NUM_DOF + j0;
#pragma unroll
        for (int i = 0; i < 3; ++i) jt[i] = tp[i];
    }
    if (!(PH & (PH_INTEGRATE)) && (PH & (PH_OBS | PH_REWARD))) {
        // unfused packs read the contact forces the integrator left in HBM
        if (is_leg) {
            const float* fp = b.contact + (e * FLY_NUM_CONTACT + FLY_NUM_ABDOMEN + leg) * 3;
            cf[0] = fp[0]; cf[1] = fp[1]; cf[2] = fp[2];
        }
        if (is_abd) {
            const float* fp = b.contact + (e * FLY_NUM_CONTACT + abd) * 3;
            cfa[0] = fp[0]; cfa[1] = fp[1]; cfa[2] = fp[2];
        }
    }

    const bool reset_after = c->reset_after_sim != 0;

    auto do_reset = [&]() {
        // fly.py:446-480: flagged envs go back to the reset pose; potentials recomputed; flags cleared
        if (rs) {
            r.px = 0.0f; r.py = 0.0f; r.pz = c->start_height; r.qx = 0.0f; r.qy = 0.0f; r.qz = 0.0f; r.qw = 1.0f;
            r.vx = r.vy = r.vz = r.wx = r.wy = r.wz = 0.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i) { jq[i] = is_leg ? pose[i] : 0.0f; jqd[i] = 0.0f; }
            float tx = c->target[0] - 0.0f, ty = c->target[1] - 0.0f;
            float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(tx, tx), __fmul_rn(ty, ty)), 0.0f));
            float p = -nrm / c->dt;
            prev_pot = p; pot = p;
            rs = 0; progress = 0;
        }
    };

    if ((PH & PH_RESET) && !(reset_after && (PH & PH_INTEGRATE))) do_reset();

    // ---- K3: FlyDyn substeps in registers ----------------------------------------------------
    if (PH & PH_INTEGRATE) {
        const int nsub = c->substeps;
        const float h = c->dt / (float)nsub;
        const float kp = c->kp, kd = c->kd, eff = c->effort, vmax = c->vmax, Jinv = 1.0f / c->joint_inertia;
        const float minv = 1.0f / c->mass, g = c->gravity;
        const float I0 = c->inertia[0], I1 = c->inertia[1], I2 = c->inertia[2];
        const float I0inv = 1.0f / I0, I1inv = 1.0f / I1, I2inv = 1.0f / I2;
        const float vlim = c->max_lin_vel, wlim = c->max_ang_vel;
        const float kc = c->kc, cd = c->cdamp, mu = c->mu, cv = c->cvisc;
        const float Lf = c->femur_len, Lt = c->tibia_len;
        const float ld = 1.0f - h * c->lin_damp, ad = 1.0f - h * c->ang_damp;
        const float att0 = c->leg_attach[leg][0], att1 = c->leg_attach[leg][1], att2 = c->leg_attach[leg][2];
        const float azim = c->leg_azimuth[leg], sg = c->leg_sigma[leg];
        const float ab0 = c->abdomen_pts[abd][0], ab1 = c->abdomen_pts[abd][1], ab2 = c->abdomen_pts[abd][2];
        const float al0 = c->alpha0, be0 = c->beta0;

        for (int s = 0; s < nsub; ++s) {
            // 1. joints (leg lanes; the others integrate zeros)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float tau = fmaf(kp, jt[i] - jq[i], -(kd * jqd[i]));
                tau = fminf(fmaxf(tau, -eff), eff);
                float v = fmaf(h * Jinv, tau, jqd[i]);
                v = fminf(fmaxf(v, -vmax), vmax);
                float x = fmaf(h, v, jq[i]);
                if (x < lo[i]) { x = lo[i]; v = fmaxf(v, 0.0f); }
                if (x > hi[i]) { x = hi[i]; v = fminf(v, 0.0f); }
                jq[i] = x; jqd[i] = v;
            }
            // rotation matrix
            const float qx = r.qx, qy = r.qy, qz = r.qz, qw = r.qw;
            const float R00 = fmaf(-2.0f, fmaf(qy, qy, qz * qz), 1.0f), R01 = 2.0f * fmaf(qx, qy, -(qz * qw)), R02 = 2.0f * fmaf(qx, qz, qy * qw);
            const float R10 = 2.0f * fmaf(qx, qy, qz * qw), R11 = fmaf(-2.0f, fmaf(qx, qx, qz * qz), 1.0f), R12 = 2.0f * fmaf(qy, qz, -(qx * qw));
            const float R20 = 2.0f * fmaf(qx, qz, -(qy * qw)), R21 = 2.0f * fmaf(qy, qz, qx * qw), R22 = fmaf(-2.0f, fmaf(qx, qx, qy * qy), 1.0f);
            // 2. this lane's contact point: leg kinematics, or a fixed abdomen point
            float psi = fmaf(sg, jq[0] - pose[0], azim);
            float al = al0 + (jq[1] - pose[1]);
            float gm = al + be0 + (jq[2] - pose[2]);
            float psid = sg * jqd[0], ald = jqd[1], gmd = ald + jqd[2];
            // hardware sin/cos (v_sin_f32 / v_cos_f32 on x/2pi): the leg angles are bounded by the
            // joint limits (|x| < 16 rad), where the absolute error stays ~1e-6, far inside the
            // stated 2e-4 one-step tolerance; the libm versions cost ~70 instructions each, 15x per step.
            const float sp = __sinf(psi), cp = __cosf(psi);
            const float sa = __sinf(al), ca = __cosf(al);
            const float sgm = __sinf(gm), cg = __cosf(gm);
            float rho = fmaf(Lf, ca, Lt * cg), zeta = fmaf(Lf, sa, Lt * sgm);
            float rhod = -fmaf(Lf * sa, ald, Lt * sgm * gmd);
            float zetad = fmaf(Lf * ca, ald, Lt * cg * gmd);
            // one contact point: world offset, point velocity, penalty normal force, capped viscous
            // friction; returns the force and accumulates its torque about the root
            float Fx_l = 0.0f, Fy_l = 0.0f, Fz_l = 0.0f, Tx_l = 0.0f, Ty_l = 0.0f, Tz_l = 0.0f;
            auto contact = [&](float rbx, float rby, float rbz, float rdx, float rdy, float rdz, bool lane_on, float (&f)[3]) {
                float rwx = dot3(R00, R01, R02, rbx, rby, rbz);
                float rwy = dot3(R10, R11, R12, rbx, rby, rbz);
                float rwz = dot3(R20, R21, R22, rbx, rby, rbz);
                float d = -(r.pz + rwz);
                float ux = r.vx + fmaf(r.wy, rwz, -(r.wz * rwy)) + dot3(R00, R01, R02, rdx, rdy, rdz);
                float uy = r.vy + fmaf(r.wz, rwx, -(r.wx * rwz)) + dot3(R10, R11, R12, rdx, rdy, rdz);
                float uz = r.vz + fmaf(r.wx, rwy, -(r.wy * rwx)) + dot3(R20, R21, R22, rdx, rdy, rdz);
                float fn = fmaxf(kc * d * fmaf(-cd, uz, 1.0f), 0.0f);
                float ut = __builtin_amdgcn_sqrtf(fmaf(ux, ux, uy * uy));      // 1-ulp hardware sqrt / rcp
                float ft = fminf(cv * ut, mu * fn);
                float sc = ft * __builtin_amdgcn_rcpf(ut + 1e-9f);
                const bool touch = lane_on && (d > 0.0f);
                float fx = touch ? -sc * ux : 0.0f;
                float fy = touch ? -sc * uy : 0.0f;
                float fz = touch ? fn : 0.0f;
                f[0] = fx; f[1] = fy; f[2] = fz;
                Fx_l += fx; Fy_l += fy; Fz_l += fz;
                Tx_l += fmaf(rwy, fz, -(rwz * fy));
                Ty_l += fmaf(rwz, fx, -(rwx * fz));
                Tz_l += fmaf(rwx, fy, -(rwy * fx));
            };
            // pass A: the leg tip (moving with the joints); pass B: the fixed abdomen point
            contact(fmaf(cp, rho, att0), fmaf(sp, rho, att1), att2 + zeta,
                    fmaf(cp, rhod, -(sp * rho * psid)), fmaf(sp, rhod, cp * rho * psid), zetad, is_leg, cf);
            contact(ab0, ab1, ab2, 0.0f, 0.0f, 0.0f, is_abd, cfa);
            float Fx = row_sum(Fx_l), Fy = row_sum(Fy_l), Fz = row_sum(Fz_l);
            float Tx = row_sum(Tx_l), Ty = row_sum(Ty_l), Tz = row_sum(Tz_l);
            // 3. root, semi-implicit Euler (replicated over the row)
            r.vx = clampf(fmaf(h, Fx * minv, r.vx) * ld, vlim);
            r.vy = clampf(fmaf(h, Fy * minv, r.vy) * ld, vlim);
            r.vz = clampf(fmaf(h, fmaf(Fz, minv, g), r.vz) * ld, vlim);
            float wbx = dot3(R00, R10, R20, r.wx, r.wy, r.wz);
            float wby = dot3(R01, R11, R21, r.wx, r.wy, r.wz);
            float wbz = dot3(R02, R12, R22, r.wx, r.wy, r.wz);
            float tbx = dot3(R00, R10, R20, Tx, Ty, Tz);
            float tby = dot3(R01, R11, R21, Tx, Ty, Tz);
            float tbz = dot3(R02, R12, R22, Tx, Ty, Tz);
            float ax = (tbx - fmaf(wby * I2, wbz, -(wbz * I1 * wby))) * I0inv;
            float ay = (tby - fmaf(wbz * I0, wbx, -(wbx * I2 * wbz))) * I1inv;
            float az = (tbz - fmaf(wbx * I1, wby, -(wby * I0 * wbx))) * I2inv;
            wbx = clampf(fmaf(h, ax, wbx) * ad, wlim);
            wby = clampf(fmaf(h, ay, wby) * ad, wlim);
            wbz = clampf(fmaf(h, az, wbz) * ad, wlim);
            r.wx = dot3(R00, R01, R02, wbx, wby, wbz);
            r.wy = dot3(R10, R11, R12, wbx, wby, wbz);
            r.wz = dot3(R20, R21, R22, wbx, wby, wbz);
            r.px = fmaf(h, r.vx, r.px); r.py = fmaf(h, r.vy, r.py); r.pz = fmaf(h, r.vz, r.pz);
            const float hh = 0.5f * h;
            float nqx = fmaf(hh, fmaf(r.wx, qw, fmaf(r.wy, qz, -(r.wz * qy))), qx);
            float nqy = fmaf(hh, fmaf(-r.wx, qz, fmaf(r.wy, qw, r.wz * qx)), qy);
            float nqz = fmaf(hh, fmaf(r.wx, qy, fmaf(-r.wy, qx, r.wz * qw)), qz);
            float nqw = fmaf(hh, -fmaf(r.wx, qx, fmaf(r.wy, qy, r.wz * qz)), qw);
            float inv = __builtin_amdgcn_rsqf(fmaf(nqx, nqx, fmaf(nqy, nqy, fmaf(nqz, nqz, nqw * nqw))));
            r.qx = nqx * inv; r.qy = nqy * inv; r.qz = nqz * inv; r.qw = nqw * inv;
        }
    }

    if ((PH & PH_RESET) && reset_after && (PH & PH_INTEGRATE)) do_reset();

    // ---- K4: observation row -----------------------------------------------------------------
    float touching_f = 0.0f;   // this leg lane's obs[67+leg]
    float obs_act[3] = {0, 0, 0};
    float z_obs = 0.0f, heading_proj = 0.0f;
    if (PH & PH_OBS) {
        prev_pot = pot;                                   // fly.py:786
        RootObs ro = root_obs(c, r);
        pot = ro.pot;                                     // fly.py:787
        float* row = obs_tile + env_in_blk * FLY_NUM_OBS;
        if (sub == LANES_PER_ENV - 1) {
            row[0] = r.pz; row[1] = ro.vl[0]; row[2] = ro.vl[1]; row[3] = ro.vl[2];
            row[4] = ro.wl[0]; row[5] = ro.wl[1]; row[6] = ro.wl[2];
            row[7] = ro.yaw; row[8] = ro.roll; row[9] = ro.ang; row[10] = ro.up_proj; row[11] = ro.heading_proj;
            row[66] = ro.pitch;
        }
        z_obs = r.pz; heading_proj = ro.heading_proj;
        if (is_leg) {
            const float vs = c->dof_vel_scale;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                // unscale(): (2x - hi - lo)/(hi - lo), separately rounded
                float u = __fsub_rn(__fsub_rn(__fmul_rn(2.0f, jq[i]), hi[i]), lo[i]) / __fsub_rn(hi[i], lo[i]);
                row[12 + j0 + i] = u;
                row[30 + j0 + i] = __fmul_rn(jqd[i], vs);
                row[48 + j0 + i] = jt[i];
                obs_act[i] = jt[i];
            }
            float ssum = __fadd_rn(__fadd_rn(cf[0], cf[1]), cf[2]);   // fly.py:797: sum of components
            touching_f = ssum > 0.0f ? 1.0f : 0.0f;
            row[67 + leg] = touching_f;
        }
        __syncthreads();
        // one contiguous 32 x 73 tile per workgroup, 16-byte stores
        const long tile_base = (long)blockIdx.x * ENVS_PER_BLOCK * FLY_NUM_OBS;
        const long total = (long)n * FLY_NUM_OBS;
        const float4* src4 = reinterpret_cast<const float4*>(obs_tile);
        const bool aligned16 = (reinterpret_cast<uintptr_t>(b.obs) & 15) == 0;   // wave-uniform
        for (int i = tid; i < ENVS_PER_BLOCK * FLY_NUM_OBS / 4; i += BLOCK) {
            long g = tile_base + 4L * i;
            if (aligned16 && g + 3 < total) {
                *reinterpret_cast<float4*>(b.obs + g) = src4[i];
            } else {
                for (int k = 0; k < 4; ++k)
                    if (g + k < total) b.obs[g + k] = obs_tile[4 * i + k];
            }
        }
    } else if (PH & PH_REWARD) {
        // unfused reward reads the observation row the pack left in HBM
        const float* row = b.obs + e * FLY_NUM_OBS;
        z_obs = row[0]; heading_proj = row[11];
        if (is_leg) {
#pragma unroll
            for (int i = 0; i < 3; ++i) obs_act[i] = row[48 + j0 + i];
        }
    }

    if (PH & PH_PROGRESS) progress += 1;                   // fly.py:678

    // ---- K5: reward and done mask -------------------------------------------------------------
    float reward = 0.0f;
    if (PH & PH_REWARD) {
        if (progress == 0) progress = 1;                   // fly.py:415-416
        const float uw = c->up_weight, hw = c->heading_weight;
        float elec = 0.0f, acost = 0.0f;
        int lim = 0, touching = 0;
        if (is_leg) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                acost += jt[i] * jt[i];
                elec += fabsf(__fsub_rn(jt[i], obs_act[i]));
                lim += (obs_act[i] > __fmul_rn(hi[i], 0.9f)) ? 1 : 0;
                lim += (obs_act[i] < __fmul_rn(lo[i], 0.9f)) ? 1 : 0;
            }
            touching = (__fadd_rn(__fadd_rn(cf[0], cf[1]), cf[2]) > 0.0f) ? 1 : 0;
        }
        elec = row_sum(elec); acost = row_sum(acost);
        lim = row_sum_i(lim); touching = row_sum_i(touching);
        // abdomen: per-body component sums added in body order (fly.py:756), bit-stable mask
        float bsum = is_abd ? __fadd_rn(__fadd_rn(cfa[0], cfa[1]), cfa[2]) : 0.0f;
        float abd_sum = 0.0f;
#pragma unroll
        for (int k = 0; k < FLY_NUM_ABDOMEN; ++k) abd_sum = __fadd_rn(abd_sum, row_get(bsum, k));
        const float z = z_obs;
        float heading_r = (heading_proj > 0.8f) ? hw : __fmul_rn(hw, heading_proj) / 0.8f;
        float up_r = 0.0f;
        if (z > 1.4f) up_r = __fadd_rn(up_r, uw);
        if (z < 2.1f) up_r = __fsub_rn(up_r, uw);
        float ori = __fadd_rn(__fmul_rn(r.qz, r.qz), __fmul_rn(r.qw, r.qw));
        float orient_r = (ori > 0.98f) ? uw : 0.0f;
        float leg_r = __fmul_rn((float)touching, 0.1f);
        float progress_r = __fsub_rn(pot, prev_pot);
        float total;
        if (c->reward_mode == 0) {                         // fly.py:750
            total = __fadd_rn(0.5f, __fmul_rn(up_r, orient_r));
            total = __fsub_rn(total, __fmul_rn(c->energy_cost_scale, elec));
            total = __fsub_rn(total, __fmul_rn((float)lim, c->joints_at_limit_cost_scale));
            total = __fadd_rn(total, leg_r);
        } else {                                           // fly.py:747-748
            total = __fadd_rn(__fmul_rn(progress_r, 2.0f), 0.5f);
            total = __fadd_rn(total, __fmul_rn(up_r, orient_r));
            total = __fadd_rn(total, heading_r);
            total = __fsub_rn(total, __fmul_rn(c->actions_cost_scale, acost));
            total = __fsub_rn(total, __fmul_rn(c->energy_cost_scale, elec));
            total = __fsub_rn(total, __fmul_rn((float)lim, c->joints_at_limit_cost_scale));
        }
        const bool dead = (z < c->termination_height) || (z > c->termination_height_up) || (ori < 0.5f) || (abd_sum > 0.0f);
        if (dead) total = c->death_cost;                   // fly.py:753-756
        if (dead || progress >= (long)c->max_episode_length - 1) rs = 1;   // fly.py:759-766
        reward = total;
    }

    // ---- store --------------------------------------------------------------------------------
    if (!valid) return;
    if (PH & (PH_SCALE)) {
        if (is_leg) {
            float* tp = b.targets + e * FLY_NUM_DOF + j0;
#pragma unroll
            for (int i = 0; i < 3; ++i) tp[i] = jt[i];
        }
    }
    if (PH & (PH_RESET | PH_INTEGRATE)) {
        if (is_leg) {
            float* dp = b.dof_state + e * (FLY_NUM_DOF * 2) + 2 * j0;
#pragma unroll
            for (int i = 0; i < 3; ++i) { dp[2 * i] = jq[i]; dp[2 * i + 1] = jqd[i]; }
        }
        if (sub == LANES_PER_ENV - 1) {
            float* rp = b.root + e * FLY_ROOT_DIM;
            rp[0] = r.px; rp[1] = r.py; rp[2] = r.pz; rp[3] = r.qx; rp[4] = r.qy; rp[5] = r.qz; rp[6] = r.qw;
            rp[7] = r.vx; rp[8] = r.vy; rp[9] = r.vz; rp[10] = r.wx; rp[11] = r.wy; rp[12] = r.wz;
        }
    }
    if (PH & PH_INTEGRATE) {
        if (is_leg) {
            float* fp = b.contact + (e * FLY_NUM_CONTACT + FLY_NUM_ABDOMEN + leg) * 3;
            fp[0] = cf[0]; fp[1] = cf[1]; fp[2] = cf[2];
        }
        if (is_abd) {
            float* fp = b.contact + (e * FLY_NUM_CONTACT + abd) * 3;
            fp[0] = cfa[0]; fp[1] = cfa[1]; fp[2] = cfa[2];
        }
    }
    if (sub == LANES_PER_ENV - 2) {
        if (PH & (PH_RESET | PH_OBS)) { b.pot[e] = pot; b.prev_pot[e] = prev_pot; }
        if (PH & (PH_RESET | PH_REWARD)) b.reset[e] = (int64_t)rs;
        if (PH & (PH_RESET | PH_REWARD | PH_PROGRESS)) b.progress[e] = (int64_t)progress;
        if (PH & PH_REWARD) {
            b.reward[e] = reward;
            if (b.ep_return) {        // episode statistics (optional)
                const float er = b.ep_return[e] + reward, el = b.ep_length[e] + 1.0f;
                if (rs) {
                    b.done_return[e] += er; b.done_length[e] += el; b.done_count[e] += 1.0f;
                    b.ep_return[e] = 0.0f; b.ep_length[e] = 0.0f;
                } else {
                    b.ep_return[e] = er; b.ep_length[e] = el;
                }
            }
        }
    }
}

inline int grid_for(int n) { return (n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK; }

}  // namespace

// Launchers used by flyhip_abi.hip
#define FLY_LAUNCH(PH)                                                                            \
    hipLaunchKernelGGL((fly_kernel<PH>), dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, \
                       dcfg, actions, *b)

extern "C" hipError_t flyhip_launch_env(int phases, const FlyConfig* dcfg, int n, const float* actions,
                                        const FlyBuffers* b, void* stream)
{
    switch (phases) {
    case PH_SCALE | PH_RESET | PH_INTEGRATE | PH_OBS | PH_PROGRESS | PH_REWARD:
        FLY_LAUNCH(PH_SCALE | PH_RESET | PH_INTEGRATE | PH_OBS | PH_PROGRESS | PH_REWARD); break;
    case PH_SCALE: FLY_LAUNCH(PH_SCALE); break;
    case PH_RESET: FLY_LAUNCH(PH_RESET); break;
    case PH_INTEGRATE: FLY_LAUNCH(PH_INTEGRATE); break;
    case PH_OBS: FLY_LAUNCH(PH_OBS); break;
    case PH_REWARD: FLY_LAUNCH(PH_REWARD); break;
    case PH_REWARD | PH_PROGRESS: FLY_LAUNCH(PH_REWARD | PH_PROGRESS); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
