/*
 * fly_oracle.h — TEST INFRASTRUCTURE.  Scalar CPU restatement of the reference's hot path
 * (petim0/fly_bProject fly.py / ppo.py) used only as the parity checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
 * (fly_bproject_amd/) never links, imports or calls anything in oracle/.
 *
 * Parity status
 *   - K1 scale, K2 reset, K4 obs pack, K5 reward/done pack, sampling/log-prob, TD+GAE, Net
 *     forward, PPO loss/update: PINNED by the .npz files under tests/golden, which were produced in the build
 *     container by executing the reference's own functions (tests/golden/gen_golden.py).
 *   - The third-party helper formulas underneath them (isaacgym.torch_utils,
 *     isaacgymenvs.utils.torch_jit_utils; unpinned, absent) are restated from their public
 *     definitions: PARITY UNPINNED at that boundary.
 *   - K3 physics: the reference calls closed-source PhysX (fly.py:482-485).  The rigid-body
 *     model here ("FlyDyn") is build-defined: PARITY UNPINNED vs the reference
 *     physics; this file is the specification the HIP kernel is checked against.
 *
 * Layouts are plain row-major AoS ([N][13], [N][18], [N][11][3], [N][73]) — deliberately NOT
 * the product's env-minor HBM layout, so that a layout bug cannot cancel out.
 */
#ifndef FLY_ORACLE_H
#define FLY_ORACLE_H
#include <stdint.h>

#define ORC_NDOF 18
#define ORC_NOBS 73
#define ORC_NLEG 6
#define ORC_NABD 5
#define ORC_NCON 11

typedef struct OrcConfig {
    int32_t num_envs;
    int32_t substeps;
    int32_t reset_after_sim;
    int32_t reward_mode;
    int32_t max_episode_length;
    float dt;
    float gravity;
    float kp, kd, effort, vmax;
    float joint_inertia;
    float mass;
    float inertia[3];
    float kc, cdamp, mu, cvisc;
    float lin_damp, ang_damp;
    float max_lin_vel, max_ang_vel;   /* per-component velocity clamps (Isaac Gym AssetOptions defaults 1000 / 64, fly.py:195) */
    float femur_len, tibia_len, alpha0, beta0;
    float dof_lo[ORC_NDOF];
    float dof_hi[ORC_NDOF];
    float dof_pose[ORC_NDOF];
    float leg_attach[ORC_NLEG][3];
    float leg_azimuth[ORC_NLEG];
    float leg_sigma[ORC_NLEG];
    float abdomen_pts[ORC_NABD][3];
    float start_height;
    float target[3];
    float dof_vel_scale;
    float up_weight;
    float heading_weight;
    float actions_cost_scale;
    float energy_cost_scale;
    float joints_at_limit_cost_scale;
    float death_cost;
    float termination_height;
    float termination_height_up;
} OrcConfig;

void orc_set_threads(int n);
/* fly.py:626-657 */
void orc_scale_actions(const OrcConfig* c, const float* actions, float* targets, int64_t n);
/* fly.py:446-480; returns number of envs reset */
int64_t orc_reset_masked(const OrcConfig* c, float* root, float* dof_pos, float* dof_vel,
                         float* pot, float* prev_pot, int64_t* reset, int64_t* progress, int64_t n);
/* build-defined stand-in for fly.py:482-485 */
void orc_physics_step(const OrcConfig* c, float* root, float* dof_pos, float* dof_vel,
                      const float* targets, float* contact, int64_t n);
void orc_physics_step_f64(const OrcConfig* c, double* root, double* dof_pos, double* dof_vel,
                          const double* targets, double* contact, int64_t n);
/* fly.py:397-411, :771-805 */
void orc_pack_obs(const OrcConfig* c, const float* root, const float* dof_pos, const float* dof_vel,
                  const float* targets, const float* contact, float* pot, float* prev_pot,
                  float* obs, float* up_vec, float* heading_vec, int64_t n);
/* fly.py:413-443, :685-768 */
void orc_pack_reward(const OrcConfig* c, const float* obs, const float* targets, const float* root,
                     const float* contact, const float* pot, const float* prev_pot,
                     int64_t* progress, float* reward, int64_t* reset, int64_t n);
/* fly.py:504-546: the viewer's P-key dump, terms [n][9] (heading, alive, up, orient, actions_cost, electricity_cost,
 * dof_at_limit_cost, progress_reward, leg_reward) */
void orc_reward_terms(const OrcConfig* c, const float* obs, const float* targets, const float* root,
                      const float* contact, const float* pot, const float* prev_pot, float* terms, int64_t n);
/* fly.py:624-681 */
void orc_env_step(const OrcConfig* c, const float* actions, float* root, float* dof_pos,
                  float* dof_vel, float* targets, float* contact, float* pot, float* prev_pot,
                  float* obs, float* reward, int64_t* reset, int64_t* progress, int64_t n);
/* ppo.py:213-220 */
void orc_sample_logprob(const float* mu, const float* var, const float* eps, float* act,
                        float* logp, int64_t n);
/* ppo.py:157-171 */
void orc_td_gae(const float* reward, const float* v, const float* v_next, const float* done,
                float gamma, float lambda, int64_t T, int64_t N, float* target, float* adv,
                int mode_flags);
/* ppo.py:10-102: out = head(shared(x)); head 0 = pi (18 outputs, ELU), 1 = v (1 output) */
void orc_net_forward(const float* const* w, const float* const* b, const float* x, int64_t n,
                     int head, float* out);
/* UselessFiles/dqn.py:89-100 and :64-85 */
void orc_dqn_eps_greedy(const float* q, const float* coin_u, const float* rand_u, float epsilon, int A,
                        float* act_out, int64_t n);
void orc_dqn_huber_td(const float* q_table, const float* act, const float* reward, const float* q_next,
                      const float* done, float discount, int A, int64_t B, float* dq, float* loss_out);
#endif
