// flyhip_abi.hip — the extern "C" surface declared in include/flyhip.h.
// Argument checking, handle lifetime and error strings live here; kernels live in fly_env.hip
// and ppo_kernels.hip.  Nothing here synchronises the host or allocates caller tensors.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <new>
#include "flyhip.h"

extern "C" hipError_t flyhip_launch_env(int phases, const FlyConfig* dcfg, int n, const float* actions,
                                        const FlyBuffers* b, void* stream);
extern "C" hipError_t flyhip_launch_sample_logprob(const float* mu, const float* var, const float* eps,
                                                   float* act_out, float* logp_out, int64_t n, void* stream);
extern "C" hipError_t flyhip_launch_td_gae(const float* reward, const float* v, const float* v_next,
                                           const float* done, float gamma, float lambda, int64_t T, int64_t N,
                                           float* target_out, float* adv_out, int mode, void* stream);

extern "C" hipError_t flyhip_launch_mlp_forward(const float* P, const float* PF, const float* x, int64_t n, float* mu_out,
                                                float* v_out, float* out_save, float* h1_save, float* h2_save,
                                                float* h3_save, const uint16_t* PB, void* stream);

extern "C" hipError_t flyhip_launch_mlp_backward_dx(const float* PT, const float* out_saved, const float* h1,
                                                    const float* h2, const float* h3, const float* action,
                                                    const float* old_logp, const float* adv, const float* target,
                                                    const float* var, int64_t n, float inv_batch, float clip,
                                                    float* dz4, float* dz3, float* dz2, float* dz1, float* loss_part,
                                                    const uint16_t* PTB, void* stream);
extern "C" hipError_t flyhip_launch_dqn_forward(const float* P, const float* PF, const float* x, int64_t n, float* q_out,
                                                void* stream);
extern "C" hipError_t flyhip_launch_dqn_act(const float* P, const float* PF, const float* x, int64_t n, const float* coin_u,
                                            const float* rand_u, float epsilon, float* act_out, float* q_out, void* stream);
extern "C" hipError_t flyhip_launch_dqn_td(const float* P, const float* PF, const float* PT, const float* P_tgt,
                                           const float* PF_tgt, const float* obs, const float* next_obs, const float* act,
                                           const float* reward, const float* done, int64_t n, float discount, float inv_B,
                                           float* h1, float* h2, float* dz3, float* dz2, float* dz1, float* loss_part,
                                           void* stream);
extern "C" int64_t flyhip_dqn_grad_workspace_floats(void);
extern "C" hipError_t flyhip_launch_dqn_grad_w(const float* x, const float* h1, const float* h2, const float* dz1,
                                               const float* dz2, const float* dz3, int64_t n, float* workspace, float* grad,
                                               int accumulate, void* stream);
extern "C" hipError_t flyhip_launch_dqn_adam(float* P, float* PF, float* PT, float* P_tgt, float* PF_tgt, const int* idx_f,
                                             const int* idx_t, const float* G, const float* mask, float* m, float* v,
                                             int* step, float lr, float beta1, float beta2, float eps, float tau, uint16_t* QB, uint16_t* QTB, uint16_t* QB_tgt, const int* idx_fb, const int* idx_tb,
                                             const int* grad_invalid, void* stream);
extern "C" hipError_t flyhip_p2p_alloc(int64_t n_floats, void** out);
extern "C" hipError_t flyhip_launch_p2p_allreduce(float* G, int64_t n, void* const* bases, int rank, int world,
                                                  uint32_t epoch, int* err, int64_t fail_slot, void* stream);
extern "C" int64_t flyhip_mlp_grad_workspace_floats(void);
extern "C" int64_t flyhip_mlp_fused_workspace_floats(void);
extern "C" hipError_t flyhip_launch_mlp_fused_grad(const float* P, const uint16_t* PB, const uint16_t* PTB, const float* x,
                                                   int64_t n, const float* action, const float* old_logp, const float* adv,
                                                   const float* target, const float* var, float inv_batch, float clip,
                                                   float* workspace, float* grad_out, const float* norm_mask, float* norm_ws,
                                                   int* norm_step, float* loss_part, float* const* dump, void* stream);
extern "C" hipError_t flyhip_launch_mlp_grad_w(const float* x, const float* h1, const float* h2, const float* h3,
                                               const float* dz1, const float* dz2, const float* dz3, const float* dz4,
                                               int64_t n, float* workspace, float* grad_out, const float* norm_mask,
                                               float* norm_ws, int* norm_step, const int* err, int gemm_b3, void* stream);
extern "C" hipError_t flyhip_launch_mlp_adam(float* P, float* PF, float* PT, const int* idx_f, const int* idx_t,
                                             const float* G, const float* mask, float* m,
                                             float* v, int* step, float lr, float beta1, float beta2, float eps,
                                             float max_norm, float grad_scale, float* norm_ws, int norm_ready,
                                             uint16_t* PB, uint16_t* PTB, const int* idx_fb, const int* idx_tb,
                                             int* step_out, const int* grad_invalid, uint16_t* PH, uint16_t* PTH,
                                             float* h2_scales, int h2_rescale, void* stream);
extern "C" hipError_t flyhip_launch_mlp_h2_rescale(const float* P, const int* idx_fb, const int* idx_tb, uint16_t* PH, uint16_t* PTH,
                                                   float* h2_scales, void* stream);
extern "C" int64_t flyhip_mlp_fused_h2_workspace_floats(void);
extern "C" hipError_t flyhip_launch_mlp_fused_grad_h2(const float* P, const uint16_t* PH, const uint16_t* PTH, float* fsc, int* ovf,
                                                      int freeze, const float* x, int64_t n, const float* action,
                                                      const float* old_logp, const float* adv, const float* target, const float* var,
                                                      float inv_batch, float clip, float* workspace, float* grad_out,
                                                      const float* norm_mask, float* norm_ws, int* norm_step, float* loss_part,
                                                      float* const* dump, void* stream);

extern "C" hipError_t flyhip_launch_bookkeeping(const float* reward, int64_t n, float* score_acc, float score_scale,
                                                float* action_var, int nvar, float var_decay, float var_min,
                                                void* stream);

extern "C" hipError_t flyhip_launch_dqn_eps_greedy(const float* q, const float* coin_u, const float* rand_u, float epsilon,
                                                   int A, float* act_out, int64_t n, void* stream);
extern "C" hipError_t flyhip_launch_dqn_huber_td(const float* q_table, const float* act, const float* reward,
                                                 const float* q_next, const float* done, float discount, int A, int64_t B,
                                                 float* dq, float* loss_part, void* stream);

extern "C" hipError_t flyhip_launch_mlp_forward_sample(const float* P, const float* PF, const float* x, int64_t n,
                                                       const float* eps, const float* var, int var_steps,
                                                       float var_decay, float var_min, float* act_out,
                                                       float* logp_out, float* mu_out, float* v_out, const uint16_t* PB,
                                                       const int* var_base, void* stream);

extern "C" hipError_t flyhip_launch_mlp_fwd_bwd(const float* P, const float* PF, const float* PT, const float* x, int64_t n,
                                                float* out_save, float* h1_save, float* h2_save, float* h3_save,
                                                const float* action, const float* old_logp, const float* adv,
                                                const float* target, const float* var, float inv_batch, float clip,
                                                float* dz4, float* dz3, float* dz2, float* dz1, float* loss_part,
                                                int* flags, int epoch, int* err, const uint16_t* PB, const uint16_t* PTB,
                                                int coherent, void* stream);
extern "C" hipError_t flyhip_launch_rollout_all(const FlyConfig* dcfg, const FlyBuffers* b, const float* P, const float* PF,
                                                float* obs_ring, int64_t n, const float* eps_all, const float* var,
                                                float var_decay, float var_min, float* act_all, float* logp_all, float* v_ring,
                                                float* reward_all, int T, const int* rows_applied, const uint16_t* PB,
                                                int64_t* reset_rows, int64_t* progress_rows, void* stream,
                                                unsigned long long* stamps);
extern "C" hipError_t flyhip_launch_rollout_bookkeeping(const float* reward, int64_t rows, int64_t n, float* terms,
                                                        float* score_acc, float score_scale, float* action_var, int nvar,
                                                        float var_decay, float var_min, int* rows_applied, void* stream);
extern "C" hipError_t flyhip_launch_rollout_step(const FlyConfig* dcfg, const FlyBuffers* b, const float* P, const float* PF,
                                                 const float* x, int64_t n, const float* eps, const float* var, int var_steps,
                                                 float var_decay, float var_min, float* act, float* logp, float* v_out,
                                                 const uint16_t* PB, const int* var_base, void* stream);
extern "C" hipError_t flyhip_launch_adv_stats(const float* adv, int64_t n, float* stats, void* stream);
extern "C" hipError_t flyhip_launch_adv_apply(float* adv, int64_t n, const float* totals, float count, float eps, void* stream);

struct FlyEnv {
    FlyConfig host;
    FlyConfig* dev;
};

namespace {
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what)
{
    return fail(FLY_E_HIP, "%s: %s", what, hipGetErrorString(e));
}

enum : int { PH_SCALE = 1, PH_RESET = 2, PH_INTEGRATE = 4, PH_OBS = 8, PH_REWARD = 16, PH_PROGRESS = 32 };

int check_buffers(const FlyBuffers* b, int phases)
{
    if (!b) return fail(FLY_E_ARG, "FlyBuffers is null");
    const bool need_root = phases & (PH_RESET | PH_INTEGRATE | PH_OBS | PH_REWARD);
    const bool need_dof = phases & (PH_RESET | PH_INTEGRATE | PH_OBS);
    const bool need_tgt = phases & (PH_SCALE | PH_INTEGRATE | PH_OBS | PH_REWARD);
    const bool need_con = phases & (PH_INTEGRATE | PH_OBS | PH_REWARD);
    const bool need_pot = phases & (PH_RESET | PH_OBS | PH_REWARD);
    const bool need_obs = phases & (PH_OBS | PH_REWARD);
    const bool need_flags = phases & (PH_RESET | PH_REWARD | PH_PROGRESS);
    if (need_root && !b->root) return fail(FLY_E_ARG, "root is null");
    if (need_dof && !b->dof_state) return fail(FLY_E_ARG, "dof_state is null");
    if (need_tgt && !b->targets) return fail(FLY_E_ARG, "targets is null");
    if (need_con && !b->contact) return fail(FLY_E_ARG, "contact is null");
    if (need_pot && (!b->pot || !b->prev_pot)) return fail(FLY_E_ARG, "pot/prev_pot is null");
    if (need_obs && !b->obs) return fail(FLY_E_ARG, "obs is null");
    if (need_obs && ((uintptr_t)b->obs & 3)) return fail(FLY_E_ARG, "obs must be 4-byte aligned");
    if ((phases & PH_REWARD) && !b->reward) return fail(FLY_E_ARG, "reward is null");
    if (need_flags && (!b->reset || !b->progress)) return fail(FLY_E_ARG, "reset/progress is null");
    return FLY_OK;
}

int launch(FlyHandle h, int phases, const float* actions, const FlyBuffers* b, void* stream)
{
    if (!h) return fail(FLY_E_ARG, "handle is null");
    if ((phases & PH_SCALE) && !actions) return fail(FLY_E_ARG, "actions is null");
    int rc = check_buffers(b, phases);
    if (rc) return rc;
    hipError_t e = flyhip_launch_env(phases, h->dev, h->host.num_envs, actions, b, stream);
    if (e != hipSuccess) return hip_fail(e, "fly env kernel launch");
    return FLY_OK;
}
}  // namespace

extern "C" {

const char* fly_last_error(void) { return g_err; }
int fly_abi_version(void) { return 12; }

int fly_create(const FlyConfig* cfg, FlyHandle* out)
{
    if (!cfg || !out) return fail(FLY_E_ARG, "fly_create: null argument");
    if (cfg->num_envs <= 0) return fail(FLY_E_CONFIG, "num_envs must be > 0 (got %d)", cfg->num_envs);
    if (cfg->substeps <= 0 || cfg->substeps > 1024) return fail(FLY_E_CONFIG, "substeps out of range (%d)", cfg->substeps);
    if (!(cfg->dt > 0.0f)) return fail(FLY_E_CONFIG, "dt must be > 0");
    if (!(cfg->mass > 0.0f) || !(cfg->joint_inertia > 0.0f)) return fail(FLY_E_CONFIG, "mass / joint_inertia must be > 0");
    for (int i = 0; i < 3; ++i)
        if (!(cfg->inertia[i] > 0.0f)) return fail(FLY_E_CONFIG, "inertia[%d] must be > 0", i);
    for (int j = 0; j < FLY_NUM_DOF; ++j)
        if (!(cfg->dof_hi[j] > cfg->dof_lo[j])) return fail(FLY_E_CONFIG, "dof limits of joint %d are not ordered", j);
    if (cfg->reward_mode != 0 && cfg->reward_mode != 1) return fail(FLY_E_CONFIG, "reward_mode must be 0 or 1");
    FlyEnv* h = new (std::nothrow) FlyEnv;
    if (!h) return fail(FLY_E_ARG, "out of host memory");
    h->host = *cfg;
    h->dev = nullptr;
    hipError_t e = hipMalloc((void**)&h->dev, sizeof(FlyConfig));
    if (e != hipSuccess) { delete h; return hip_fail(e, "hipMalloc(FlyConfig)"); }
    e = hipMemcpy(h->dev, cfg, sizeof(FlyConfig), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(h->dev); delete h; return hip_fail(e, "hipMemcpy(FlyConfig)"); }
    *out = h;
    return FLY_OK;
}

int fly_destroy(FlyHandle h)
{
    if (!h) return fail(FLY_E_ARG, "handle is null");
    hipError_t e = hipFree(h->dev);
    delete h;
    if (e != hipSuccess) return hip_fail(e, "hipFree(FlyConfig)");
    return FLY_OK;
}

int fly_step(FlyHandle h, const float* actions, const FlyBuffers* b, void* stream)
{
    return launch(h, PH_SCALE | PH_RESET | PH_INTEGRATE | PH_OBS | PH_PROGRESS | PH_REWARD, actions, b, stream);
}

int ppo_rollout_step(FlyHandle h, const FlyBuffers* b, const float* params, const float* params_frag, const float* x,
                     const float* eps, const float* var, int32_t var_steps, float var_decay, float var_min,
                     float* act_out, float* logp_out, float* v_out, const uint16_t* params_b3,
                     const int32_t* var_steps_base, void* stream)
{
    if (!h) return fail(FLY_E_ARG, "handle is null");
    if (!params || !params_frag || !x || !eps || !var || !act_out || !logp_out)
        return fail(FLY_E_ARG, "ppo_rollout_step: null pointer");
    if (var_steps < 0 || var_steps > (1 << 20)) return fail(FLY_E_ARG, "ppo_rollout_step: var_steps out of range");
    int rc = check_buffers(b, PH_SCALE | PH_RESET | PH_INTEGRATE | PH_OBS | PH_PROGRESS | PH_REWARD);
    if (rc) return rc;
    hipError_t e = flyhip_launch_rollout_step(h->dev, b, params, params_frag, x, h->host.num_envs, eps, var, var_steps,
                                              var_decay, var_min, act_out, logp_out, v_out, params_b3, var_steps_base, stream);
    if (e != hipSuccess) return hip_fail(e, "ppo_rollout_step launch");
    return FLY_OK;
}

static int rollout_all_impl(FlyHandle h, const FlyBuffers* b, const float* params, const float* params_frag, float* obs_ring,
                            const float* eps_all, const float* var, float var_decay, float var_min, float* act_all,
                            float* logp_all, float* v_ring, float* reward_all, int32_t T, const int32_t* rows_applied,
                            const uint16_t* params_b3, int64_t* reset_rows, int64_t* progress_rows, void* stream,
                            unsigned long long* stamps)
{
    if ((reset_rows != nullptr) != (progress_rows != nullptr)) return fail(FLY_E_ARG, "ppo_rollout_all: reset_rows and progress_rows go together");
    if (!h) return fail(FLY_E_ARG, "handle is null");
    if (!params || !params_frag || !obs_ring || !eps_all || !var || !act_all || !logp_all || !v_ring || !reward_all)
        return fail(FLY_E_ARG, "ppo_rollout_all: null pointer");
    if (T <= 0 || T > (1 << 20)) return fail(FLY_E_ARG, "ppo_rollout_all: T out of range");
    if (!b) return fail(FLY_E_ARG, "ppo_rollout_all: buffers are null");
    FlyBuffers bb = *b;
    bb.obs = obs_ring; bb.reward = reward_all;              // checked as present; the kernel walks the rows itself
    int rc = check_buffers(&bb, PH_SCALE | PH_RESET | PH_INTEGRATE | PH_OBS | PH_PROGRESS | PH_REWARD);
    if (rc) return rc;
    hipError_t e = flyhip_launch_rollout_all(h->dev, &bb, params, params_frag, obs_ring, h->host.num_envs, eps_all, var,
                                             var_decay, var_min, act_all, logp_all, v_ring, reward_all, T, rows_applied,
                                             params_b3, reset_rows, progress_rows, stream, stamps);
    if (e != hipSuccess) return hip_fail(e, "ppo_rollout_all launch");
    return FLY_OK;
}

int ppo_rollout_all(FlyHandle h, const FlyBuffers* b, const float* params, const float* params_frag, float* obs_ring,
                    const float* eps_all, const float* var, float var_decay, float var_min, float* act_all,
                    float* logp_all, float* v_ring, float* reward_all, int32_t T, const int32_t* rows_applied,
                    const uint16_t* params_b3, int64_t* reset_rows, int64_t* progress_rows, void* stream)
{
    return rollout_all_impl(h, b, params, params_frag, obs_ring, eps_all, var, var_decay, var_min, act_all, logp_all, v_ring,
                            reward_all, T, rows_applied, params_b3, reset_rows, progress_rows, stream, nullptr);
}

// diagnostic: the same launch through the stamped instantiation of rollout_all_fs_kernel (bench.py's policy / env split of a
// rollout step, tools/ab_rollout.py); stamps u64 [tiles][T + 1][8].  Not part of the ABI header.
extern "C" int flyhip_debug_rollout_all_stamped(FlyHandle h, const FlyBuffers* b, const float* params, const float* params_frag,
                                                float* obs_ring, const float* eps_all, const float* var, float var_decay,
                                                float var_min, float* act_all, float* logp_all, float* v_ring, float* reward_all,
                                                int32_t T, const uint16_t* params_b3, int64_t* reset_rows, int64_t* progress_rows,
                                                unsigned long long* stamps, void* stream)
{
    if (!stamps) return fail(FLY_E_ARG, "stamps is null");
    return rollout_all_impl(h, b, params, params_frag, obs_ring, eps_all, var, var_decay, var_min, act_all, logp_all, v_ring,
                            reward_all, T, nullptr, params_b3, reset_rows, progress_rows, stream, stamps);
}

int fly_scale_actions(FlyHandle h, const float* actions, float* targets, void* stream)
{
    FlyBuffers b = {};
    b.targets = targets;
    return launch(h, PH_SCALE, actions, &b, stream);
}

int fly_reset_masked(FlyHandle h, const FlyBuffers* b, void* stream) { return launch(h, PH_RESET, nullptr, b, stream); }
int fly_integrate(FlyHandle h, const FlyBuffers* b, void* stream) { return launch(h, PH_INTEGRATE, nullptr, b, stream); }
int fly_pack_obs(FlyHandle h, const FlyBuffers* b, void* stream) { return launch(h, PH_OBS, nullptr, b, stream); }
int fly_pack_reward(FlyHandle h, const FlyBuffers* b, int add_progress, void* stream)
{
    return launch(h, add_progress ? (PH_REWARD | PH_PROGRESS) : PH_REWARD, nullptr, b, stream);
}

int ppo_sample_logprob(const float* mu, const float* var, const float* eps, float* act_out,
                       float* logp_out, int64_t n, void* stream)
{
    if (!mu || !var || !eps || !act_out || !logp_out) return fail(FLY_E_ARG, "ppo_sample_logprob: null pointer");
    if (n <= 0) return fail(FLY_E_ARG, "ppo_sample_logprob: n must be > 0");
    hipError_t e = flyhip_launch_sample_logprob(mu, var, eps, act_out, logp_out, n, stream);
    if (e != hipSuccess) return hip_fail(e, "ppo_sample_logprob launch");
    return FLY_OK;
}

int ppo_td_gae(const float* reward, const float* v, const float* v_next, const float* done,
               float gamma, float lambda, int64_t T, int64_t N, float* target_out, float* adv_out,
               int mode_flags, void* stream)
{
    if (!reward || !v || !v_next || !done || !target_out || !adv_out) return fail(FLY_E_ARG, "ppo_td_gae: null pointer");
    if (T <= 0 || N <= 0) return fail(FLY_E_ARG, "ppo_td_gae: T and N must be > 0");
    if ((mode_flags & PPO_GAE_SCAN) && (mode_flags & PPO_GAE_MASK_RECURRENCE))
        return fail(FLY_E_ARG, "ppo_td_gae: PPO_GAE_SCAN does not implement the masked recurrence");
    hipError_t e = flyhip_launch_td_gae(reward, v, v_next, done, gamma, lambda, T, N, target_out, adv_out, mode_flags, stream);
    if (e != hipSuccess) return hip_fail(e, "ppo_td_gae launch");
    return FLY_OK;
}

int ppo_adv_stats(const float* adv, int64_t n, float* stats, void* stream)
{
    if (!adv || !stats || n <= 1) return fail(FLY_E_ARG, "ppo_adv_stats: bad argument");
    hipError_t e = flyhip_launch_adv_stats(adv, n, stats, stream);
    if (e != hipSuccess) return hip_fail(e, "ppo_adv_stats launch");
    return FLY_OK;
}

int ppo_adv_apply(float* adv, int64_t n, const float* totals, float count, float eps, void* stream)
{
    if (!adv || !totals || n <= 0 || !(count > 1.0f)) return fail(FLY_E_ARG, "ppo_adv_apply: bad argument");
    hipError_t e = flyhip_launch_adv_apply(adv, n, totals, count, eps, stream);
    if (e != hipSuccess) return hip_fail(e, "ppo_adv_apply launch");
    return FLY_OK;
}

int ppo_rollout_bookkeeping(const float* reward, int64_t rows, int64_t n, float* terms, float* score_acc,
                            float score_scale, float* action_var, int32_t nvar, float var_decay,
                            float var_min, int32_t* rows_applied, void* stream)
{
    if (!reward || !terms || !score_acc || !action_var) return fail(FLY_E_ARG, "ppo_rollout_bookkeeping: null pointer");
    if (rows < 0 || n <= 0 || nvar < 0 || nvar > 63) return fail(FLY_E_ARG, "ppo_rollout_bookkeeping: bad size");
    if (rows == 0) return FLY_OK;
    hipError_t e = flyhip_launch_rollout_bookkeeping(reward, rows, n, terms, score_acc, score_scale, action_var, nvar,
                                                     var_decay, var_min, rows_applied, stream);
    if (e != hipSuccess) return hip_fail(e, "ppo_rollout_bookkeeping launch");
    return FLY_OK;
}

int ppo_step_bookkeeping(const float* reward, int64_t n, float* score_acc, float score_scale,
                         float* action_var, int32_t nvar, float var_decay, float var_min, void* stream)
{
    if (!reward || !score_acc || !action_var) return fail(FLY_E_ARG, "ppo_step_bookkeeping: null pointer");
    if (n <= 0 || nvar < 0 || nvar > 256) return fail(FLY_E_ARG, "ppo_step_bookkeeping: bad size");
    hipError_t e = flyhip_launch_bookkeeping(reward, n, score_acc, score_scale, action_var, nvar, var_decay, var_min, stream);
    if (e != hipSuccess) return hip_fail(e, "ppo_step_bookkeeping launch");
    return FLY_OK;
}

int mlp_forward(const float* params, const float* params_frag, const float* x, int64_t n, float* mu_out, float* v_out,
                float* out_save, float* h1_save, float* h2_save, float* h3_save, const uint16_t* params_b3, void* stream)
{
    if (((uintptr_t)params_b3 & 15)) return fail(FLY_E_ARG, "mlp_forward: params_b3 must be 16-byte aligned");
    if (!params || !params_frag || !x) return fail(FLY_E_ARG, "mlp_forward: null params/x");
    if (n <= 0) return fail(FLY_E_ARG, "mlp_forward: n must be > 0");
    if (((uintptr_t)params_frag & 15)) return fail(FLY_E_ARG, "mlp_forward: params_frag must be 16-byte aligned");
    hipError_t e = flyhip_launch_mlp_forward(params, params_frag, x, n, mu_out, v_out, out_save, h1_save, h2_save, h3_save,
                                             params_b3, stream);
    if (e != hipSuccess) return hip_fail(e, "mlp_forward launch");
    return FLY_OK;
}

int mlp_forward_sample(const float* params, const float* params_frag, const float* x, int64_t n,
                       const float* eps, const float* var, int32_t var_steps, float var_decay,
                       float var_min, float* act_out, float* logp_out, float* mu_out, float* v_out,
                       const uint16_t* params_b3, const int32_t* var_steps_base, void* stream)
{
    if (!params || !params_frag || !x || !eps || !var || !act_out || !logp_out)
        return fail(FLY_E_ARG, "mlp_forward_sample: null pointer");
    if (n <= 0) return fail(FLY_E_ARG, "mlp_forward_sample: n must be > 0");
    if (var_steps < 0 || var_steps > (1 << 20)) return fail(FLY_E_ARG, "mlp_forward_sample: var_steps out of range");
    hipError_t e = flyhip_launch_mlp_forward_sample(params, params_frag, x, n, eps, var, var_steps, var_decay, var_min, act_out, logp_out,
                                                    mu_out, v_out, params_b3, var_steps_base, stream);
    if (e != hipSuccess) return hip_fail(e, "mlp_forward_sample launch");
    return FLY_OK;
}

int64_t mlp_grad_workspace_floats(void) { return flyhip_mlp_grad_workspace_floats(); }

int mlp_backward_dx(const float* params_t, const float* out_saved, const float* h1_saved,
                    const float* h2_saved, const float* h3_saved, const float* action,
                    const float* old_logp, const float* adv, const float* target, const float* var,
                    int64_t n, float inv_batch, float clip, float* dz4, float* dz3, float* dz2,
                    float* dz1, float* loss_part, const uint16_t* params_t_b3, void* stream)
{
    if (!params_t || !out_saved || !h1_saved || !h2_saved || !h3_saved || !action || !old_logp || !adv ||
        !target || !var || !dz4 || !dz3 || !dz2 || !dz1)
        return fail(FLY_E_ARG, "mlp_backward_dx: null pointer");
    if (n <= 0) return fail(FLY_E_ARG, "mlp_backward_dx: n must be > 0");
    hipError_t e = flyhip_launch_mlp_backward_dx(params_t, out_saved, h1_saved, h2_saved, h3_saved, action, old_logp,
                                                 adv, target, var, n, inv_batch, clip, dz4, dz3, dz2, dz1, loss_part,
                                                 params_t_b3, stream);
    if (e != hipSuccess) return hip_fail(e, "mlp_backward_dx launch");
    return FLY_OK;
}

int mlp_forward_backward(const float* params, const float* params_frag, const float* params_t_frag,
                         const float* x, int64_t n, float* out_save, float* h1_save, float* h2_save,
                         float* h3_save, const float* action, const float* old_logp, const float* adv,
                         const float* target, const float* var, float inv_batch, float clip,
                         float* dz4, float* dz3, float* dz2, float* dz1, float* loss_part,
                         int32_t* flags, int32_t epoch, int32_t* err, const uint16_t* params_b3,
                         const uint16_t* params_t_b3, int32_t coherent, void* stream)
{
    if ((params_b3 == nullptr) != (params_t_b3 == nullptr))
        return fail(FLY_E_ARG, "mlp_forward_backward: params_b3 and params_t_b3 go together");
    if (!params || !params_frag || !params_t_frag || !x || !out_save || !h1_save || !h2_save || !h3_save || !action ||
        !old_logp || !adv || !target || !var || !dz4 || !dz3 || !dz2 || !dz1 || !flags || !err)
        return fail(FLY_E_ARG, "mlp_forward_backward: null pointer");
    if (n <= 0) return fail(FLY_E_ARG, "mlp_forward_backward: n must be > 0");
    if (epoch <= 0 || epoch >= (1 << 27)) return fail(FLY_E_ARG, "mlp_forward_backward: epoch must be in [1, 2^27)");
    hipError_t e = flyhip_launch_mlp_fwd_bwd(params, params_frag, params_t_frag, x, n, out_save, h1_save, h2_save, h3_save,
                                             action, old_logp, adv, target, var, inv_batch, clip, dz4, dz3, dz2, dz1,
                                             loss_part, flags, epoch, err, params_b3, params_t_b3, coherent, stream);
    if (e != hipSuccess) return hip_fail(e, "mlp_forward_backward launch");
    return FLY_OK;
}

int mlp_grad_w(const float* x, const float* h1_saved, const float* h2_saved, const float* h3_saved,
               const float* dz1, const float* dz2, const float* dz3, const float* dz4, int64_t n,
               float* workspace, float* grad, const float* norm_mask, float* norm_ws, int32_t* norm_step,
               const int32_t* err, int32_t gemm_b3, void* stream)
{
    if ((norm_ws != nullptr) != (norm_mask != nullptr) || (norm_ws != nullptr) != (norm_step != nullptr))
        return fail(FLY_E_ARG, "mlp_grad_w: norm_mask, norm_ws and norm_step go together");
    if (!x || !h1_saved || !h2_saved || !h3_saved || !dz1 || !dz2 || !dz3 || !dz4 || !workspace || !grad)
        return fail(FLY_E_ARG, "mlp_grad_w: null pointer");
    if (n <= 0) return fail(FLY_E_ARG, "mlp_grad_w: n must be > 0");
    hipError_t e = flyhip_launch_mlp_grad_w(x, h1_saved, h2_saved, h3_saved, dz1, dz2, dz3, dz4, n, workspace, grad, norm_mask,
                                            norm_ws, norm_step, err, gemm_b3, stream);
    if (e != hipSuccess) return hip_fail(e, "mlp_grad_w launch");
    return FLY_OK;
}

int64_t mlp_fused_workspace_floats(void) { return flyhip_mlp_fused_workspace_floats(); }

int mlp_fused_grad(const float* params, const uint16_t* params_b3, const uint16_t* params_t_b3, const float* x, int64_t n,
                   const float* action, const float* old_logp, const float* adv, const float* target, const float* var,
                   float inv_batch, float clip, float* workspace, float* grad, const float* norm_mask, float* norm_ws,
                   int32_t* norm_step, float* loss_part, float* const* debug_dump, void* stream)
{
    if ((norm_ws != nullptr) != (norm_mask != nullptr) || (norm_ws != nullptr) != (norm_step != nullptr))
        return fail(FLY_E_ARG, "mlp_fused_grad: norm_mask, norm_ws and norm_step go together");
    if (!params || !params_b3 || !params_t_b3) return fail(FLY_E_ARG, "mlp_fused_grad: needs params and both bf16x3 plane buffers");
    if (!x || !action || !old_logp || !adv || !target || !var || !workspace || !grad)
        return fail(FLY_E_ARG, "mlp_fused_grad: null pointer");
    if (n <= 0) return fail(FLY_E_ARG, "mlp_fused_grad: n must be > 0");
    if (debug_dump) {
        if (!debug_dump[0]) return fail(FLY_E_ARG, "mlp_fused_grad: debug_dump[0] is null");
        if (debug_dump[1])      // the chain dump: all eight; {stamps, NULL}: the diagnostic stamp build (tools/stamp_fused.py)
            for (int i = 2; i < 8; ++i)
                if (!debug_dump[i]) return fail(FLY_E_ARG, "mlp_fused_grad: debug_dump[%d] is null", i);
    }
    hipError_t e = flyhip_launch_mlp_fused_grad(params, params_b3, params_t_b3, x, n, action, old_logp, adv, target, var, inv_batch,
                                                clip, workspace, grad, norm_mask, norm_ws, norm_step, loss_part, debug_dump, stream);
    if (e != hipSuccess) return hip_fail(e, "mlp_fused_grad launch");
    return FLY_OK;
}

int64_t mlp_fused_h2_workspace_floats(void) { return flyhip_mlp_fused_h2_workspace_floats(); }

int mlp_h2_rescale(const float* params, const int32_t* idx_b3, const int32_t* idx_t_b3, uint16_t* params_h2, uint16_t* params_t_h2,
                   float* h2_scales, void* stream)
{
    if (!params || !idx_b3 || !idx_t_b3 || !params_h2 || !params_t_h2 || !h2_scales) return fail(FLY_E_ARG, "mlp_h2_rescale: null pointer");
    hipError_t e = flyhip_launch_mlp_h2_rescale(params, idx_b3, idx_t_b3, params_h2, params_t_h2, h2_scales, stream);
    if (e != hipSuccess) return hip_fail(e, "mlp_h2_rescale launch");
    return FLY_OK;
}

int mlp_fused_grad_h2(const float* params, const uint16_t* params_h2, const uint16_t* params_t_h2, float* h2_scales,
                      int32_t* h2_overflow, int32_t freeze, const float* x, int64_t n, const float* action,
                      const float* old_logp, const float* adv, const float* target, const float* var, float inv_batch,
                      float clip, float* workspace, float* grad, const float* norm_mask, float* norm_ws, int32_t* norm_step,
                      float* loss_part, float* const* debug_dump, void* stream)
{
    if ((norm_ws != nullptr) != (norm_mask != nullptr) || (norm_ws != nullptr) != (norm_step != nullptr))
        return fail(FLY_E_ARG, "mlp_fused_grad_h2: norm_mask, norm_ws and norm_step go together");
    if (!params || !params_h2 || !params_t_h2 || !h2_scales || !h2_overflow)
        return fail(FLY_E_ARG, "mlp_fused_grad_h2: needs params, both fp16x2 plane buffers, the scale table and the overflow word");
    if (!x || !action || !old_logp || !adv || !target || !var || !workspace || !grad)
        return fail(FLY_E_ARG, "mlp_fused_grad_h2: null pointer");
    if (n <= 0) return fail(FLY_E_ARG, "mlp_fused_grad_h2: n must be > 0");
    if (debug_dump) {
        if (!debug_dump[0]) return fail(FLY_E_ARG, "mlp_fused_grad_h2: debug_dump[0] is null");
        if (debug_dump[1])
            for (int i = 2; i < 8; ++i)
                if (!debug_dump[i]) return fail(FLY_E_ARG, "mlp_fused_grad_h2: debug_dump[%d] is null", i);
    }
    hipError_t e = flyhip_launch_mlp_fused_grad_h2(params, params_h2, params_t_h2, h2_scales, h2_overflow, freeze, x, n, action, old_logp,
                                                   adv, target, var, inv_batch, clip, workspace, grad, norm_mask, norm_ws, norm_step,
                                                   loss_part, debug_dump, stream);
    if (e != hipSuccess) return hip_fail(e, "mlp_fused_grad_h2 launch");
    return FLY_OK;
}

int mlp_adam_step(float* params, float* params_frag, float* params_t_frag, const int32_t* idx_frag,
                  const int32_t* idx_t_frag, const float* grad, const float* mask, float* exp_avg,
                  float* exp_avg_sq, int32_t* step, float lr, float beta1, float beta2, float eps,
                  float max_norm, float grad_scale, float* norm_ws, int32_t norm_ready, uint16_t* params_b3,
                  uint16_t* params_t_b3, const int32_t* idx_b3, const int32_t* idx_t_b3, int32_t* step_out,
                  const int32_t* grad_invalid, uint16_t* params_h2, uint16_t* params_t_h2, float* h2_scales, int32_t h2_rescale,
                  void* stream)
{
    if (params_b3 && (!params_t_b3 || !idx_b3 || !idx_t_b3))
        return fail(FLY_E_ARG, "mlp_adam_step: params_b3 needs params_t_b3, idx_b3 and idx_t_b3");
    if (params_h2 && (!params_t_h2 || !h2_scales || !params_b3))
        return fail(FLY_E_ARG, "mlp_adam_step: params_h2 needs params_t_h2, h2_scales and the bf16x3 planes with their index maps");
    if (!params || !params_frag || !params_t_frag || !idx_frag || !idx_t_frag || !grad || !mask || !exp_avg ||
        !exp_avg_sq || !step || !norm_ws)
        return fail(FLY_E_ARG, "mlp_adam_step: null pointer");
    hipError_t e = flyhip_launch_mlp_adam(params, params_frag, params_t_frag, idx_frag, idx_t_frag, grad, mask, exp_avg, exp_avg_sq, step, lr, beta1, beta2,
                                          eps, max_norm, grad_scale, norm_ws, norm_ready, params_b3, params_t_b3, idx_b3, idx_t_b3,
                                          step_out, grad_invalid, params_h2, params_t_h2, h2_scales, h2_rescale, stream);
    if (e != hipSuccess) return hip_fail(e, "mlp_adam_step launch");
    return FLY_OK;
}

int dqn_eps_greedy(const float* q, const float* coin_u, const float* rand_u, float epsilon, int32_t A,
                   float* act_out, int64_t n, void* stream)
{
    if (!q || !coin_u || !rand_u || !act_out) return fail(FLY_E_ARG, "dqn_eps_greedy: null pointer");
    if (n <= 0 || A < 2) return fail(FLY_E_ARG, "dqn_eps_greedy: need n > 0 and A >= 2");
    hipError_t e = flyhip_launch_dqn_eps_greedy(q, coin_u, rand_u, epsilon, A, act_out, n, stream);
    if (e != hipSuccess) return hip_fail(e, "dqn_eps_greedy launch");
    return FLY_OK;
}

int dqn_huber_td(const float* q_table, const float* act, const float* reward, const float* q_next,
                 const float* done, float discount, int32_t A, int64_t B, float* dq, float* loss_part,
                 void* stream)
{
    if (!q_table || !act || !reward || !q_next || !done || !dq || !loss_part)
        return fail(FLY_E_ARG, "dqn_huber_td: null pointer");
    if (B <= 0 || A < 2) return fail(FLY_E_ARG, "dqn_huber_td: need B > 0 and A >= 2");
    hipError_t e = flyhip_launch_dqn_huber_td(q_table, act, reward, q_next, done, discount, A, B, dq, loss_part, stream);
    if (e != hipSuccess) return hip_fail(e, "dqn_huber_td launch");
    return FLY_OK;
}

int dqn_forward(const float* params, const float* params_frag, const float* x, int64_t n, float* q_out, void* stream)
{
    if (!params || !params_frag || !x || !q_out) return fail(FLY_E_ARG, "dqn_forward: null pointer");
    if (n <= 0) return fail(FLY_E_ARG, "dqn_forward: n must be > 0");
    hipError_t e = flyhip_launch_dqn_forward(params, params_frag, x, n, q_out, stream);
    if (e != hipSuccess) return hip_fail(e, "dqn_forward launch");
    return FLY_OK;
}

int dqn_act(const float* params, const float* params_frag, const float* x, int64_t n, const float* coin_u,
            const float* rand_u, float epsilon, float* act_out, float* q_out, void* stream)
{
    if (!params || !params_frag || !x || !coin_u || !rand_u || !act_out) return fail(FLY_E_ARG, "dqn_act: null pointer");
    if (n <= 0) return fail(FLY_E_ARG, "dqn_act: n must be > 0");
    hipError_t e = flyhip_launch_dqn_act(params, params_frag, x, n, coin_u, rand_u, epsilon, act_out, q_out, stream);
    if (e != hipSuccess) return hip_fail(e, "dqn_act launch");
    return FLY_OK;
}

int dqn_td_step(const float* params, const float* params_frag, const float* params_t_frag,
                const float* target_params, const float* target_params_frag, const float* obs,
                const float* next_obs, const float* act, const float* reward, const float* done, int64_t n,
                float discount, float inv_B, float* h1, float* h2, float* dz3, float* dz2, float* dz1,
                float* loss_part, void* stream)
{
    if (!params || !params_frag || !params_t_frag || !target_params || !target_params_frag || !obs || !next_obs || !act ||
        !reward || !done || !h1 || !h2 || !dz3 || !dz2 || !dz1 || !loss_part)
        return fail(FLY_E_ARG, "dqn_td_step: null pointer");
    if (n <= 0) return fail(FLY_E_ARG, "dqn_td_step: n must be > 0");
    hipError_t e = flyhip_launch_dqn_td(params, params_frag, params_t_frag, target_params, target_params_frag, obs, next_obs,
                                        act, reward, done, n, discount, inv_B, h1, h2, dz3, dz2, dz1, loss_part, stream);
    if (e != hipSuccess) return hip_fail(e, "dqn_td_step launch");
    return FLY_OK;
}

int64_t dqn_grad_workspace_floats(void) { return flyhip_dqn_grad_workspace_floats(); }

int dqn_grad_w(const float* x, const float* h1, const float* h2, const float* dz1, const float* dz2,
               const float* dz3, int64_t n, float* workspace, float* grad, int32_t accumulate, void* stream)
{
    if (!x || !h1 || !h2 || !dz1 || !dz2 || !dz3 || !workspace || !grad) return fail(FLY_E_ARG, "dqn_grad_w: null pointer");
    if (n <= 0) return fail(FLY_E_ARG, "dqn_grad_w: n must be > 0");
    hipError_t e = flyhip_launch_dqn_grad_w(x, h1, h2, dz1, dz2, dz3, n, workspace, grad, accumulate, stream);
    if (e != hipSuccess) return hip_fail(e, "dqn_grad_w launch");
    return FLY_OK;
}

int dqn_adam_soft_update(float* params, float* params_frag, float* params_t_frag, float* target_params,
                         float* target_params_frag, const int32_t* idx_frag, const int32_t* idx_t_frag,
                         const float* grad, const float* mask, float* exp_avg, float* exp_avg_sq, int32_t* step,
                         float lr, float beta1, float beta2, float eps, float tau, uint16_t* params_b3, uint16_t* params_t_b3,
                         uint16_t* target_params_b3, const int32_t* idx_b3, const int32_t* idx_t_b3, const int32_t* grad_invalid,
                         void* stream)
{
    if (!params || !params_frag || !params_t_frag || !target_params || !target_params_frag || !idx_frag || !idx_t_frag ||
        !grad || !mask || !exp_avg || !exp_avg_sq || !step)
        return fail(FLY_E_ARG, "dqn_adam_soft_update: null pointer");
    if (params_b3 && (!params_t_b3 || !target_params_b3 || !idx_b3 || !idx_t_b3))
        return fail(FLY_E_ARG, "dqn_adam_soft_update: params_b3 needs params_t_b3, target_params_b3, idx_b3 and idx_t_b3");
    hipError_t e = flyhip_launch_dqn_adam(params, params_frag, params_t_frag, target_params, target_params_frag, idx_frag,
                                          idx_t_frag, grad, mask, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, tau,
                                          params_b3, params_t_b3, target_params_b3, idx_b3, idx_t_b3, grad_invalid, stream);
    if (e != hipSuccess) return hip_fail(e, "dqn_adam_soft_update launch");
    return FLY_OK;
}

extern "C" int64_t flyhip_dqn_fused_workspace_floats(void);
extern "C" int64_t flyhip_dqn_fused_image_halves(int64_t rows);
extern "C" hipError_t flyhip_launch_dqn_fused_update(const float* P, const uint16_t* QB, const uint16_t* QTB, const float* P_tgt,
                                                     const uint16_t* QB_tgt, const void* chunks, int S, int64_t n, float discount,
                                                     float inv_B, uint16_t* images, float* workspace, float* grad, float* loss_part,
                                                     int rows_aligned16, void* stream);

int64_t dqn_fused_workspace_floats(void) { return flyhip_dqn_fused_workspace_floats(); }
int64_t dqn_fused_image_halves(int64_t rows) { return flyhip_dqn_fused_image_halves(rows); }

int dqn_fused_update(const float* params, const uint16_t* params_b3, const uint16_t* params_t_b3, const float* target_params,
                     const uint16_t* target_params_b3, const void* chunks, int32_t num_chunks, int64_t n, float discount, float inv_B,
                     uint16_t* images, float* workspace, float* grad, float* loss_part, int32_t rows_aligned16, void* stream)
{
    if (!params || !params_b3 || !params_t_b3 || !target_params || !target_params_b3 || !chunks || !images || !workspace || !grad ||
        !loss_part)
        return fail(FLY_E_ARG, "dqn_fused_update: null pointer");
    if (num_chunks <= 0 || n <= 0 || (n % 32) != 0)
        return fail(FLY_E_ARG, "dqn_fused_update: needs num_chunks > 0 and n a positive multiple of 32 (whole tiles; use dqn_td_step + dqn_grad_w otherwise)");
    hipError_t e = flyhip_launch_dqn_fused_update(params, params_b3, params_t_b3, target_params, target_params_b3, chunks, num_chunks, n,
                                                  discount, inv_B, images, workspace, grad, loss_part, rows_aligned16 ? 1 : 0, stream);
    if (e != hipSuccess) return hip_fail(e, "dqn_fused_update launch");
    return FLY_OK;
}

extern "C" int64_t flyhip_dqn_fused_h2_workspace_floats(void);
extern "C" int64_t flyhip_dqn_fused_h2_image_halves(int64_t rows);
extern "C" hipError_t flyhip_launch_dqn_fused_update_h2(const float* P, uint16_t* QH, uint16_t* QTH, const float* P_tgt, uint16_t* QH_tgt,
                                                        const int* idx_fb, const int* idx_tb, float* fsc, int* ovf, const void* chunks,
                                                        int S, int64_t n, float discount, float inv_B, uint16_t* images, float* workspace,
                                                        float* grad, float* loss_part, int rows_aligned16, int flags, void* stream);

int64_t dqn_fused_h2_workspace_floats(void) { return flyhip_dqn_fused_h2_workspace_floats(); }
int64_t dqn_fused_h2_image_halves(int64_t rows) { return flyhip_dqn_fused_h2_image_halves(rows); }

int dqn_fused_update_h2(const float* params, uint16_t* params_h2, uint16_t* params_t_h2, const float* target_params,
                        uint16_t* target_params_h2, const int32_t* idx_b3, const int32_t* idx_t_b3, float* h2_scales,
                        int32_t* h2_overflow, const void* chunks, int32_t num_chunks, int64_t n, float discount, float inv_B,
                        uint16_t* images, float* workspace, float* grad, float* loss_part, int32_t rows_aligned16, int32_t flags,
                        void* stream)
{
    if (!params || !params_h2 || !params_t_h2 || !target_params || !target_params_h2 || !idx_b3 || !idx_t_b3 || !h2_scales ||
        !h2_overflow || !chunks || !images || !workspace || !grad || !loss_part)
        return fail(FLY_E_ARG, "dqn_fused_update_h2: null pointer");
    if (num_chunks <= 0 || n <= 0 || (n % 32) != 0)
        return fail(FLY_E_ARG, "dqn_fused_update_h2: needs num_chunks > 0 and n a positive multiple of 32 (whole tiles; use dqn_td_step + dqn_grad_w otherwise)");
    if (!(inv_B > 0.0f) || !(inv_B < 1.0e30f)) return fail(FLY_E_ARG, "dqn_fused_update_h2: inv_B must be a positive finite number");
    hipError_t e = flyhip_launch_dqn_fused_update_h2(params, params_h2, params_t_h2, target_params, target_params_h2, idx_b3, idx_t_b3,
                                                     h2_scales, h2_overflow, chunks, num_chunks, n, discount, inv_B, images, workspace, grad,
                                                     loss_part, rows_aligned16 ? 1 : 0, flags & 7, stream);
    if (e != hipSuccess) return hip_fail(e, "dqn_fused_update_h2 launch");
    return FLY_OK;
}

int dp_p2p_alloc(int64_t n_floats, void** window_out)
{
    if (!window_out || n_floats <= 0 || (n_floats & 3)) return fail(FLY_E_ARG, "dp_p2p_alloc: need a pointer and n_floats a positive multiple of 4");
    hipError_t e = flyhip_p2p_alloc(n_floats, window_out);
    if (e != hipSuccess) return hip_fail(e, "dp_p2p_alloc");
    return FLY_OK;
}

int dp_p2p_free(void* window)
{
    if (!window) return FLY_OK;
    hipError_t e = hipFree(window);
    if (e != hipSuccess) return hip_fail(e, "dp_p2p_free");
    return FLY_OK;
}

int dp_ipc_export(const void* window, uint8_t handle_out[64])
{
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (!window || !handle_out) return fail(FLY_E_ARG, "dp_ipc_export: null pointer");
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, const_cast<void*>(window));
    if (e != hipSuccess) return hip_fail(e, "hipIpcGetMemHandle");
    memcpy(handle_out, &h, 64);
    return FLY_OK;
}

int dp_ipc_import(const uint8_t handle[64], void** window_out)
{
    if (!handle || !window_out) return fail(FLY_E_ARG, "dp_ipc_import: null pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, 64);
    hipError_t e = hipIpcOpenMemHandle(window_out, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return hip_fail(e, "hipIpcOpenMemHandle");
    return FLY_OK;
}

int dp_ipc_close(void* window)
{
    if (!window) return FLY_OK;
    hipError_t e = hipIpcCloseMemHandle(window);
    if (e != hipSuccess) return hip_fail(e, "hipIpcCloseMemHandle");
    return FLY_OK;
}

int dp_allreduce_p2p(float* grad, int64_t n_floats, void* const* windows, int32_t rank, int32_t world, uint32_t epoch,
                     int32_t* err, int64_t fail_slot, void* stream)
{
    if (fail_slot >= n_floats) return fail(FLY_E_ARG, "dp_allreduce_p2p: fail_slot outside the buffer");
    if (!grad || !windows || !err) return fail(FLY_E_ARG, "dp_allreduce_p2p: null pointer");
    if (world < 1 || world > 16 || rank < 0 || rank >= world) return fail(FLY_E_ARG, "dp_allreduce_p2p: bad rank / world (max 16)");
    if (n_floats <= 0 || (n_floats & 3)) return fail(FLY_E_ARG, "dp_allreduce_p2p: n_floats must be a positive multiple of 4");
    if (n_floats / 4 > 1024L * 256) return fail(FLY_E_ARG, "dp_allreduce_p2p: buffer too large for a co-resident grid");
    if (epoch == 0) return fail(FLY_E_ARG, "dp_allreduce_p2p: epochs start at 1");
    for (int r = 0; r < world; ++r)
        if (!windows[r]) return fail(FLY_E_ARG, "dp_allreduce_p2p: window %d is null", r);
    hipError_t e = flyhip_launch_p2p_allreduce(grad, n_floats, windows, rank, world, epoch, err, fail_slot, stream);
    if (e != hipSuccess) return hip_fail(e, "dp_allreduce_p2p launch");
    return FLY_OK;
}

}  // extern "C"
