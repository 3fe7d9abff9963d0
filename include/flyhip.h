/*
 * flyhip.h — C ABI of libflyhip.so, the MI355X (gfx950) replacement for the one hot path of
 * petim0/fly_bProject: the vectorised Fly environment step and the PPO rollout/update math.
 *
 * What this boundary replaces.  The reference has no FFI of its own; its de-facto native
 * boundary is the Isaac Gym tensor API that fly.py drives with raw device pointers
 * (gymtorch.unwrap_tensor):
 *     acquire_*_tensor / wrap_tensor            fly.py:380-391
 *     refresh_*_tensor                          fly.py:401-403
 *     set_dof_position_target_tensor            fly.py:657
 *     set_actor_root_state_tensor_indexed,
 *     set_dof_state_tensor_indexed              fly.py:462-468
 *     simulate + fetch_results                  fly.py:484-485
 * plus the torch op chains in fly.py:626-657, :685-805 and ppo.py:157-237.  Each entry point
 * below cites the reference lines it stands in for.
 *
 * Conventions
 *   - extern "C", plain C types only.  Every pointer is a DEVICE pointer owned by the caller
 *     (torch-ROCm allocations); the library never allocates or frees caller tensors.
 *   - Every launch takes a hipStream_t (passed as void*); nothing synchronises the host.
 *     All entry points are hipGraph-capturable.
 *   - Return 0 on success, negative FLY_E_* on error; fly_last_error() returns a
 *     thread-local message.  No exceptions cross the boundary.
 *   - Not re-entrant per handle: callers serialise calls on one handle.
 *
 * HBM layout: env-major records — every per-env record is contiguous, exactly the shapes the
 * reference's Isaac Gym tensors had, so the 8 lanes that own one env (fly_env.hip) touch one
 * or two cache lines per field and a wave (8 envs) reads a contiguous span:
 *   root      f32 [N][13]    pos xyz | quat xyzw | linvel xyz | angvel xyz   (fly.py:95-100)
 *   dof_state f32 [N][18][2] (pos, vel) per DoF, sim DoF order               (fly.py:89-90, :393)
 *   targets   f32 [N][18]    PD position targets = scaled actions (fly.py:636 `self.actions`)
 *   contact   f32 [N][11][3] net contact force of the 11 tracked bodies: 0-4 abdomen
 *                            A1A2,A3,A4,A5,A6; 5-10 leg tips LF,LH,LM,RF,RH,RM (fly.py:299-300, :386)
 *   pot, prev_pot f32 [N]    potentials (fly.py:121-123)
 *   reset, progress i64 [N]  reference dtypes kept (fly.py:175-177)
 *   actions   f32 [N][18]    policy output in [-1,1]
 *   obs       f32 [N][73]    observation rows (fly.py:799-803), staged through LDS and written
 *                            as one contiguous tile per workgroup
 *   reward    f32 [N]
 */
#ifndef FLYHIP_H
#define FLYHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLY_NUM_DOF 18
#define FLY_NUM_OBS 73
#define FLY_NUM_LEGS 6
#define FLY_NUM_ABDOMEN 5
#define FLY_NUM_CONTACT 11
#define FLY_ROOT_DIM 13

#define FLY_OK 0
#define FLY_E_ARG (-1)      /* bad argument (null pointer, bad size, misalignment) */
#define FLY_E_HIP (-2)      /* a HIP runtime call failed */
#define FLY_E_CONFIG (-3)   /* inconsistent FlyConfig */

/* GAE mode flags for ppo_td_gae (ppo.py:157-171 quirks). */
#define PPO_GAE_COMPAT 0        /* reference semantics: done mask broadcast from a [N] row, no reset of the recurrence */
#define PPO_GAE_DONE_PER_STEP 1 /* done is [T][N] instead of [N] */
#define PPO_GAE_MASK_RECURRENCE 2 /* adv_t = delta_t + gamma*lambda*done_t*adv_{t+1} (non-reference) */
#define PPO_GAE_SCAN 4          /* one wave per env, time on lanes, shuffle scan of the chunk carries: for
                                   few envs x very long rollouts; agrees with the loop to fp32 rounding */

/*
 * FlyConfig: everything fly.py hard-codes in Fly.__init__ / create_sim / create_envs
 * (fly.py:16-51, :147-167, :220-228) plus the parameters of the build-defined rigid-body
 * model that stands in for gym.simulate() (DESIGN.md "FlyDyn").  Plain floats/ints, 4-byte
 * aligned, no padding.
 */
typedef struct FlyConfig {
    int32_t num_envs;
    int32_t substeps;            /* fly.py:154 (15); flyLowGrav.py:151 (2) */
    int32_t reset_after_sim;     /* 0: reset before simulate (fly.py:660); 1: after (flyLowGrav.py:661-663) */
    int32_t reward_mode;         /* 0: standing (fly.py:750); 1: walking (fly.py:747-748, commented out upstream) */
    int32_t max_episode_length;  /* fly.py:34 */
    float dt;                    /* fly.py:16 */
    float gravity;               /* fly.py:151 (z component) */
    /* PD position drive, fly.py:224-228 */
    float kp, kd, effort, vmax;
    float joint_inertia;         /* build-defined */
    /* root rigid body, build-defined (URDF total mass 1e-3) */
    float mass;
    float inertia[3];
    /* contact model, build-defined; mu = fly.py:39-40 */
    float kc, cdamp, mu, cvisc;
    float lin_damp, ang_damp;
    float max_lin_vel, max_ang_vel;   /* per-component velocity clamps (Isaac Gym AssetOptions defaults 1000 / 64, fly.py:195) */
    /* leg kinematics, build-defined */
    float femur_len, tibia_len, alpha0, beta0;
    float dof_lo[FLY_NUM_DOF];   /* URDF joint limits, sim DoF order */
    float dof_hi[FLY_NUM_DOF];
    float dof_pose[FLY_NUM_DOF]; /* pose_default.yaml, rad */
    float leg_attach[FLY_NUM_LEGS][3];
    float leg_azimuth[FLY_NUM_LEGS];
    float leg_sigma[FLY_NUM_LEGS];
    float abdomen_pts[FLY_NUM_ABDOMEN][3];
    /* task constants, fly.py:33-51, :134 */
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
} FlyConfig;

typedef struct FlyEnv* FlyHandle;

/* Device-pointer bundle of one environment batch (layouts: file header). */
typedef struct FlyBuffers {
    float* root;        /* [N][13] */
    float* dof_state;   /* [N][18][2] */
    float* targets;     /* [N][18] */
    float* contact;     /* [N][11][3] */
    float* pot;         /* [N] */
    float* prev_pot;    /* [N] */
    float* obs;         /* [N][73] */
    float* reward;      /* [N] */
    int64_t* reset;     /* [N] */
    int64_t* progress;  /* [N] */
    /* optional episode statistics (NULL = off), updated by fly_step / fly_pack_reward: the running
     * return and length of the current episode, and per-env totals over finished episodes
     * (sum of returns, sum of lengths, count).  Not in the reference; BASELINE's second metric
     * (mean episode return) is read from here without a per-step host sync. */
    float* ep_return;   /* [N] */
    float* ep_length;   /* [N] */
    float* done_return; /* [N] sum of returns of finished episodes */
    float* done_length; /* [N] sum of lengths of finished episodes */
    float* done_count;  /* [N] number of finished episodes */
} FlyBuffers;

const char* fly_last_error(void);
int fly_abi_version(void);

/* Stands in for acquire_gym/create_sim/prepare_sim (fly.py:57-59, :139): validates and
 * uploads the config.  No tensor memory is allocated. */
int fly_create(const FlyConfig* cfg, FlyHandle* out);
int fly_destroy(FlyHandle h);

/* One whole Fly.step (fly.py:624-681) in ONE launch: K1 scale -> K2 masked reset -> K3
 * substepped integrator -> K4 obs pack -> progress+=1 -> K5 reward/done pack (K2/K3 swapped
 * when cfg.reset_after_sim).  `actions` is f32 [N][18]. */
int fly_step(FlyHandle h, const float* actions, const FlyBuffers* b, void* stream);

/* Unfused pieces, one launch each (parity tests, and callers that interleave their own work). */
/* fly.py:626-657 + isaacgym scale(): targets[e][j] = 0.5*(a+1)*(hi-lo)+lo */
int fly_scale_actions(FlyHandle h, const float* actions, float* targets, void* stream);
/* fly.py:446-480: masked reset of flagged envs; clears reset/progress. */
int fly_reset_masked(FlyHandle h, const FlyBuffers* b, void* stream);
/* fly.py:482-485 (gym.simulate + fetch_results): `substeps` semi-implicit Euler substeps. */
int fly_integrate(FlyHandle h, const FlyBuffers* b, void* stream);
/* fly.py:397-411, :771-805: observation rows + potentials. */
int fly_pack_obs(FlyHandle h, const FlyBuffers* b, void* stream);
/* fly.py:678 + :413-443, :685-768: progress+=1 (when add_progress), reward, done mask. */
int fly_pack_reward(FlyHandle h, const FlyBuffers* b, int add_progress, void* stream);

/* ppo.py:213-220: act = mu + sqrt(var)*eps; logp of the UNCLIPPED act; act_out = clip(act,-1,1).
 * mu,eps,act_out f32 [n][18]; var f32 [18]; logp_out f32 [n]. */
int ppo_sample_logprob(const float* mu, const float* var, const float* eps,
                       float* act_out, float* logp_out, int64_t n, void* stream);

/* ppo.py:157-171: target = r + gamma*v_next*done; delta = target - v;
 * adv_t = gamma*lambda*adv_{t+1} + delta_t.  reward,v,v_next,target_out,adv_out f32 [T][N];
 * done f32 [N] (or [T][N] with PPO_GAE_DONE_PER_STEP). */
int ppo_td_gae(const float* reward, const float* v, const float* v_next, const float* done,
               float gamma, float lambda, int64_t T, int64_t N,
               float* target_out, float* adv_out, int mode_flags, void* stream);

/* Advantage normalisation (opt-in; the reference uses raw advantages, ppo.py:171).
 *   ppo_adv_stats: stats[0] = sum(adv), stats[1] = sum(adv^2) over n elements (stats: >= 514 floats).
 *   ppo_adv_apply: adv = (adv - mean) / (std + eps) with mean/std from totals[0..1] over `count`
 *   elements (unbiased std, as torch.std).  Data-parallel callers all-reduce stats[0..1] and pass
 *   the global count. */
int ppo_adv_stats(const float* adv, int64_t n, float* stats, void* stream);
int ppo_adv_apply(float* adv, int64_t n, const float* totals, float count, float eps, void* stream);

/* ppo.py:233 and :237 without the per-step host sync: *score_acc += mean(reward) * score_scale;
 * action_var[j] = max(var_min, action_var[j] - var_decay) (skipped when var_decay <= 0). */
int ppo_step_bookkeeping(const float* reward, int64_t n, float* score_acc, float score_scale,
                         float* action_var, int32_t nvar, float var_decay, float var_min, void* stream);

/* ppo.py:233 and :236-237 for `rows` consecutive env steps in two small launches, bit for bit what
 * `rows` calls of ppo_step_bookkeeping on the rows of reward f32 [rows][n] leave (score terms are
 * added in row order, the variance is decayed `rows` times).  `terms` is a device scratch of
 * >= rows floats.  The rollout calls it when the score is printed and before an update instead of
 * launching the per-step form on every env step.  rows_applied (optional device int32) += rows: the
 * word mlp_forward_sample / ppo_rollout_step subtract from their row index (var_steps_base). */
int ppo_rollout_bookkeeping(const float* reward, int64_t rows, int64_t n, float* terms, float* score_acc,
                            float score_scale, float* action_var, int32_t nvar, float var_decay,
                            float var_min, int32_t* rows_applied, void* stream);

/*
 * Actor-critic MLP on the matrix cores (fp32-in/fp32-accumulate MFMA), reference ppo.py:10-102
 * (`Net.pi` / `Net.v`; 73-256-128 shared trunk, 128-64-18 actor with ELU on the mean, 128-64-1
 * critic).  `params` is the packed master buffer (MLP_PACKED_FLOATS_ABI floats, row-major layers:
 * biases are read from it) and `params_frag` the same weights in MFMA fragment order
 * (MLP_FRAG_FLOATS_ABI floats); layouts: fly_bproject_amd/csrc/mlp_layout.h.  x f32 [n][73].
 * Every output pointer is optional (NULL):
 *   mu_out [n][18]  actor mean (after its ELU)        v_out [n]  critic value
 *   out_save [n][32], h1_save [n][256], h2_save [n][128], h3_save [n][128]: activations kept
 *   for the backward pass.  These (and the dz* tensors below) are kernel-to-kernel scratch in
 *   TILE-FRAGMENT order, not row-major: per 32-row tile and 32-column tile one block of 1024
 *   floats [register group g][lane][4] with lane = 32*((col%8)/4) + row%32, g = (col%32)/8,
 *   blocks ordered [row tile][column tile]; allocate ceil32(n) rows.
 * GEMM arithmetic: with params_b3 = NULL the layers run on v_mfma_f32_32x32x2_f32 (fp32 products,
 * fp32 accumulate).  With params_b3 = the weights as three bf16 terms each (w = w0 + w1 + w2 exactly,
 * MLP_PB_HALVES_ABI 16-bit words, layout in mlp_layout.h; mlp_adam_step maintains it) the same GEMMs
 * run on v_mfma_f32_32x32x16_bf16 with both operands split that way and six product terms per k
 * block, fp32 accumulate: measured error below the fp32 MFMA chain's (tools/bf16x3_gemm.hip), 2.7x
 * less matrix time.  Inputs, outputs and saved activations are fp32 either way.
 */
#define MLP_PB_HALVES_ABI 221184
#define MLP_PTB_HALVES_ABI 159744
#define MLP_PACKED_FLOATS_ABI 74272
#define MLP_FRAG_FLOATS_ABI 73728
#define MLP_FRAG_T_FLOATS_ABI 53248
int mlp_forward(const float* params, const float* params_frag, const float* x, int64_t n,
                float* mu_out, float* v_out, float* out_save, float* h1_save, float* h2_save,
                float* h3_save, const uint16_t* params_b3, void* stream);
/* ppo.py:214-220 in ONE launch: mu = Net.pi(x), act = mu + sqrt(var)*eps, log-prob of the unclipped
 * act, act_out = clip(act, -1, 1).  eps, act_out f32 [n][18]; var f32 [18]; logp_out f32 [n];
 * mu_out f32 [n][18] and v_out f32 [n] (= Net.v(x), the same rows the critic pass of ppo.py:158-159
 * would recompute with unchanged weights) are optional.  The variance used is `var` after
 * `var_steps` applications of v <- max(var_min, v - var_decay) (ppo.py:236-237), computed in the
 * kernel without touching `var`: the caller may apply the decays of a whole rollout to the tensor
 * later (ppo_rollout_bookkeeping).  var_steps = 0 uses `var` as is.  var_steps_base (optional device int32):
 * the pending decays are var_steps - *var_steps_base -- pass the rollout's row index as var_steps and the
 * word ppo_rollout_bookkeeping advances: the launch arguments then never change for a given row, so the
 * launch can be replayed from a captured hipGraph while the bookkeeping stays deferred. */
int mlp_forward_sample(const float* params, const float* params_frag, const float* x, int64_t n,
                       const float* eps, const float* var, int32_t var_steps, float var_decay,
                       float var_min, float* act_out, float* logp_out, float* mu_out, float* v_out,
                       const uint16_t* params_b3, const int32_t* var_steps_base, void* stream);

/* One env step of the rollout (ppo.py:213-230) in ONE launch: mlp_forward_sample on x f32 [N][73]
 * (N = the handle's num_envs; fp32 MFMA arithmetic) writing act_out [N][18], logp_out [N] and the
 * optional v_out [N], then fly_step of the same envs with those actions (`b` as for fly_step: its
 * obs / reward rows are where the step's results go).  Workgroup k does both for envs 32k..32k+31,
 * so nothing waits on another workgroup.  Bit for bit what the two calls leave.  params_b3 != NULL runs the
 * policy on the bf16x3 GEMM arithmetic (as mlp_forward_sample with params_b3). */
int ppo_rollout_step(FlyHandle h, const FlyBuffers* b, const float* params, const float* params_frag,
                     const float* x, const float* eps, const float* var, int32_t var_steps,
                     float var_decay, float var_min, float* act_out, float* logp_out, float* v_out,
                     const uint16_t* params_b3, const int32_t* var_steps_base, void* stream);

/* T consecutive steps of the rollout (ppo.py:204-237) in ONE launch: ppo_rollout_step for t = 0 .. T-1 on row t of
 * the rollout tensors -- obs_ring f32 [T+1][N][73] (row 0 = the first observation; row t+1 receives step t's),
 * eps_all / act_all f32 [T][N][18], logp_all / v_ring f32 [T][N] (v_ring may be [T+1][N]), reward_all f32 [T][N] --
 * with the env state carried in registers from step to step (`b` gives the state tensors; its obs / reward members
 * are ignored).  Envs of a 32-env tile depend on no other tile and the policy is constant inside a rollout, so each
 * workgroup runs its own tile's T steps.  `var` must ALREADY be the variance of row 0: the launch is the first device work of
 * its rollout and decays it once per step from there; `rows_applied` is ignored (kept for the ABI's shape: the per-step entry
 * ppo_rollout_step is the one that subtracts it) -- a caller with decays still pending applies them before this call.
 * Bit for bit what T calls of ppo_rollout_step leave.  The launch runs for T x (one step's time): keep T <= a few
 * thousand.  reset_rows / progress_rows (both or neither, int64 [T][N]): step t writes its reset / progress flags to row t
 * instead of b->reset / b->progress (which are then only READ, once, at the start), so that a host that walks the rollout
 * step by step after the launch finds the per-step buffers of fly.py:175-177 at row t. */
int ppo_rollout_all(FlyHandle h, const FlyBuffers* b, const float* params, const float* params_frag, float* obs_ring,
                    const float* eps_all, const float* var, float var_decay, float var_min, float* act_all,
                    float* logp_all, float* v_ring, float* reward_all, int32_t T, const int32_t* rows_applied,
                    const uint16_t* params_b3, int64_t* reset_rows, int64_t* progress_rows, void* stream);


/*
 * One PPO minibatch backward (ppo.py:184-197), three launches on the saved activations of
 * mlp_forward (n rows = the minibatch; inv_batch = 1/n of the GLOBAL minibatch):
 *   mlp_backward_dx : loss gradient at the outputs (clipped surrogate with torch's subgradient
 *                     conventions + scalar-mean Huber, ppo.py:191-194) and the dX chain;
 *                     writes dz4 [n][32], dz3 [n][128], dz2 [n][128], dz1 [n][256] and
 *                     loss_part [ceil(n/32)][2] (sum of -min(surr1,surr2), sum of huber).
 *   mlp_grad_w      : dW = dZ^T A and db = colsum(dZ) for all four layers into `grad`
 *                     (packed layout of `params`); `workspace` holds
 *                     mlp_grad_workspace_floats() floats.  With norm_mask/norm_ws/norm_step
 *                     non-NULL (single rank: nothing sits between this call and the optimizer)
 *                     the reduction also leaves the clip-norm partial sums in norm_ws (>= 1280
 *                     floats) and advances *norm_step; pass norm_ready = 1 to mlp_adam_step then.
 *                     gemm_b3 != 0 runs the products on the bf16 matrix pipe with both operands
 *                     split into three bf16 terms at staging (the bf16x3 arithmetic of mlp_forward).
 *                     `err` (optional) = the err word of mlp_forward_backward: while it is nonzero
 *                     grad[76] (padding column 76 of W1 row 0: masked, never a parameter) is set
 *                     to 1 instead of 0 and *norm_step is not advanced.  The flag rides inside the
 *                     gradient so that a data-parallel all-reduce (sum) spreads it to every rank.
 *   mlp_adam_step   : grad *= grad_scale; clip_grad_norm_(max_norm); Adam with torch defaults on
 *                     `params`; every updated weight is also scattered into params_frag /
 *                     params_t_frag through idx_frag / idx_t_frag (int32 [MLP_PACKED_FLOATS_ABI],
 *                     -1 = no copy); with params_b3 != NULL also its three bf16 terms into
 *                     params_b3 / params_t_b3 (term 0 at idx_b3[i] / idx_t_b3[i], terms 1 and 2
 *                     512 and 1024 16-bit words later).  `mask` (packed layout, 0/1)
 *                     freezes padding and structural zeros.  `step` is a device int counter;
 *                     `norm_ws` is a device scratch of >= 1280 floats, norm_ws[0] returns the
 *                     pre-clip gradient norm.  step_out != NULL (data-parallel ranks, the gradient
 *                     comes out of an all-reduce): ONE launch -- every workgroup sums the masked
 *                     gradient itself (norm_ready is ignored) and the counter ping-pongs: *step is
 *                     only read, *step_out = *step + 1 is written; pass the two words alternately.
 *                     FAIL CLOSED: with grad[76] != 0 (see mlp_grad_w), or with *grad_invalid != 0
 *                     (optional device word, NULL = none: the err word of dp_allreduce_p2p, which ANY of
 *                     its workgroups sets when it gave up on a peer -- so a gradient that is only partly
 *                     reduced is refused as well), the call changes nothing -- parameters, moments and
 *                     *step keep their values.
 */
int64_t mlp_grad_workspace_floats(void);
int mlp_backward_dx(const float* params_t_frag, const float* out_saved, const float* h1_saved,
                    const float* h2_saved, const float* h3_saved, const float* action,
                    const float* old_logp, const float* adv, const float* target, const float* var,
                    int64_t n, float inv_batch, float clip, float* dz4, float* dz3, float* dz2,
                    float* dz1, float* loss_part, const uint16_t* params_t_b3, void* stream);

/* The forward and the loss + dX chain of one PPO minibatch (ppo.py:184-197: net.pi / net.v on the
 * minibatch, ratio, clipped surrogate, smooth_l1, backward down to the first layer's pre-activations)
 * -- i.e. mlp_forward (activations saved) and mlp_backward_dx of the same n rows -- in ONE launch: both are
 * row-local, so the backward workgroup of a 32-row tile starts as soon as that tile's forward has
 * published its flag, and fills the slots the forward launch's ragged tail would leave idle.
 * Results are bit-identical to the two separate calls.  `flags` int32 [ceil(n/32)] device scratch
 * (zero-initialised; re-zero it before `epoch` wraps), `epoch` in [1, 2^27) a value that differs
 * from every earlier call on these flags (a counter), `err` int32 [1] device word, normally 0:
 * 1 = a workgroup gave up waiting for its tile (the wait is bounded, a lost flag cannot hang the
 * device), 2 = (coherent = 0 only) a tile's forward and backward workgroups were not on the same XCD.
 * Assumptions, stated because HIP promises neither: LIVENESS needs workgroups to be dispatched in
 * blockIdx order (a consumer's producer is resident or done before it polls; otherwise err = 1, never a
 * hang).  VISIBILITY: coherent != 0 (the product default) hands the tile over with sc1 write-through
 * stores, drained, an sc1 flag, and sc1 L1-bypassing loads on the consumer -- no placement assumption;
 * coherent = 0 uses plain accesses and REQUIRES producer and consumer on one XCD (round-robin placement,
 * checked through the XCC id, err = 2) and a consumer L1 that does not hold the tile's lines.
 * Non-zero err => the dZ rows of that call are invalid: pass `err` to mlp_grad_w, which then marks the
 * gradient (grad[76] = 1, a masked padding element) so that mlp_adam_step skips the step on every rank
 * that receives it -- and redo the minibatch with mlp_forward + mlp_backward_dx.  params_b3 /
 * params_t_b3 (both or neither) select the bf16x3 GEMM arithmetic, see mlp_forward. */
int mlp_forward_backward(const float* params, const float* params_frag, const float* params_t_frag,
                         const float* x, int64_t n, float* out_save, float* h1_save, float* h2_save,
                         float* h3_save, const float* action, const float* old_logp, const float* adv,
                         const float* target, const float* var, float inv_batch, float clip,
                         float* dz4, float* dz3, float* dz2, float* dz1, float* loss_part,
                         int32_t* flags, int32_t epoch, int32_t* err, const uint16_t* params_b3,
                         const uint16_t* params_t_b3, int32_t coherent, void* stream);
int mlp_grad_w(const float* x, const float* h1_saved, const float* h2_saved, const float* h3_saved,
               const float* dz1, const float* dz2, const float* dz3, const float* dz4, int64_t n,
               float* workspace, float* grad, const float* norm_mask, float* norm_ws, int32_t* norm_step,
               const int32_t* err, int32_t gemm_b3, void* stream);

/*
 * The whole gradient of one PPO minibatch (ppo.py:184-197: net.pi / net.v on the minibatch, ratio, clipped surrogate,
 * smooth_l1, loss.mean().backward()) in ONE persistent launch plus the fixed-order reduction: what
 * mlp_forward_backward + mlp_grad_w compute, without handing activations or dZ through HBM.  One workgroup per CU
 * walks 32-row tiles; a tile's forward, loss, dX chain and dW / db contributions all happen in that workgroup with
 * the activations in LDS (bf16x3 term planes that serve the row reads of the forward / dX GEMMs and, through the
 * transposing LDS read, the row-reducing dW GEMMs) and dW accumulated in registers across the workgroup's tiles;
 * each workgroup writes one partial slab (the packed-gradient layout) into `workspace`
 * (mlp_fused_workspace_floats() floats), which the reduction sums into `grad`.  bf16x3 arithmetic only
 * (params_b3 / params_t_b3 required; see mlp_forward).  norm_mask / norm_ws / norm_step as in mlp_grad_w (all
 * three or none).  loss_part f32 [ceil(n/32)][2] as in mlp_backward_dx.  There is no cross-workgroup hand-off
 * inside the launch, hence no err word.
 * `debug_dump` (normally NULL): 8 device pointers {out, h1, h2, h3, dz4, dz3, dz2, dz1} sized and laid out like
 * mlp_forward's saves / mlp_backward_dx's outputs; the launch then also writes the chain's values there (tests).
 */
int64_t mlp_fused_workspace_floats(void);
int mlp_fused_grad(const float* params, const uint16_t* params_b3, const uint16_t* params_t_b3, const float* x, int64_t n,
                   const float* action, const float* old_logp, const float* adv, const float* target, const float* var,
                   float inv_batch, float clip, float* workspace, float* grad, const float* norm_mask, float* norm_ws,
                   int32_t* norm_step, float* loss_part, float* const* debug_dump, void* stream);
int mlp_adam_step(float* params, float* params_frag, float* params_t_frag, const int32_t* idx_frag,
                  const int32_t* idx_t_frag, const float* grad, const float* mask, float* exp_avg,
                  float* exp_avg_sq, int32_t* step, float lr, float beta1, float beta2, float eps,
                  float max_norm, float grad_scale, float* norm_ws, int32_t norm_ready,
                  uint16_t* params_b3, uint16_t* params_t_b3, const int32_t* idx_b3,
                  const int32_t* idx_t_b3, int32_t* step_out, const int32_t* grad_invalid,
                  uint16_t* params_h2, uint16_t* params_t_h2, float* h2_scales, int32_t h2_rescale, void* stream);

/*
 * The same gradient in the fp16x2 arithmetic (csrc/mlp_fused_h2.inc): every fp32 operand of a GEMM is carried as TWO fp16 terms of
 * its value times a per-tensor-class power of two, a k block is THREE v_mfma_f32_32x32x16_f16 products (fp32 accumulate) instead of
 * the six bf16 products of mlp_fused_grad: half the matrix time.  What it needs besides mlp_fused_grad's arguments:
 *   params_h2 / params_t_h2  the weights as two fp16 terms of w * s_layer (MLP_PH_HALVES_ABI / MLP_PTH_HALVES_ABI 16-bit words: the
 *                 layout of params_b3 / params_t_b3 with 1024-word blocks instead of 1536: term 0, term 1); mlp_adam_step
 *                 splits every updated weight into them (params_h2 != NULL; it needs params_b3 and its index maps too) under the
 *                 layer scales h2_scales[8 .. 11]; a call with h2_rescale != 0 first re-derives those scales from the weights as they
 *                 stand (max(max |w_layer|, 2^-4) -> [2^11, 2^12): 16x of headroom) and publishes them.  An Adam step moves a weight by
 *                 <= 3.2 lr, so a rescale step at least every 0.9 / (3.2 lr) steps (the host: every 64) keeps the planes inside fp16;
 *   h2_scales     device float [MLP_H2_SCALE_FLOATS_ABI]: s[c] at [c], 1 / s[c] at [16 + c] for the classes c = 0 X, 1 H1, 2 H2,
 *                 3 H3, 4 dZ4, 5 dZ3, 6 dZ2, 7 dZ1, 8 .. 11 W1 .. W4; [32 + c] = the largest |scaled value| the last launch saw.
 *                 The launch READS the table and its reduction WRITES the activation / gradient entries for the NEXT launch (this
 *                 launch's class maxima steered into [2^9, 2^10)) unless `freeze` != 0;
 *   h2_overflow   device int32, STICKY: set when a value of some launch did not fit fp16 (class maximum >= 65504).  From then on
 *                 every launch marks its gradient invalid (grad[76] = 1, as mlp_grad_w does with a nonzero `err`) and does not
 *                 advance *norm_step, so mlp_adam_step refuses the step: the caller reads the step counter at the end of the
 *                 update, clears the word and redoes the refused steps, in order, with mlp_fused_grad;
 *   mlp_h2_rescale: the same scales and both plane buffers from `params` without an optimizer step (one slow block): before the
 *                 first step and after any out-of-band change of the weights.
 * `workspace` holds mlp_fused_h2_workspace_floats() floats.  A first launch on a new network should be preceded by a few launches
 * whose result is discarded (scales converge in one launch per class that overflowed; fly_bproject_amd/policy.py: calibrate_h2).
 */
#define MLP_PH_HALVES_ABI 147456
#define MLP_PTH_HALVES_ABI 106496
#define MLP_H2_SCALE_FLOATS_ABI 48
int mlp_h2_rescale(const float* params, const int32_t* idx_b3, const int32_t* idx_t_b3, uint16_t* params_h2, uint16_t* params_t_h2,
                   float* h2_scales, void* stream);
int64_t mlp_fused_h2_workspace_floats(void);
int mlp_fused_grad_h2(const float* params, const uint16_t* params_h2, const uint16_t* params_t_h2, float* h2_scales,
                      int32_t* h2_overflow, int32_t freeze, const float* x, int64_t n, const float* action,
                      const float* old_logp, const float* adv, const float* target, const float* var, float inv_batch,
                      float clip, float* workspace, float* grad, const float* norm_mask, float* norm_ws, int32_t* norm_step,
                      float* loss_part, float* const* debug_dump, void* stream);


/*
 * DQN variant (reference UselessFiles/dqn.py, BASELINE configs[4]).
 *   dqn_eps_greedy (dqn.py:89-100): per env, the FIRST maximal entry of its Q row -> idx/(A-1);
 *       act = coin_u < epsilon ? rand_u : that; act_out = 2*(act-0.5).  q f32 [n][A]; coin_u,
 *       rand_u, act_out f32 [n].
 *   dqn_huber_td (dqn.py:68-79): idx = round(0.5*(act+1)*(A-1)); target = reward + discount *
 *       max_a q_next * done; loss = mean smooth_l1(q_table[b,idx] - target).  Writes the gradient
 *       of that loss w.r.t. q_table into dq [B][A] and per-256-row partial loss sums into
 *       loss_part [ceil(B/256)].
 */
int dqn_eps_greedy(const float* q, const float* coin_u, const float* rand_u, float epsilon, int32_t A,
                   float* act_out, int64_t n, void* stream);
int dqn_huber_td(const float* q_table, const float* act, const float* reward, const float* q_next,
                 const float* done, float discount, int32_t A, int64_t B, float* dq, float* loss_part,
                 void* stream);

/*
 * DQN variant on the matrix cores: the Q-network of UselessFiles/dqn.py:17-29 (73 -> 256 -> 256 -> 18, LeakyReLU;
 * D1: num_obs is the environment's 73) as packed parameters in the layout of csrc/dqn_layout.h
 * (DQN_PACKED_FLOATS_ABI floats; `params_frag` / `params_t_frag` = its fragment-ordered copies of
 * DQN_FRAG_FLOATS_ABI / DQN_FRAG_T_FLOATS_ABI floats).  fp32 MFMA arithmetic.
 *   dqn_forward   : q_out f32 [n][18] = Net(x), x f32 [n][73]                                   (dqn.py:28)
 *   dqn_act       : dqn.py:89-100 in ONE launch: Net(x), per row the FIRST maximal entry -> idx/17, mixed
 *                   with rand_u where coin_u < epsilon, mapped to [-1,1]; act_out f32 [n]; q_out optional.
 *   dqn_td_step   : dqn.py:64-79 for ONE sampled replay step (n rows, pointers straight into the replay
 *                   ring: no gather): q_target(next_obs).max(1), q(obs), target = reward + discount * max *
 *                   done, smooth_l1 (loss_part f32 [ceil(n/32)]: per-tile sums of the Huber terms), its
 *                   gradient at the Q table scaled by inv_B (1 / rows of the WHOLE batch), and the backward
 *                   chain down to the first layer's pre-activations.  Leaves h1, h2, dz1, dz2 f32 [n32][256]
 *                   and dz3 f32 [n32][32] (n32 = n rounded up to 32; tile-fragment order) for dqn_grad_w.
 *   dqn_grad_w    : dW / db of the three layers from those tensors.  One call per sampled step of a batch:
 *                   accumulate bit 0 = add to the partial slabs in `workspace` (every step but the first),
 *                   bit 1 = last step: reduce the slabs into `grad` (packed layout) now.  A single-step batch
 *                   passes 2.  `workspace` holds dqn_grad_workspace_floats() floats and must not be touched
 *                   between the calls of one batch.
 *   dqn_adam_soft_update : dqn.py:81-84: Adam (torch defaults, no clipping) on `params` with the refresh
 *                   of its fragment copies, then target = target * tau + params * (1 - tau)
 *                   (dqn.py:33-36) with the target's forward fragment copy.  `mask` (packed, 0/1) freezes
 *                   padding; `step` device int counter; idx_* int32 [DQN_PACKED_FLOATS_ABI] (-1: no copy).
 *                   params_b3 != NULL (all five or none): also keeps the three-term bf16 planes the fused update
 *                   streams -- online forward (DQN_QB_HALVES_ABI 16-bit words) and transposed (DQN_QTB_HALVES_ABI),
 *                   target forward -- term 0 at idx_b3[i] / idx_t_b3[i], terms 1 and 2 512 and 1024 words later
 *                   (layout: csrc/dqn_layout.h).
 *                   FAIL CLOSED: grad_invalid != NULL and *grad_invalid != 0 (dqn_fused_update_h2's overflow word) -> nothing
 *                   moves and the step counter stays, for this call and every later one until the host clears the word.
 */
#define DQN_QB_HALVES_ABI 282624
#define DQN_QTB_HALVES_ABI 221184
#define DQN_PACKED_FLOATS_ABI 94752
#define DQN_FRAG_FLOATS_ABI 94208
#define DQN_FRAG_T_FLOATS_ABI 73728
int dqn_forward(const float* params, const float* params_frag, const float* x, int64_t n, float* q_out, void* stream);
int dqn_act(const float* params, const float* params_frag, const float* x, int64_t n, const float* coin_u,
            const float* rand_u, float epsilon, float* act_out, float* q_out, void* stream);
int dqn_td_step(const float* params, const float* params_frag, const float* params_t_frag,
                const float* target_params, const float* target_params_frag, const float* obs,
                const float* next_obs, const float* act, const float* reward, const float* done, int64_t n,
                float discount, float inv_B, float* h1, float* h2, float* dz3, float* dz2, float* dz1,
                float* loss_part, void* stream);
int64_t dqn_grad_workspace_floats(void);
int dqn_grad_w(const float* x, const float* h1, const float* h2, const float* dz1, const float* dz2,
               const float* dz3, int64_t n, float* workspace, float* grad, int32_t accumulate, void* stream);
/* The whole gradient of ONE update (dqn.py:64-80) over all its sampled replay steps in two persistent launches + one reduction,
 * bf16x3 arithmetic (params_b3 / params_t_b3 / target_params_b3: the planes dqn_adam_soft_update maintains): `chunks` is a DEVICE
 * array of num_chunks records {obs, next_obs, act, reward, done} (five device pointers: rows of the replay ring where they lie, n rows
 * each, n a multiple of 32).  A 32-row tile's target forward, online forward, TD target, Huber loss, dX chain and dW1 / dW3 run in the
 * workgroup that owns the tile, with dW accumulated in registers across ALL tiles of the update; only the two plane images the
 * 256 x 256 layer's dW needs (H1, dZ2: 96 KB per tile) go to `images` (dqn_fused_image_halves(num_chunks * n) 16-bit words), which
 * the second launch streams back once.  `workspace`: dqn_fused_workspace_floats() floats of partial slabs; `grad`: the packed
 * gradient (dqn_adam_soft_update's input); loss_part f32 [num_chunks * n / 32]: per-tile sums of the Huber terms.
 * rows_aligned16 != 0: the caller vouches that every obs / next_obs pointer of `chunks` is 16-byte aligned (the table lives on the
 * device, the library cannot look): the rows then travel by LDS-DMA, requested a pass ahead; 0: plain loads where they are needed. */
int64_t dqn_fused_workspace_floats(void);
int64_t dqn_fused_image_halves(int64_t rows);
int dqn_fused_update(const float* params, const uint16_t* params_b3, const uint16_t* params_t_b3, const float* target_params,
                     const uint16_t* target_params_b3, const void* chunks, int32_t num_chunks, int64_t n, float discount,
                     float inv_B, uint16_t* images, float* workspace, float* grad, float* loss_part,
                     int32_t rows_aligned16, void* stream);
/* The same gradient in the fp16x2 arithmetic of mlp_fused_grad_h2 (fly_bproject_amd/csrc/dqn_fused_h2.inc): two fp16 terms per GEMM
 * operand, three MFMA products per k block, two-plane images (64 KB per tile).  ONE call = weight planes + weight scales from the
 * CURRENT params / target_params (params_h2 DQN_QH_HALVES_ABI, params_t_h2 DQN_QTH_HALVES_ABI, target_params_h2 DQN_QH_HALVES_ABI
 * 16-bit words: scratch the library owns the contents of; idx_b3 / idx_t_b3 = dqn_adam_soft_update's plane maps), the two persistent
 * launches, the slab reduction, and the NEXT call's activation / gradient scales from this call's class maxima.
 *   h2_scales    device float [MLP_H2_SCALE_FLOATS_ABI]: s[c] at [c], 1 / s[c] at [16 + c], the last call's maximum of |scaled value|
 *                at [32 + c]; classes 0 X, 1 H1, 2 H2, 5 dZ2, 6 dZ1 (lagged), 8 .. 10 / 11 .. 13 online / target W1 .. W3 (exact, per
 *                call).  Start it at 1.0 / 1.0 / 0 and run two calls with flags = 2 on the first update's chunks.
 *   h2_overflow  device int: set to 1 when a value of THIS call did not fit fp16 under the lagged scales -- `grad` is then invalid.
 *                Either read it before the optimizer launch (clear it, form the gradient with dqn_fused_update -- bf16x3; nothing has
 *                been applied yet -- and calibrate again), or hand it to dqn_adam_soft_update as grad_invalid and look later: the
 *                optimizer then refuses every update until the word is cleared (the device step counter says how many).
 *   flags        bit 0: leave the lagged scales unchanged; bit 1: calibration pass (class maxima -> scales only; `grad` not written).
 *                bit 2: dZ2's plane image is not written: the chain kernel leaves a 2304-byte record per tile (dq s_z2 and the action per
 *                row, LeakyReLU' flags) and the dW2 kernel rebuilds dZ2 from it and the fp32 W3 rows of `params` (a third less image traffic).
 * `workspace`: dqn_fused_h2_workspace_floats() floats; `images`: dqn_fused_h2_image_halves(num_chunks * n) 16-bit words. */
#define DQN_QH_HALVES_ABI 188416
#define DQN_QTH_HALVES_ABI 147456
int64_t dqn_fused_h2_workspace_floats(void);
int64_t dqn_fused_h2_image_halves(int64_t rows);
int dqn_fused_update_h2(const float* params, uint16_t* params_h2, uint16_t* params_t_h2, const float* target_params,
                        uint16_t* target_params_h2, const int32_t* idx_b3, const int32_t* idx_t_b3, float* h2_scales,
                        int32_t* h2_overflow, const void* chunks, int32_t num_chunks, int64_t n, float discount, float inv_B,
                        uint16_t* images, float* workspace, float* grad, float* loss_part, int32_t rows_aligned16, int32_t flags,
                        void* stream);
int dqn_adam_soft_update(float* params, float* params_frag, float* params_t_frag, float* target_params,
                         float* target_params_frag, const int32_t* idx_frag, const int32_t* idx_t_frag,
                         const float* grad, const float* mask, float* exp_avg, float* exp_avg_sq, int32_t* step,
                         float lr, float beta1, float beta2, float eps, float tau, uint16_t* params_b3,
                         uint16_t* params_t_b3, uint16_t* target_params_b3, const int32_t* idx_b3,
                         const int32_t* idx_t_b3, const int32_t* grad_invalid, void* stream);


/*
 * One-shot peer-to-peer all-reduce (sum) of the packed gradient across the data-parallel ranks of ONE node, in
 * place of torch.distributed.all_reduce (RCCL) between mlp_grad_w and mlp_adam_step -- the call that stands where
 * the reference's single-process update has nothing to exchange (ppo.py:196-199).  Every rank allocates a
 * fine-grained window (dp_p2p_alloc: 2 x n_floats of publish space + flags), exports it (dp_ipc_export, 64 opaque
 * bytes to hand to the peers by any means), opens the peers' (dp_ipc_import) and passes the `world` window
 * pointers -- its own at index `rank` -- in rank order.  dp_allreduce_p2p(grad, ...) with epoch = 1, 2, 3, ...
 * (the same on every rank): one launch publishes grad, waits for every rank's flag of this epoch (bounded:
 * *err = 1 instead of a hang; the budget is 2^FLY_P2P_POLL_LOG2 polls, default 2^24 = several seconds) and leaves
 * the sum over ranks, added in rank order (bit-identical on every rank), in grad.  world <= 16.
 * fail_slot >= 0: a rank whose wait expired also writes 1.0f to grad[fail_slot] (mlp_adam_step's "invalid
 * gradient" mark, element 76 of the packed gradient), so its optimizer launch refuses the un-reduced gradient;
 * the caller then reads *err (nonzero) and stops: a lost peer is fatal for the run, it is never papered over.
 */
int dp_p2p_alloc(int64_t n_floats, void** window_out);
int dp_p2p_free(void* window);
int dp_ipc_export(const void* window, uint8_t handle_out[64]);
int dp_ipc_import(const uint8_t handle[64], void** window_out);
int dp_ipc_close(void* window);
int dp_allreduce_p2p(float* grad, int64_t n_floats, void* const* windows, int32_t rank, int32_t world,
                     uint32_t epoch, int32_t* err, int64_t fail_slot, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLYHIP_H */
