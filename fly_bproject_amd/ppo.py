"""`Net` and `PPO`: the reference's agent (ppo.py:10-285) on the MI355X-native environment.

Same classes, attributes and methods as the reference (`PPO(args)`, `.run() -> bool`, `.update()`,
`.make_data()`, `.save(suffix)`, `.net.pi/.v`, state-dict keys `shared_net.{0,2}`,
`to_mean.{0,2}`, `to_value.{0,2}`), same hyper-parameters and the same quirks,
each named where it is reproduced.

What changed underneath (MI355X-first, not a translation):
  * rollout rows live in one ring `[T+1, N, 73]`; `all_obs = ring[:T]`, `all_next_obs = ring[1:]`
    are views (next_obs[t] is obs[t+1] by construction, ppo.py:210/:228), and the env kernel
    writes each observation row straight into the ring (`Fly.bind_obs`), so the two 2.4 MB
    row copies per step are gone;
  * the policy forward, sampling, log-prob and clip are ONE launch writing the action / log-prob rows
    of the rollout in place (`mlp_forward_sample`);
  * TD target + GAE is one kernel (`ppo_td_gae`) instead of a T-long Python loop, and the two
    critic passes over the rollout collapse into one pass over the ring;
  * no per-step host sync: the score accumulates on the device and is read every
    `num_eval_freq` steps, when it is printed;
  * the update runs on the MFMA kernels of csrc/mlp_mfma.hip (forward, loss + dX chain, dW, clip +
    Adam) over contiguous minibatch slices of the rollout;
  * data-parallel training: one packed-gradient all-reduce over RCCL per optimizer step.
"""
import ctypes as C
import collections
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .fly import Fly
from .params import NUM_DOF


class Net(nn.Module):
    """ppo.py:10-102: shared 73-256-128 (ELU); actor 128-64-18 with ELU after BOTH layers;
    critic 128-64-1.  `pi` and `v` each run the shared trunk."""

    def __init__(self, num_obs, num_act):
        super().__init__()
        self.shared_net = nn.Sequential(nn.Linear(num_obs, 256), nn.ELU(), nn.Linear(256, 128), nn.ELU())
        self.to_mean = nn.Sequential(nn.Linear(128, 64), nn.ELU(), nn.Linear(64, num_act), nn.ELU())
        self.to_value = nn.Sequential(nn.Linear(128, 64), nn.ELU(), nn.Linear(64, 1))

    _policy = None      # PackedPolicy, attached by PPO: inference then runs on the MFMA kernel

    def pi(self, x):
        if self._policy is not None and not torch.is_grad_enabled():
            return self._policy.forward(x, want_mu=True, want_v=False)[0]
        return self.to_mean(self.shared_net(x))

    def v(self, x):
        if self._policy is not None and not torch.is_grad_enabled():
            return self._policy.forward(x, want_mu=False, want_v=True)[1]
        return self.to_value(self.shared_net(x))


def diag_gauss_logprob(mu, action, var):
    """log N(action; mu, diag(var)) exactly as MultivariateNormal(mu, scale_tril=cholesky(diag(var)))
    evaluates it (ppo.py:185-189): -0.5 (k log 2pi + |L^-1 (a-mu)|^2) - sum log L_jj."""
    L = torch.sqrt(var)
    x = (action - mu) / L
    M = (x * x).sum(-1)
    half_log_det = torch.log(L).sum(-1)
    return -0.5 * (mu.shape[-1] * 1.8378770664093453 + M) - half_log_det


class PPO:
    def __init__(self, args, env=None):
        self.args = args
        self.env = env if env is not None else Fly(args)           # ppo.py:110
        self.num_acts = self.env.num_act
        self.num_obs = self.env.num_obs
        self.epoch = 5
        self.lr = 0.001
        self.gamma = 0.99
        self.lmbda = 0.95
        self.clip = 0.2
        self.mini_batch_size = 40960
        self.chuck_number = 16
        n = int(args.num_envs)
        self.mini_chunk_size = self.mini_batch_size // n            # ppo.py:120 (Q9: 0 for N > 40960)
        if self.mini_chunk_size < 1:
            raise ValueError("num_envs=%d gives mini_chunk_size 0 (ppo.py:120); use num_envs <= 40960" % n)
        print("mini_chunk_size: ", self.mini_chunk_size)
        self.rollout_size = self.mini_chunk_size * self.chuck_number
        print("rollout_size: ", self.rollout_size)
        self.num_eval_freq = 100
        self.mini_batch_number = 0

        dev = self.device = self.env.device
        T = self.rollout_size
        # ppo.py:132-138, with all_obs / all_next_obs as two views of one ring
        self._obs_ring = torch.zeros((T + 1, n, self.num_obs), device=dev)
        self.all_obs = self._obs_ring[:T]
        self.all_next_obs = self._obs_ring[1:]
        # v(obs_t) falls out of the rollout's policy launch (same kernel, same weights, same rows as the
        # critic pass of ppo.py:158-159), so make_data only has to evaluate the last next_obs
        self._v_ring = torch.zeros((T + 1, n, 1), device=dev)
        self._v_have, self._v_version = 0, -1
        self.reuse_rollout_values = bool(getattr(args, "reuse_rollout_values", True))
        self.all_acts = torch.zeros((T, n, self.num_acts), device=dev)
        self.all_reward = torch.zeros((T, n, 1), device=dev)
        self._all_done = None                                        # see the all_done property (Q1)
        self.all_log_prob = torch.zeros((T, n), device=dev)
        self.all_advantage = torch.zeros((T, n, 1), device=dev)
        self._target = torch.zeros((T, n, 1), device=dev)
        self._eps_all = torch.zeros((T, n, self.num_acts), device=dev)   # one normal_() per rollout
        self._mu = torch.zeros((n, self.num_acts), device=dev)
        self.normalize_advantage = bool(getattr(args, "normalize_advantage", False))
        self._adv_stats = torch.zeros(514, device=dev)
        self.use_graph = bool(getattr(args, "graph", False))
        # one launch per ROLLOUT (ppo_rollout_all): each workgroup loops over the T steps of its own 32 envs.  The device then
        # runs ahead of the host's step count inside a rollout, but everything `run()` reads per step is a ROW the launch wrote
        # for that step: observation / reward / action / log-prob rows as always, and `env.reset_buf` / `env.progress_buf`
        # (fly.py:175-177) are re-pointed at row t of per-step [T, N] tensors -- so it is the default whenever one launch stays
        # short (T <= 4096 steps; the reference's 16-env shape, T = 40 960, steps launch by launch).  `persistent_rollout=False`
        # (or FLY_PERSISTENT_ROLLOUT=0) keeps one launch per step; the env's STATE tensors (root_tensor, dof_states, ...) show the
        # rollout's end while the host is still counting through it.
        want = getattr(args, "persistent_rollout", None)
        if want is None:
            want = os.environ.get("FLY_PERSISTENT_ROLLOUT", "1") != "0"
        self.persistent_rollout = bool(want) and T <= 4096 and not bool(getattr(args, "graph", False))
        self._reset_rows = self._progress_rows = None
        self._graphs = {}
        self._fwd_args = None
        self._score_acc = torch.zeros((), device=dev)
        # The reference prints its score line from a host read of device values (ppo.py:257-260).  A blocking read in the middle
        # of a rollout that runs as ONE launch stalls the host until the launch ends, and the device then idles while the host
        # walks the rest of the rollout's run() calls (measured: 150-270 us per iteration).  So log lines go through an ordered
        # queue: a score line is an asynchronous copy into pinned memory plus an event, and lines are written, in order, as soon
        # as the head of the queue is ready -- at the latest by flush_log() / exit() / the next blocking point.  The text and the
        # order of the lines are the reference's; only the moment they appear moves.  `async_log=False` reads and prints at once.
        want_async = getattr(args, "async_log", None)
        if want_async is None:
            want_async = os.environ.get("FLY_ASYNC_LOG", "1") != "0"
        self._async_log = bool(want_async) and dev.type == "cuda"
        self._log_q = collections.deque()
        # opt-in (`--log_throughput`): the score line also carries env-steps/s (all ranks) over the steps since the previous score
        # line, by the host's clock.  Off by default: stdout then is the reference's lines (ppo.py:257-260), character for character.
        self.log_throughput = bool(getattr(args, "log_throughput", False))
        self._rate_mark = None                                      # (perf_counter, run_step) of the previous score line
        self._pending_step = None                                   # deferred check of the device step counter (_update_hip)
        self.env.bind_obs(self._obs_ring[0])                        # first policy input: zeros (Q8)

        self.score = 0
        self.run_step = 0
        self.optim_step = 0

        self.net = Net(self.env.num_obs, self.env.num_act).to(dev)
        if getattr(self.args, "load", False):                       # ppo.py:147-149
            print("loaded from: ", str(self.args.load_path))
            self.net.load_state_dict(torch.load(self.args.load_path, map_location=dev, weights_only=True))
        self._loaded = bool(getattr(self.args, "load", False))
        from .policy import PackedPolicy
        self.policy = PackedPolicy(self.net, dev)                   # parameters become views of one packed buffer
        self.net._policy = self.policy
        # "hip": MFMA forward/backward + fused clip/Adam kernels; "torch": torch-ROCm autograd (A/B reference)
        self.update_backend = getattr(args, "update_backend", "hip")
        self.policy.init_training(self.mini_chunk_size * n, lr=self.lr)
        action_var = 0.01 if self.args.testing else 0.2             # ppo.py:152
        # ppo.py:236-237 decays the variance after every env step and ppo.py:233 adds the step's mean
        # reward to the score.  Both are applied to the device tensors LAZILY, for a run of steps at
        # once (`_flush_bookkeeping`): when the score is printed, before an update, and whenever
        # `action_var` is read; in between the policy launch derives the step's variance from the
        # tensor and the number of pending decays.  Bit for bit the per-step result.
        self._action_var = torch.full((self.env.num_act,), action_var, device=dev)
        self._book_from = 0            # first rollout row whose bookkeeping is still pending
        self._rows_done = 0            # rollout rows stepped so far in this rollout
        self.optim = torch.optim.Adam(self.net.parameters(), lr=self.lr)

        self._lib = _lib.load()
        self._gen = torch.Generator(device=dev)
        self._gen.manual_seed(int(getattr(args, "seed", 0)) + 1000003 * int(getattr(args, "rank", 0)))
        self.world_size = int(getattr(args, "world_size", 1))
        # "grad_allreduce" (default): one packed-gradient all-reduce per optimizer step = the reference's
        # update on the global minibatch.  "param_average": one exchange per PPO update (non-parity).
        self.dp_mode = getattr(args, "dp_mode", "grad_allreduce")
        if self.dp_mode not in ("grad_allreduce", "param_average"):
            raise ValueError("dp_mode must be grad_allreduce or param_average")
        # the per-step gradient exchange: "rccl" = torch.distributed.all_reduce (RCCL over xGMI; the default and the
        # fallback) or "p2p" = the one-shot peer-to-peer kernel of csrc/dp_p2p.hip (one node, <= 16 ranks)
        # "auto" = p2p if its start-up self-test against the collective passes on this node, else rccl.
        self.dp_allreduce = getattr(args, "dp_allreduce", None) or os.environ.get("FLY_DP_ALLREDUCE", "rccl")
        if self.dp_allreduce not in ("rccl", "p2p", "auto"):
            raise ValueError("dp_allreduce must be rccl, p2p or auto")
        self._p2p = None
        self._flat_grad = None
        if self.world_size > 1:
            from .dist import FlatGradAllReduce
            self._flat_grad = FlatGradAllReduce(self.net.parameters(), self.world_size)

    # ppo.py:230 replaces the whole [T,N,1] buffer by the LAST step's [N,1] mask after every step
    # (Q1).  reset_buf only changes inside env.step, so deriving the mask on demand is the same
    # thing without two tiny launches per step; an explicit assignment (tests) overrides it.
    @property
    def all_done(self):
        if self._all_done is not None:
            return self._all_done
        return (1 - self.env.reset_buf).unsqueeze(-1)

    @all_done.setter
    def all_done(self, value):
        self._all_done = value

    # ------------------------------------------------------------------------------------------
    @property
    def action_var(self):
        if getattr(self, "_book_terms", None) is not None:
            self._flush_bookkeeping()
        return self._action_var

    @action_var.setter
    def action_var(self, value):
        """The reference rebinds this attribute (ppo.py:237); here the rollout launches hold a POINTER to
        the variance tensor, so an assignment first applies the decays still pending on the old value and
        then writes the new one in place -- every later launch (sampling, log-prob, loss) sees it."""
        if getattr(self, "_action_var", None) is None:
            self._action_var = value
            return
        if getattr(self, "persistent_rollout", False) and 0 < getattr(self, "_rows_done", 0) < self.rollout_size:
            # one launch per rollout: the steps of the rollout in progress have already been sampled on the device
            import warnings
            warnings.warn("action_var assigned inside a rollout that was launched as ONE kernel: the new value takes effect at the "
                          "next rollout (PPO(..., persistent_rollout=False) launches step by step and applies it at once)")
        if getattr(self, "_book_terms", None) is not None:
            self._flush_bookkeeping()
        with torch.no_grad():
            v = torch.as_tensor(value, dtype=torch.float32, device=self._action_var.device)
            self._action_var.copy_(v.expand_as(self._action_var))

    def _flush_bookkeeping(self):
        """Apply the score terms and variance decays of rollout rows [_book_from, _rows_done)."""
        rows = self._rows_done - self._book_from
        if rows <= 0:
            return
        n = int(self.args.num_envs)
        P = C.c_void_p
        _lib.check(self._lib.ppo_rollout_bookkeeping(
            P(self.all_reward[self._book_from].data_ptr()), C.c_int64(rows), C.c_int64(n), P(self._book_terms.data_ptr()),
            P(self._score_acc.data_ptr()), C.c_float(1.0 / self.num_eval_freq), P(self._action_var.data_ptr()),
            C.c_int(self.num_acts), C.c_float(self._var_decay), self._var_min, P(self._rows_applied.data_ptr()),
            _lib.stream_ptr()), "ppo_rollout_bookkeeping")
        self._book_from = self._rows_done

    def make_data(self):
        """ppo.py:157-171: TD target and GAE.  `all_done` is the [N,1] mask of the LAST step,
        broadcast over T (Q1), and the recurrence never resets at episode ends (Q2)."""
        T, n = self.rollout_size, self.args.num_envs
        with torch.no_grad():
            if self.reuse_rollout_values and self._v_have == T and self._v_version == self.policy.version:
                self._v_ring[T].copy_(self.net.v(self._obs_ring[T]))
                values = self._v_ring                               # rows 0..T-1 were written by the rollout launches
            else:
                values = self.net.v(self._obs_ring)                 # [T+1, N, 1]: v(obs) and v(next_obs) in one pass
            self._v_have = 0
            done = self.all_done
            per_step = done.dim() == 3 and done.shape[0] == T and done.shape[1] == n
            mode = 1 if per_step else 0                             # reference path: [N,1]
            if n < 512 and T >= 1024:
                mode |= 4                                           # PPO_GAE_SCAN: few envs x long rollout
            done_f = done.to(torch.float32).contiguous()
            _lib.check(self._lib.ppo_td_gae(
                C.c_void_p(self.all_reward.data_ptr()), C.c_void_p(values[:T].data_ptr()),
                C.c_void_p(values[1:].data_ptr()), C.c_void_p(done_f.data_ptr()),
                C.c_float(self.gamma), C.c_float(self.lmbda), C.c_int64(T), C.c_int64(n),
                C.c_void_p(self._target.data_ptr()), C.c_void_p(self.all_advantage.data_ptr()),
                C.c_int(mode), _lib.stream_ptr()), "ppo_td_gae")
            self._keep = (values, done_f)                           # alive until the stream has consumed them
            if self.normalize_advantage:
                self._normalize_advantage()
        return self.all_obs, self.all_acts, self.all_log_prob, self._target, self.all_advantage

    def _normalize_advantage(self):
        """Opt-in (`normalize_advantage`): adv <- (adv - mean) / (std + 1e-8) over the whole rollout of
        ALL ranks.  Not in the reference (ppo.py:171 uses raw advantages); named by BASELINE's north_star."""
        import torch.distributed as dist
        cnt = self.all_advantage.numel()
        p = lambda x: C.c_void_p(x.data_ptr())   # noqa: E731
        _lib.check(self._lib.ppo_adv_stats(p(self.all_advantage), C.c_int64(cnt), p(self._adv_stats), _lib.stream_ptr()),
                   "ppo_adv_stats")
        if self.world_size > 1:
            dist.all_reduce(self._adv_stats[:2], op=dist.ReduceOp.SUM)
        _lib.check(self._lib.ppo_adv_apply(p(self.all_advantage), C.c_int64(cnt), p(self._adv_stats),
                                           C.c_float(float(cnt * self.world_size)), C.c_float(1e-8), _lib.stream_ptr()),
                   "ppo_adv_apply")

    def minibatch_loss(self, obs_mc, action_mc, old_log_prob_mc, target_mc, advantage_mc):
        """ppo.py:184-194.  Old log-prob is of the unclipped sample, the new one of the stored
        clipped action under the current (decayed) variance (Q6); the Huber term is a scalar
        mean added to every element (Q7)."""
        mu = self.net.pi(obs_mc)
        log_prob = diag_gauss_logprob(mu, action_mc, self._action_var)
        ratio = torch.exp(log_prob - old_log_prob_mc).unsqueeze(-1)
        surr1 = ratio * advantage_mc
        surr2 = torch.clamp(ratio, 1 - self.clip, 1 + self.clip) * advantage_mc
        loss = -torch.min(surr1, surr2) + F.smooth_l1_loss(self.net.v(obs_mc), target_mc)
        return loss.mean()

    def update(self):
        """ppo.py:173-202: 5 epochs x 15 contiguous-in-T minibatches; the 16th chunk is never
        visited (Q3).  With world_size > 1 the flat gradient is all-reduced (mean) before the clip."""
        if getattr(self, "_book_terms", None) is not None:
            self._flush_bookkeeping()                               # the loss uses the variance after this rollout's decays
        obs, action, old_log_prob, target, advantage = self.make_data()
        if self.update_backend == "hip":
            return self._update_hip(obs, action, old_log_prob, target, advantage)
        for _ in range(self.epoch):
            k = 0
            for j in range(self.mini_chunk_size, self.rollout_size, self.mini_chunk_size):
                loss = self.minibatch_loss(obs[k:j], action[k:j], old_log_prob[k:j], target[k:j], advantage[k:j])
                if self._flat_grad is not None:
                    self._flat_grad.zero()                          # grads are views of the flat buffer
                else:
                    self.optim.zero_grad()
                loss.backward()
                if self._flat_grad is not None:
                    self._flat_grad.allreduce_mean()
                nn.utils.clip_grad_norm_(self.net.parameters(), 1.0)
                self.optim.step()
                self.optim_step += 1
                k = j
        self.policy.refresh()       # torch wrote the master weights: rebuild the fragment-ordered copies

    def _update_hip(self, obs, action, old_log_prob, target, advantage):
        """ppo.py:179-202 on the MFMA kernels: per minibatch one fused forward, the loss gradient +
        dX chain, the split-row dW, (world_size > 1: ONE all-reduce of the packed gradient), and
        the fused clip + Adam step.  Minibatches are contiguous slices of the rollout, so no
        gather/copy happens."""
        import torch.distributed as dist
        mc, n = self.mini_chunk_size, int(self.args.num_envs)
        rows = mc * n
        pol = self.policy
        sync_grads = self.world_size > 1 and self.dp_mode == "grad_allreduce"
        slices = [(j - mc, j) for _ in range(self.epoch) for j in range(mc, self.rollout_size, mc)]   # 5 x 15 (Q3)
        self.prepare()                   # idempotent; callers that time the update call it themselves before warm-up

        def run(todo):
            for k, j in todo:
                pol.minibatch_grad(obs[k:j].view(rows, self.num_obs), action[k:j].view(rows, self.num_acts),
                                   old_log_prob[k:j].view(rows), advantage[k:j].view(rows),
                                   target[k:j].view(rows), self._action_var, self.clip,
                                   fuse_norm=not sync_grads)
                if sync_grads:
                    if self._p2p is not None:
                        self._p2p.allreduce_(pol.G)                 # one launch, one xGMI hop, sum in rank order
                    else:
                        dist.all_reduce(pol.G, op=dist.ReduceOp.SUM)    # 297 KB, latency-bound on xGMI
                # either way ONE optimizer launch per step; it reads the exchange's err word itself and refuses a gradient
                # that ANY workgroup of the exchange left un-reduced (fail closed, on the device)
                pol.adam_step(grad_scale=1.0 / self.world_size if sync_grads else 1.0, norm_ready=not sync_grads,
                              self_norm=sync_grads, grad_invalid=self._p2p.err if (sync_grads and self._p2p is not None) else None)

        self._check_step_counter()       # the previous update's counter, copied while this rollout ran
        if pol.h2_live() and pol.fused_step and not pol.h2_calibrated:
            # fp16x2 step: the per-class scales are measured on the first minibatch before anything depends on them (a new
            # network, or an update whose values outgrew the scales: a few discarded launches, host-synchronising, rare)
            k, j = slices[0]
            pol.calibrate_h2(obs[k:j].view(rows, self.num_obs), action[k:j].view(rows, self.num_acts), old_log_prob[k:j].view(rows),
                             advantage[k:j].view(rows), target[k:j].view(rows), self._action_var, self.clip)
        run(slices)
        if self._p2p is not None and not self._p2p.check():
            # FIRST, before anything looks at the step counter: a bounded wait of the peer-to-peer exchange expired on this
            # rank.  Its optimizer launches have refused the un-reduced gradients (fail closed) and the counter is behind, but
            # redoing steps is no remedy here -- no peer is at those epochs.  Fatal by design (dist.py: "the caller must stop").
            raise _lib.FlyHipError("dp_allreduce_p2p: a rank never published its gradient (bounded wait expired; "
                                   "FLY_P2P_POLL_LOG2 raises the budget, --dp_allreduce rccl avoids the kernel)")
        if not pol.update_can_be_refused():
            # No launch of this update path can leave an invalid gradient (the fused optimizer step hands nothing from workgroup to
            # workgroup), so the device counter only CONFIRMS the count.
            if self._p2p is None and self._async_log:
                # copied asynchronously and compared when the next update begins (or at exit) -- the host goes straight on to the
                # next rollout instead of draining the queue here (measured: ~120 us of launch-bound idle per iteration behind a
                # blocking read)
                host = torch.empty(1, dtype=pol.step.dtype, pin_memory=True)
                host.copy_(pol.step, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._pending_step = (ev, host, pol.steps_issued)
            else:
                got = int(pol.step.item())
                if got != pol.steps_issued:
                    raise _lib.FlyHipError("update: the device step counter says %d optimizer steps, %d were issued"
                                           % (got, pol.steps_issued))
                self._drain_log(block=True)
            self.optim_step += len(slices)
            self._finish_update(sync_grads)
            return
        # ONE host sync per update.  A fused forward+backward launch whose backward could not get a tile
        # leaves an invalid gradient; the optimizer kernels refuse such a step ON THE DEVICE (on every
        # rank: the flag rides inside the all-reduced gradient), and every later step of this update too.
        # So the device step counter says how many minibatches really happened: redo the rest through the
        # two-launch path -- bit for bit what an undisturbed update leaves.  (Only this path -- the tile
        # hand-off of mlp_forward_backward -- is ever redone.)
        h2 = pol.h2_live() and pol.fused_step
        for attempt in range(3):
            short = pol.steps_issued - int(pol.step.item())
            if (pol.fuse_fwd_bwd and not h2) or short:
                pol.check_fused_launch()
            if short == 0:
                break
            if short < 0 or short > len(slices) or attempt == 2:
                raise _lib.FlyHipError("update: device step counter is %d steps behind the %d issued" % (short, pol.steps_issued))
            pol.steps_issued -= short
            if h2 and not pol.h2_suspended:
                # a value of some launch did not fit fp16 under the scales its predecessor left (mlp_fused_h2.inc): that step and
                # every later one were refused on the device (the sticky word), on every rank.  Redo them, in order, on the bf16x3
                # kernel -- bit for bit what an undisturbed bf16x3 step leaves -- and measure the scales afresh before the next update.
                pol.h2_overflow.zero_()
                pol.h2_suspended = True
                pol.h2_calibrated = False
                pol.h2_overflows += 1
            else:
                pol.fuse_fwd_bwd = False
                print("mlp_forward_backward: %d of %d optimizer steps were refused on the device (a backward workgroup "
                      "could not get its tile); redoing them with two launches" % (short, len(slices)))
            run(slices[len(slices) - short:])
            if self._p2p is not None and not self._p2p.check():
                raise _lib.FlyHipError("dp_allreduce_p2p: a rank never published its gradient while refused steps were redone")
        pol.h2_suspended = False
        self.optim_step += len(slices)
        self._drain_log(block=True)                     # the queue is drained anyway: pending log lines cost nothing here
        self._finish_update(sync_grads)

    def _finish_update(self, sync_grads):
        import torch.distributed as dist
        pol = self.policy
        if self.world_size > 1 and not sync_grads:
            # dp_mode "param_average": ONE exchange per PPO update (BASELINE's north_star wording) --
            # ranks take their 75 optimizer steps locally, then parameters and Adam moments are
            # averaged.  NOT the reference algorithm on a larger batch (that is the default mode).
            for buf in (pol.P, pol.exp_avg, pol.exp_avg_sq):
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
                buf.div_(self.world_size)
            pol.refresh()

    def prepare(self):
        """Everything an update needs that is not part of an update: with data-parallel ranks and `dp_allreduce` "p2p" or
        "auto", open the peer windows and self-test the one-shot exchange against the collective.  A COLLECTIVE call (every
        rank makes it, in the same place); idempotent.  `bench.py` calls it before the warm-up so that none of it lands in a
        timed region; `_update_hip` calls it too, so a plain `run()` loop needs nothing extra."""
        if getattr(self, "_prepared", False):
            return
        self._prepared = True
        self.p2p_selftest = None
        sync_grads = self.world_size > 1 and self.dp_mode == "grad_allreduce"
        if sync_grads and self.update_backend == "hip" and self.dp_allreduce in ("p2p", "auto"):
            self._p2p = self._open_p2p(self.policy.G.numel())

    def _open_p2p(self, n):
        """Open the peer windows and hold the one-shot kernel to the collective on random data (3 epochs, both
        parities).  Every rank takes the same decision at every stage (`P2PAllReduce` votes inside its constructor, the
        self-test verdicts are all-reduced): on any failure -- no IPC between these devices, a wrong sum, a rank that
        never publishes -- "auto" falls back to RCCL on every rank together, "p2p" raises on every rank together."""
        import torch.distributed as dist
        from .dist import P2PAllReduce, P2PUnavailable
        from .policy import ERR_SLOT
        p2p, why = None, ""
        try:
            p2p = P2PAllReduce(n, self.device, fail_slot=ERR_SLOT)          # raises on ALL ranks or on none
        except P2PUnavailable as e:
            why = str(e)[:300]
        good = p2p is not None
        if good:
            flag = torch.ones(1, device=self.device, dtype=torch.int32)
            gen = torch.Generator(device=self.device)
            gen.manual_seed(7 + int(getattr(self.args, "rank", 0)))
            for _ in range(3):
                a = torch.randn(n, device=self.device, generator=gen)
                a[ERR_SLOT] = 0.0
                b = a.clone()
                p2p.allreduce_(a)
                dist.all_reduce(b, op=dist.ReduceOp.SUM)
                mine = p2p.check() and torch.allclose(a, b, rtol=1e-5, atol=1e-5)
                flag.fill_(1 if mine else 0)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) != 1:
                    good, why = False, "self-test against the collective failed"
                    break
        self.p2p_selftest = "passed" if good else ("failed: " + (why or "a peer failed"))
        if good:
            self.dp_allreduce = "p2p"
            return p2p
        if p2p is not None:
            p2p.close()
        if self.dp_allreduce == "p2p":
            raise _lib.FlyHipError("dp_allreduce=p2p is not usable on this node: %s" % (why or "a peer failed"))
        if int(getattr(self.args, "rank", 0)) == 0:
            print("dp_allreduce auto: peer-to-peer all-reduce unavailable (%s); using the RCCL collective" % (why or "a peer failed"))
        self.dp_allreduce = "rccl"
        return None

    # ------------------------------------------------------------------------------------------
    def _prepare_step_args(self):
        """Every pointer of step t is a fixed row of a preallocated rollout tensor, so the ctypes
        argument tuples of the three launches are built ONCE per t: a step then costs three foreign
        calls and two attribute stores on the host (the rollout was host-bound at ~50 us/step
        against ~35 us of GPU work)."""
        T, n = self.rollout_size, int(self.args.num_envs)
        P = C.c_void_p
        pol = self.policy
        self._obs_rows = [self._obs_ring[t] for t in range(T + 1)]
        self._reward_rows = [self.all_reward[t].view(-1) for t in range(T)]
        self._act_rows = [self.all_acts[t] for t in range(T)]
        fwd, bufs, step = [], [], []
        self._var_decay = 0.0 if self.args.testing else 0.00001     # ppo.py:236
        self._var_min = C.c_float(0.01)
        self._book_terms = torch.zeros(T, device=self.device)       # scratch of ppo_rollout_bookkeeping
        # The policy launch of row t derives its variance from the tensor and the decays still pending:
        # t - (rows of this rollout already applied).  The second term lives in a DEVICE word that
        # ppo_rollout_bookkeeping advances, so the launch arguments of row t never change -- the same tuple
        # serves every rollout, eager or replayed from a captured hipGraph.
        self._rows_applied = torch.zeros(1, dtype=torch.int32, device=self.device)
        base_ptr = P(self._rows_applied.data_ptr())
        var_ptr = P(self._action_var.data_ptr())
        for t in range(T):
            fwd.append((P(pol.P.data_ptr()), P(pol.PF.data_ptr()), P(self._obs_rows[t].data_ptr()), C.c_int64(n),
                        P(self._eps_all[t].data_ptr()), var_ptr, C.c_int(t), C.c_float(self._var_decay), self._var_min,
                        P(self._act_rows[t].data_ptr()), P(self.all_log_prob[t].data_ptr()), None,
                        P(self._v_ring[t].data_ptr()), pol.infer_pb_ptr(), base_ptr))
            step.append((P(pol.P.data_ptr()), P(pol.PF.data_ptr()), P(self._obs_rows[t].data_ptr()),
                         P(self._eps_all[t].data_ptr()), var_ptr, C.c_int(t), C.c_float(self._var_decay), self._var_min,
                         P(self._act_rows[t].data_ptr()), P(self.all_log_prob[t].data_ptr()), P(self._v_ring[t].data_ptr()),
                         pol.infer_pb_ptr(), base_ptr))
            bufs.append((self._obs_rows[t + 1].data_ptr(), self._reward_rows[t].data_ptr()))
        self._fwd_args, self._buf_ptrs, self._step_args = fwd, bufs, step
        self._graphs = {}                                            # captured steps hold the old pointers
        # one launch per env step (policy + sampling + env step), eager or captured; FLY_FUSE_ROLLOUT_STEP=0 keeps the
        # two-launch form (mlp_forward_sample + fly_step) for the A/B test
        self.fuse_rollout_step = os.environ.get("FLY_FUSE_ROLLOUT_STEP", "1") != "0"
        self._args_infer_gemm = pol.gemm_infer

    def _launch_step(self, t):
        """The device work of one env step (ppo.py:213-237): ONE launch (`ppo_rollout_step`), no host logic.  Rows of
        the rollout are written in place (obs row t+1, action/log-prob/reward rows t); the score and
        variance bookkeeping of the step is deferred (`_flush_bookkeeping`)."""
        lib, env = self._lib, self.env
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if t == 0:
            self._eps_all.normal_(generator=self._gen)              # the eps of MultivariateNormal.sample, whole rollout
            self._rows_applied.zero_()                              # a new rollout: no row's bookkeeping applied yet
        env.obs_buf, env.reward_buf = self._obs_rows[t + 1], self._reward_rows[t]      # ppo.py:228-229
        env._bufs.obs, env._bufs.reward = self._buf_ptrs[t]
        if self.fuse_rollout_step:
            # policy + sampling + env step of every 32-env tile in ONE launch (ppo.py:214-229)
            rc = lib.ppo_rollout_step(env._handle, C.byref(env._bufs), *self._step_args[t], st)
        else:
            rc = lib.mlp_forward_sample(*self._fwd_args[t], st)     # ppo.py:214-220, :227 (policy + sampling fused)
            rc |= lib.fly_step(env._handle, C.c_void_p(self._act_rows[t].data_ptr()), C.byref(env._bufs), st)  # ppo.py:223
        if rc:
            _lib.check(rc, "rollout step")
        self._after_step(t)
        env.render_count += 1

    def _launch_rollout(self):
        """The device work of a WHOLE rollout in one launch (`ppo_rollout_all`): eps draw, then every workgroup runs
        the T steps of its own 32 envs with the env state in registers.  Bit for bit what T `_launch_step` calls leave."""
        P = C.c_void_p
        pol, env, T = self.policy, self.env, self.rollout_size
        self._eps_all.normal_(generator=self._gen)
        self._rows_applied.zero_()
        if self._reset_rows is None:
            n = int(self.args.num_envs)
            self._reset_rows = torch.zeros((T, n), dtype=torch.long, device=self.device)
            self._progress_rows = torch.zeros((T, n), dtype=torch.long, device=self.device)
        # env._bufs.reset / .progress point at the CURRENT flags (the env's own tensors, or the last row of the previous
        # rollout): the launch reads them once, then writes step t's flags to row t
        _lib.check(self._lib.ppo_rollout_all(
            env._handle, C.byref(env._bufs), P(pol.P.data_ptr()), P(pol.PF.data_ptr()), P(self._obs_ring.data_ptr()),
            P(self._eps_all.data_ptr()), P(self._action_var.data_ptr()), C.c_float(self._var_decay), self._var_min,
            P(self.all_acts.data_ptr()), P(self.all_log_prob.data_ptr()), P(self._v_ring.data_ptr()),
            P(self.all_reward.data_ptr()), C.c_int(T), P(self._rows_applied.data_ptr()), pol.infer_pb_ptr(),
            P(self._reset_rows.data_ptr()), P(self._progress_rows.data_ptr()), _lib.stream_ptr()), "ppo_rollout_all")

    def _after_step(self, t):
        """Host-side state of a step that has been issued (launched or replayed): rows whose score / variance
        bookkeeping is pending (ppo.py:233, :236-237, applied by _flush_bookkeeping) and how many rows of
        v(obs_t) the rollout's policy launches have left for make_data."""
        if t == 0:
            self._book_from = 0
            self._v_have, self._v_version = 1, self.policy.version
        elif self._v_have == t:
            self._v_have = t + 1
        self._rows_done = t + 1

    SCORE_LINE = 'Steps: {:04d} | Opt Step: {:04d} | Reward {:.04f} | Action Var {:.04f}'

    def _emit(self, text):
        """A log line, behind whatever is still pending (see `_async_log`)."""
        self._log_q.append(text)
        self._drain_log()

    def _throughput_suffix(self):
        """' | Env-steps/s ...' for the score line when `log_throughput` is on: env steps of all ranks since the previous score
        line over the host's wall clock (inside a rollout that runs as ONE launch the host counts ahead of the device, so a
        window is exact only from rollout boundary to rollout boundary; over several windows it is the loop's rate)."""
        if not self.log_throughput:
            return ""
        import time
        now, mark = time.perf_counter(), self._rate_mark
        self._rate_mark = (now, self.run_step)
        if mark is None or now <= mark[0]:
            return " | Env-steps/s n/a"
        steps = (self.run_step - mark[1]) * int(self.args.num_envs) * self.world_size
        return " | Env-steps/s {:.4g}".format(steps / (now - mark[0]))

    def _emit_score(self):
        """ppo.py:257-260: the score line of this step.  Values are read from the device asynchronously (pinned memory + event);
        the accumulator is cleared on the stream, behind the copy."""
        rank0 = int(getattr(self.args, "rank", 0)) == 0
        suffix = self._throughput_suffix()
        if not self._async_log:
            self._drain_log(block=True)
            score = float(self._score_acc.item())
            self._score_acc.zero_()
            if rank0:
                print(self.SCORE_LINE.format(self.run_step, self.optim_step, score, self._action_var[0].item()) + suffix)
            return
        dev_vals = torch.stack((self._score_acc.reshape(()), self._action_var[0]))
        host = torch.empty(2, dtype=torch.float32, pin_memory=True)
        host.copy_(dev_vals, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._score_acc.zero_()
        self._log_q.append((ev, host, dev_vals, self.run_step, self.optim_step, rank0, suffix))
        self._drain_log()

    def _drain_log(self, block=False):
        """Write the queued lines whose values have arrived, in order; block=True waits for all of them."""
        q = self._log_q
        while q:
            head = q[0]
            if isinstance(head, str):
                print(head)
            else:
                ev, host, _keep, run_step, optim_step, rank0, suffix = head
                if block:
                    ev.synchronize()
                elif not ev.query():
                    return
                if rank0:
                    print(self.SCORE_LINE.format(run_step, optim_step, float(host[0]), float(host[1])) + suffix)
            q.popleft()

    def flush_log(self):
        """Blocks until every queued log line has been written (a host sync when one is pending)."""
        self._drain_log(block=True)

    def _check_step_counter(self):
        """The deferred half of _update_hip's step check: the device counter copied at the end of the previous update."""
        if self._pending_step is None:
            return
        ev, host, expect = self._pending_step
        self._pending_step = None
        ev.synchronize()
        if int(host[0]) != expect:
            raise _lib.FlyHipError("update: the device step counter says %d optimizer steps, %d were issued"
                                   % (int(host[0]), expect))

    def run(self):
        """ppo.py:204-264: one env step of the rollout (and an update when the rollout is full).
        `graph=True` replays the device work of a whole rollout from ONE captured hipGraph (the first
        rollout runs eagerly, the second captures)."""
        t = self.mini_batch_number
        end = self.env.end
        if self._fwd_args is None or self._args_infer_gemm != self.policy.gemm_infer:
            self._prepare_step_args()                               # (re)built when the inference arithmetic changes
        with torch.no_grad():
            if self.persistent_rollout:
                if t == 0:
                    self._launch_rollout()
                env = self.env
                env.obs_buf, env.reward_buf = self._obs_rows[t + 1], self._reward_rows[t]
                env._bufs.obs, env._bufs.reward = self._buf_ptrs[t]
                env.reset_buf, env.progress_buf = self._reset_rows[t], self._progress_rows[t]      # this step's flags (fly.py:175-177)
                env._bufs.reset, env._bufs.progress = env.reset_buf.data_ptr(), env.progress_buf.data_ptr()
                self._after_step(t)
                env.render_count += 1
            elif not self.use_graph or self.run_step < self.rollout_size:
                self._launch_step(t)
            else:
                # graph=True: the device work of a WHOLE rollout (eps draw + T one-launch env steps) is one
                # captured hipGraph, replayed when the rollout's first step is asked for; the later run() calls of
                # the rollout only advance the host-side counters.  (A graph per step cannot win: the step is
                # GPU-bound and every replay pays a graph launch of its own -- tools/graph_vs_eager.py.)  The
                # deferred bookkeeping makes this exact: row t's launch carries the frozen row index, the
                # device word `_rows_applied` is zeroed inside the graph, and a flush only ever covers the rows
                # the HOST has counted as done.
                if t == 0:
                    key = bool(self.args.testing)
                    g = self._graphs.get(key)
                    if g is None:
                        g = torch.cuda.CUDAGraph()
                        g.register_generator_state(self._gen)
                        torch.cuda.synchronize(self.device)
                        with torch.cuda.graph(g):
                            for tt in range(self.rollout_size):
                                self._launch_step(tt)
                        self._graphs[key] = g
                    g.replay()
                self.env.obs_buf, self.env.reward_buf = self._obs_rows[t + 1], self._reward_rows[t]
                self.env._bufs.obs, self.env._bufs.reward = self._buf_ptrs[t]
                self._after_step(t)
                self.env.render_count += 1

        if t + 1 == self.rollout_size:                              # ppo.py:240-252
            self._flush_bookkeeping()                               # the update reads the decayed variance
            if not self.args.testing:
                self._emit("Training")
                self.update()
            self.mini_batch_number = 0
            with torch.no_grad():
                self._obs_ring[0].copy_(self._obs_ring[self.rollout_size])
            self.env.bind_obs(self._obs_ring[0])
            if getattr(self.args, "save", False) and self.optim_step % self.args.save_freq == 0 and self.optim_step != 0:
                self._emit("saving...")
                self.save(str(self.optim_step))
                self._emit("saved!")
        else:
            self.mini_batch_number += 1

        if self.run_step % self.num_eval_freq == 0:                 # ppo.py:257-260
            self._flush_bookkeeping()
            self._emit_score()
            self.score = 0
        elif self._log_q:
            self._drain_log()

        self.run_step += 1
        return end

    def save(self, endofname=""):
        """ppo.py:266-273: state_dict only, reference key names."""
        if not getattr(self.args, "save", False):
            return
        self._check_step_counter()
        if int(getattr(self.args, "rank", 0)) != 0:
            return
        path = self.args.save_path + endofname + ".pth"
        # parameters are views of the packed buffer: save compact, contiguous copies
        torch.save({k: v.detach().clone().contiguous() for k, v in self.net.state_dict().items()}, path)

    def generate_video(self):
        self.env.generate_video()

    def exit(self):
        self._check_step_counter()
        self.flush_log()
        if self._p2p is not None:
            self._p2p.close()
            self._p2p = None
        self.env.exit()
