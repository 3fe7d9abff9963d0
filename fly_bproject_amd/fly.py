"""`Fly`: the reference's vectorised environment class (fly.py:11-681), MI355X-native.

Same constructor argument (`args` with `.sim_device`, `.num_envs`, `.headless`, ...), same public
attributes (`num_obs`, `num_act`, `obs_buf`, `reward_buf`, `reset_buf`, `progress_buf`, `end`,
`root_tensor`, `dof_states`, `dof_pos`, `dof_vel`, `force_tensor`, `potentials`, ...) and methods
(`step`, `reset`, `get_obs`, `get_reward`, `simulate`, `render`, `generate_video`, `exit`).  What
was a chain of torch ops plus Isaac Gym calls per step is ONE kernel launch through the C ABI
(`fly_step`, include/flyhip.h); the unfused methods launch the matching single-phase kernels.

No Isaac Gym, no PhysX, no CPU path: the rigid-body model behind `simulate()` is the
build-defined FlyDyn because the reference's physics is a closed binary.
"""
import ctypes as C

import torch

from . import _lib
from .params import NUM_CONTACT, NUM_DOF, NUM_OBS, ROOT_DIM, default_params


class Fly:
    def __init__(self, args, params=None):
        self.args = args
        self.end = False                                  # fly.py:15
        self.up_axis_idx = 2
        n = int(args.num_envs)
        variant = getattr(args, "variant", "bigGrav")
        reward = getattr(args, "reward", "standing")
        self.params = params if params is not None else default_params(n, variant, reward)
        if self.params.num_envs != n:
            raise ValueError("params.num_envs (%d) != args.num_envs (%d)" % (self.params.num_envs, n))
        self.dt = float(self.params.dt)                   # fly.py:16
        self.num_act = NUM_DOF                            # fly.py:31
        self.num_obs = 19 + 3 * self.num_act              # fly.py:32
        self.max_episode_length = int(self.params.max_episode_length)
        self.render_count = 0

        self.device = torch.device(args.sim_device)
        if self.device.type != "cuda":
            raise _lib.FlyHipError(
                "Fly runs on an MI355X (sim_device=%r). This build has no CPU path; the CPU restatement "
                "under oracle/ is test infrastructure only." % (args.sim_device,))
        if not torch.cuda.is_available():
            raise _lib.FlyHipError("no GPU visible: Fly needs an MI355X")
        self._lib = _lib.load()
        torch.cuda.set_device(self.device)
        self._handle = C.c_void_p()
        _lib.check(self._lib.fly_create(C.byref(self.params), C.byref(self._handle)), "fly_create")

        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        # fly.py:169-179
        self.obs_buf = torch.zeros((n, NUM_OBS), **f32)
        self.reward_buf = torch.zeros(n, **f32)
        self.reset_buf = torch.ones(n, device=dev, dtype=torch.long)
        self.progress_buf = torch.zeros(n, device=dev, dtype=torch.long)
        # fly.py:378-395, :89-100 (shapes and aliasing views as Isaac Gym exposed them)
        self.root_tensor = torch.zeros((n, ROOT_DIM), **f32)
        self.dof_states = torch.zeros((n, NUM_DOF * 2), **f32)
        self.force_tensor = torch.zeros((n * NUM_CONTACT, 3), **f32)
        self.num_dof = NUM_DOF
        self.dof_pos = self.dof_states.view(n, NUM_DOF, 2)[..., 0]
        self.dof_vel = self.dof_states.view(n, NUM_DOF, 2)[..., 1]
        self.root_positions = self.root_tensor[:, 0:3]
        self.root_orientations = self.root_tensor[:, 3:7]
        self.root_linvels = self.root_tensor[:, 7:10]
        self.root_angvels = self.root_tensor[:, 10:13]
        # fly.py:636: the scaled actions, [N*18, 1] (allocated up front; upstream creates it in the first step)
        self.actions = torch.zeros((n * NUM_DOF, 1), **f32)
        # fly.py:121-135
        self.potentials = torch.full((n,), -1000.0 / self.dt, **f32)
        self.prev_potentials = self.potentials.clone()
        self.targets = torch.tensor(list(self.params.target), **f32).repeat((n, 1))
        self.dof_limits_lower = torch.tensor(list(self.params.dof_lo), **f32)
        self.dof_limits_upper = torch.tensor(list(self.params.dof_hi), **f32)
        base = torch.arange(n, device=dev, dtype=torch.long).view(n, 1) * NUM_CONTACT
        self.index_abdomen_sim = (base + torch.arange(0, 5, device=dev)).reshape(-1)     # fly.py:311-314
        self.index_legs_tip = (base + torch.arange(5, 11, device=dev)).reshape(-1)
        # episode statistics kept by the step kernel (not in the reference): running return/length of
        # the current episode and per-env totals over finished ones
        self.episode_return_buf = torch.zeros(n, **f32)
        self.episode_length_buf = torch.zeros(n, **f32)
        self.finished_return_sum = torch.zeros(n, **f32)
        self.finished_length_sum = torch.zeros(n, **f32)
        self.finished_count = torch.zeros(n, **f32)
        self._bufs = _lib.FlyBuffers()
        self._refresh_pointers()

    # ------------------------------------------------------------------------------------------
    def _refresh_pointers(self):
        b = self._bufs
        b.root, b.dof_state = self.root_tensor.data_ptr(), self.dof_states.data_ptr()
        b.targets, b.contact = self.actions.data_ptr(), self.force_tensor.data_ptr()
        b.pot, b.prev_pot = self.potentials.data_ptr(), self.prev_potentials.data_ptr()
        b.obs, b.reward = self.obs_buf.data_ptr(), self.reward_buf.data_ptr()
        b.reset, b.progress = self.reset_buf.data_ptr(), self.progress_buf.data_ptr()
        b.ep_return, b.ep_length = self.episode_return_buf.data_ptr(), self.episode_length_buf.data_ptr()
        b.done_return, b.done_length = self.finished_return_sum.data_ptr(), self.finished_length_sum.data_ptr()
        b.done_count = self.finished_count.data_ptr()

    def bind_obs(self, rows):
        """Make `rows` (f32 [N,73], contiguous) the observation buffer: the
        step kernel then writes observation rows straight into the caller's rollout storage."""
        if rows.shape != self.obs_buf.shape or rows.dtype != torch.float32 or not rows.is_contiguous():
            raise ValueError("bind_obs needs a contiguous f32 [%d,%d] tensor" % tuple(self.obs_buf.shape))
        if rows.device != self.obs_buf.device:
            raise ValueError("bind_obs: rows live on %s, env on %s" % (rows.device, self.obs_buf.device))
        self.obs_buf = rows
        self._bufs.obs = rows.data_ptr()

    def bind_reward(self, row):
        """Make `row` (f32 [N], contiguous) the reward buffer (a row of the caller's rollout)."""
        if row.numel() != self.reward_buf.numel() or row.dtype != torch.float32 or not row.is_contiguous():
            raise ValueError("bind_reward needs a contiguous f32 tensor of %d elements" % self.reward_buf.numel())
        if row.device != self.reward_buf.device:
            raise ValueError("bind_reward: wrong device")
        self.reward_buf = row.view(-1)
        self._bufs.reward = row.data_ptr()

    def _check_actions(self, actions):
        n = self.args.num_envs
        if actions.shape != (n, NUM_DOF) or actions.dtype != torch.float32 or actions.device != self.device:
            raise ValueError("actions must be f32 [%d,%d] on %s" % (n, NUM_DOF, self.device))
        return actions if actions.is_contiguous() else actions.contiguous()

    # ------------------------------------------------------------------------------------------
    def step(self, actions):
        """fly.py:624-681 in one launch."""
        a = self._check_actions(actions)
        _lib.check(self._lib.fly_step(self._handle, C.c_void_p(a.data_ptr()), C.byref(self._bufs),
                                      _lib.stream_ptr()), "fly_step")
        self.render_count += 1

    def set_actions(self, actions):
        """fly.py:626-657 alone: scale to joint ranges and hand the targets to the sim."""
        a = self._check_actions(actions)
        _lib.check(self._lib.fly_scale_actions(self._handle, C.c_void_p(a.data_ptr()),
                                               C.c_void_p(self.actions.data_ptr()), _lib.stream_ptr()),
                   "fly_scale_actions")

    def reset(self):
        """fly.py:446-480.  Returns whether any env was flagged (one host sync, as upstream)."""
        any_flag = bool(self.reset_buf.any().item())
        if not any_flag:
            return False
        _lib.check(self._lib.fly_reset_masked(self._handle, C.byref(self._bufs), _lib.stream_ptr()),
                   "fly_reset_masked")
        return True

    def reset_async(self):
        """fly.py:446-480 without the host sync of the boolean return value."""
        _lib.check(self._lib.fly_reset_masked(self._handle, C.byref(self._bufs), _lib.stream_ptr()),
                   "fly_reset_masked")

    def simulate(self):
        """fly.py:482-485."""
        _lib.check(self._lib.fly_integrate(self._handle, C.byref(self._bufs), _lib.stream_ptr()), "fly_integrate")

    def get_obs(self):
        """fly.py:397-411."""
        _lib.check(self._lib.fly_pack_obs(self._handle, C.byref(self._bufs), _lib.stream_ptr()), "fly_pack_obs")

    def get_reward(self):
        """fly.py:413-443."""
        _lib.check(self._lib.fly_pack_reward(self._handle, C.byref(self._bufs), 0, _lib.stream_ptr()),
                   "fly_pack_reward")

    compute_observations = get_obs      # names used by BASELINE.json's north_star
    compute_reward = get_reward

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _quat_rotate(q, v):
        qw = q[:, 3:4]
        qv = q[:, :3]
        a = v * (2.0 * qw ** 2 - 1.0)
        b = torch.cross(qv, v, dim=-1) * qw * 2.0
        c = qv * (qv * v).sum(-1, keepdim=True) * 2.0
        return a + b + c

    @property
    def up_vec(self):
        """fly.py:405 output of compute_heading_and_up, derived on demand."""
        z = torch.tensor([0.0, 0.0, 1.0], device=self.device).expand(self.args.num_envs, 3)
        return self._quat_rotate(self.root_orientations, z)

    @property
    def heading_vec(self):
        x = torch.tensor([1.0, 0.0, 0.0], device=self.device).expand(self.args.num_envs, 3)
        return self._quat_rotate(self.root_orientations, x)

    def episode_stats(self, reset=False):
        """(mean return, mean length, count) over the episodes finished since the last reset of the
        statistics.  One host sync; call it when logging, not per step."""
        cnt = float(self.finished_count.sum().item())
        if cnt == 0:
            out = (float("nan"), float("nan"), 0)
        else:
            out = (float(self.finished_return_sum.sum().item()) / cnt, float(self.finished_length_sum.sum().item()) / cnt, int(cnt))
        if reset:
            self.finished_return_sum.zero_(); self.finished_length_sum.zero_(); self.finished_count.zero_()
        return out

    def reward_terms(self):
        """The per-term reward dump behind the reference viewer's P key (fly.py:504-546), as a dict
        of [N] tensors computed from the current buffers (diagnostics; not on the hot path)."""
        n, p = self.args.num_envs, self.params
        obs = self.obs_buf
        acts = self.actions.view(n, -1)
        hw, uw = float(p.heading_weight), float(p.up_weight)
        heading = torch.where(obs[:, 11] > 0.8, torch.full_like(obs[:, 11], hw), hw * obs[:, 11] / 0.8)
        up = torch.where(obs[:, 0] > 1.4, torch.full_like(heading, uw), torch.zeros_like(heading))      # fly.py:512-513
        q = self.root_orientations
        orient = torch.where(q[:, 2] ** 2 + q[:, 3] ** 2 > 0.92, torch.full_like(up, uw), torch.zeros_like(up))   # fly.py:518
        start = 12 + 2 * self.num_act
        stored = obs[:, start:start + self.num_act]
        elec = (acts - stored).abs().sum(-1)
        lim = ((stored > self.dof_limits_upper * 0.9).sum(-1) + (stored < self.dof_limits_lower * 0.9).sum(-1))
        legs = (self.force_tensor[self.index_legs_tip].sum(1).view(n, -1) > 0).long().sum(1) * 0.1
        return {"heading_reward": heading, "alive_reward": torch.full_like(heading, 0.5), "up_reward": up,
                "orient_reward": orient, "actions_cost": (acts ** 2).sum(-1), "electricity_cost": elec,
                "dof_at_limit_cost": lim * float(p.joints_at_limit_cost_scale),
                "progress_reward": self.potentials - self.prev_potentials, "leg_reward": legs}

    def render(self):
        """fly.py:487-562: there is no viewer in this build (headless only)."""
        return None

    def generate_video(self):
        """fly.py:592-610: recording needs the Isaac Gym camera API; not available here."""
        if getattr(self.args, "record", False):
            print("recording is not supported by the MI355X build (no renderer)")

    def exit(self):
        """fly.py:617-621."""
        if self._handle:
            torch.cuda.synchronize(self.device)
            _lib.check(self._lib.fly_destroy(self._handle), "fly_destroy")
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                self._lib.fly_destroy(self._handle)
                self._handle = C.c_void_p()
        except Exception:
            pass
