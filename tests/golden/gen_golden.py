#!/usr/bin/env python3
"""Golden-vector generator.  Runs ONLY in the build container (needs /root/reference).

It imports the reference's own `fly.py` / `ppo.py` (helper math for the two absent NVIDIA
packages comes from oracle/isaac_stubs, see its README) and records inputs + the reference's
outputs as small .npz fixtures.  Fixtures hold numbers only; nothing of the reference travels.

    python tests/golden/gen_golden.py          # rewrites tests/golden/g*.npz

Fixtures
  g1_obs      compute_fly_observations                      fly.py:771-805
  g2_reward   compute_fly_reward2                           fly.py:685-768
  g3_step     Fly.step / Fly.reset orchestration (bigGrav and lowGrav ordering) with the
              oracle's FlyDyn plugged in as `simulate`      fly.py:624-681, :446-480
  g4_net      Net.pi / Net.v                                ppo.py:10-102
  g5_sample   MultivariateNormal sample / log_prob / clip   ppo.py:213-220
  g6_gae      PPO.make_data                                 ppo.py:157-171
  g7_update   PPO.update (75 optimizer steps)               ppo.py:173-202
  g8_dqn      DQN.act / DQN.update (TD loss + gradient)     UselessFiles/dqn.py:64-100
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("FLY_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REPO, "oracle", "isaac_stubs"))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import fly as ref_fly  # noqa: E402  (the reference)
import ppo as ref_ppo  # noqa: E402  (the reference)
from oracle import oracle as O  # noqa: E402

torch.set_num_threads(1)
NA, NO, NC = 18, 73, 11


def rand_unit_quats(rng, n):
    q = rng.normal(size=(n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)


def pack_inputs(rng, n, cfg):
    """Random but adversarial pack inputs: near-upright and arbitrary orientations, gimbal
    cases |sinp|->1, heights straddling every threshold, contact rows of all signs."""
    root = np.zeros((n, 13), np.float32)
    root[:, 0:2] = rng.normal(0, 3, (n, 2))
    root[:, 2] = rng.uniform(0.5, 7.0, n)
    q = rand_unit_quats(rng, n)
    k = n // 2
    small = rng.normal(0, 0.08, (k, 3)).astype(np.float32)        # near upright half
    q[:k, :3] = small
    q[:k, 3] = 1.0
    q[:k] /= np.linalg.norm(q[:k], axis=1, keepdims=True)
    # gimbal: rotation of +-90deg about y -> sinp = +-1 (up to rounding)
    s = np.float32(np.sqrt(0.5))
    q[0] = [0, s, 0, s]
    q[1] = [0, -s, 0, s]
    q[2] = [0, 0, 0, 1]
    root[:, 3:7] = q
    # exact threshold heights
    for i, z in enumerate([1.1, 1.4, 2.1, 6.0, 1.0999999, 1.4000001, 2.0999999, 6.0000005]):
        root[3 + i, 2] = np.float32(z)
    root[:, 7:10] = rng.normal(0, 5, (n, 3))
    root[:, 10:13] = rng.normal(0, 3, (n, 3))
    lo = np.array(cfg.dof_lo[:], np.float32)
    hi = np.array(cfg.dof_hi[:], np.float32)
    dof_pos = rng.uniform(lo, hi, (n, NA)).astype(np.float32)
    dof_vel = rng.normal(0, 1, (n, NA)).astype(np.float32)
    act = rng.uniform(-1, 1, (n, NA)).astype(np.float32)
    act[::5] = np.sign(act[::5])                                  # saturated -> "at limit" counts
    targets = O.scale_actions(cfg, act)
    contact = np.zeros((n, NC, 3), np.float32)
    m = rng.random((n, NC)) < 0.5
    contact[m] = rng.normal(0, 1, (int(m.sum()), 3)).astype(np.float32)
    contact[:, :5][rng.random((n, 5)) < 0.8] = 0.0                # abdomen mostly clear
    pot = rng.normal(-60000, 10, n).astype(np.float32)
    return root, dof_pos, dof_vel, targets, contact, pot


def ref_tables(n, cfg):
    lo = torch.tensor(cfg.dof_lo[:], dtype=torch.float32)
    hi = torch.tensor(cfg.dof_hi[:], dtype=torch.float32)
    action_idx = torch.arange(NA, dtype=torch.long).view(NA, 1)            # fly.py:288-290
    base = (torch.arange(n, dtype=torch.long) * NC).view(n, 1)
    idx_abd = (base + torch.arange(0, 5)).reshape(-1)                      # fly.py:311-314
    idx_leg = (base + torch.arange(5, 11)).reshape(-1)
    return lo, hi, action_idx, idx_abd, idx_leg


def gen_obs_reward(rng):
    out_obs, out_rew = {}, {}
    for n in (16, 257):
        cfg = O.default_config(n)
        root, dof_pos, dof_vel, targets, contact, pot = pack_inputs(rng, n, cfg)
        lo, hi, action_idx, idx_abd, idx_leg = ref_tables(n, cfg)
        t = torch.from_numpy
        tgt = torch.tensor([1000.0, 0.0, 0.0]).repeat(n, 1)
        inv_start = torch.tensor([-0.0, -0.0, -0.0, 1.0]).repeat(n, 1)
        b0 = torch.tensor([1.0, 0.0, 0.0]).repeat(n, 1)
        b1 = torch.tensor([0.0, 0.0, 1.0]).repeat(n, 1)
        force = t(contact.reshape(n * NC, 3).copy())
        obs, pot_new, prev_pot_new, up_vec, heading_vec = ref_fly.compute_fly_observations(
            torch.zeros(n, NO), t(root.copy()), tgt, t(pot.copy()), inv_start, t(dof_pos.copy()),
            t(dof_vel.copy()), lo, hi, 0.2, t(targets.copy()), 1 / 60, b0, b1, 2, action_idx,
            force, idx_leg, n)
        p = "n%d_" % n
        out_obs.update({p + "root": root, p + "dof_pos": dof_pos, p + "dof_vel": dof_vel,
                        p + "targets": targets, p + "contact": contact, p + "pot_in": pot,
                        p + "obs": obs.numpy(), p + "pot": pot_new.numpy(),
                        p + "prev_pot": prev_pot_new.numpy(), p + "up_vec": up_vec.numpy(),
                        p + "heading_vec": heading_vec.numpy()})
        # reward: feed the reference obs, plus a variant whose obs[48:66] differs from the actions
        progress = rng.choice([1, 700, 1498, 1499, 1500], n).astype(np.int64)
        reset_in = (rng.random(n) < 0.1).astype(np.int64)
        obs_r = obs.numpy().copy()
        obs_r[n // 3:, 48:66] += rng.normal(0, 0.3, (n - n // 3, NA)).astype(np.float32)
        for variant, ecs in (("big", 0.005), ("low", 1.0)):
            reward, reset = ref_fly.compute_fly_reward2(
                t(obs_r.copy()), t(reset_in.copy()), t(progress.copy()), t(targets.copy()),
                0.75, 0.5, pot_new, prev_pot_new, 0.005, ecs, 0.1, 1.1, 6.0, -2.0, 1500.0,
                action_idx, hi, lo, n, t(root[:, 3:7].copy()), NA, force, idx_abd, idx_leg)
            out_rew.update({p + variant + "_reward": reward.numpy(), p + variant + "_reset": reset.numpy()})
        out_rew.update({p + "obs": obs_r, p + "targets": targets, p + "root": root,
                        p + "contact": contact, p + "pot": pot_new.numpy(),
                        p + "prev_pot": prev_pot_new.numpy(), p + "progress": progress,
                        p + "reset_in": reset_in})
    np.savez_compressed(os.path.join(HERE, "g1_obs.npz"), **out_obs)
    np.savez_compressed(os.path.join(HERE, "g2_reward.npz"), **out_rew)


class _DummyGym:
    """Stands where the Isaac Gym handle was: records the PD targets, everything else no-op."""

    def __init__(self):
        self.targets = None

    def set_dof_position_target_tensor(self, sim, t):
        self.targets = t.clone()
        return True

    def __getattr__(self, name):
        return lambda *a, **k: True


class GoldenFly(ref_fly.Fly):
    """The reference Fly with its own step/reset/get_obs/get_reward untouched; only __init__
    (which needs a simulator) and simulate() (closed-source PhysX) are replaced.  simulate() runs
    the oracle's build-defined FlyDyn on the same tensors the reference aliases."""

    def __init__(self, n, cfg, reset_after):
        self.cfg = cfg
        self.args = types.SimpleNamespace(num_envs=n, sim_device="cpu", headless=True, record=False)
        self.print_once = False
        self.end = False
        self.dt = 1 / 60
        self.up_axis_idx = 2
        self.num_act = NA
        self.num_obs = 19 + 3 * NA
        self.starting_height = 2
        self.max_episode_length = 1500
        self.render_count = 0
        self.dof_vel_scale = 0.2
        self.heading_weight = 0.5
        self.up_weight = 0.75
        self.actions_cost_scale = 0.005
        self.energy_cost_scale = float(np.float32(cfg.energy_cost_scale))
        self.joints_at_limit_cost_scale = 0.1
        self.death_cost = -2.0
        self.termination_height = 1.1
        self.termination_height_up = 6
        self.obs_buf, self.reward_buf, self.reset_buf, self.progress_buf = self.init_buffers()
        self.gym = _DummyGym()
        self.sim = None
        self.num_dof = NA
        lo, hi, action_idx, idx_abd, idx_leg = ref_tables(n, cfg)
        self.dof_limits_lower, self.dof_limits_upper = lo, hi
        self.action_indexes_one = action_idx
        self.action_indexes = torch.arange(n * NA, dtype=torch.long).view(-1, 1)
        self.initial_dofs = torch.zeros((n * NA, 2), dtype=torch.float32)
        self.initial_dofs[:, 0] = torch.tensor(cfg.dof_pose[:], dtype=torch.float32).repeat(n)
        self.initial_dofs_one = self.initial_dofs[:NA]
        self.dof_states = torch.zeros(n, NA * 2)
        self.root_tensor = torch.zeros(n, 13)
        self.force_tensor = torch.zeros(n * NC, 3)
        self.dof_pos = self.dof_states.view(n, NA, 2)[..., 0]
        self.dof_vel = self.dof_states.view(n, NA, 2)[..., 1]
        self.index_abdomen_sim, self.index_legs_tip = idx_abd, idx_leg
        self.root_orientations = self.root_tensor.view(n, 13)[:, 3:7]
        self.origin_root_tensor = self.create_origin_root_tensor()
        self.potentials = torch.tensor([-1000. / self.dt], dtype=torch.float32).repeat(n)
        self.prev_potentials = self.potentials.clone()
        self.up_vec = torch.tensor([0.0, 0.0, 1.0]).repeat(n, 1)
        self.heading_vec = torch.tensor([1.0, 0.0, 0.0]).repeat(n, 1)
        self.inv_start_rot = torch.tensor([-0.0, -0.0, -0.0, 1.0]).repeat(n, 1)
        self.basis_vec0 = self.heading_vec.clone()
        self.basis_vec1 = self.up_vec.clone()
        self.targets = torch.tensor([1000.0, 0.0, 0.0]).repeat(n, 1)
        self._reset_after = reset_after

    def simulate(self):
        n = self.args.num_envs
        s = O.EnvState(n)
        s.root[:] = self.root_tensor.numpy()
        ds = self.dof_states.view(n, NA, 2).numpy()
        s.dof_pos[:] = ds[..., 0]
        s.dof_vel[:] = ds[..., 1]
        s.targets[:] = self.gym.targets.view(n, NA).numpy()
        O.physics_step(self.cfg, s)
        self.root_tensor[:] = torch.from_numpy(s.root)
        self.dof_states.view(n, NA, 2)[..., 0] = torch.from_numpy(s.dof_pos)
        self.dof_states.view(n, NA, 2)[..., 1] = torch.from_numpy(s.dof_vel)
        self.force_tensor[:] = torch.from_numpy(s.contact.reshape(n * NC, 3))


def gen_step(rng):
    out = {}
    for variant, reset_after in (("bigGrav", 0), ("lowGrav", 1)):
        n, steps = 16, 48
        cfg = O.default_config(n, variant)
        env = GoldenFly(n, cfg, reset_after)
        lo = np.array(cfg.dof_lo[:]); hi = np.array(cfg.dof_hi[:]); pose = np.array(cfg.dof_pose[:])
        a0 = ((2 * pose - hi - lo) / (hi - lo)).astype(np.float32)
        acts = np.clip(a0 + rng.normal(0, 0.6, (steps, n, NA)), -1, 1).astype(np.float32)
        acts[:, 0] = a0
        acts[:, 1] = 1.0              # drives every joint to its upper limit: falls over
        rec = {k: [] for k in ("obs", "reward", "reset", "progress", "root", "dof_pos", "dof_vel", "pot", "prev_pot")}
        for t in range(steps):
            if variant == "lowGrav":
                # flyLowGrav.py:657-663 ordering: simulate, then reset.  The reference keeps two
                # files for this; here the same class is driven in that order.
                ref_step_lowgrav(env, torch.from_numpy(acts[t].copy()))
            else:
                env.step(torch.from_numpy(acts[t].copy()))
            if t == 20:
                env.progress_buf[3] = 1497                      # exercise the episode-length reset
            rec["obs"].append(env.obs_buf.numpy().copy())
            rec["reward"].append(env.reward_buf.numpy().copy())
            rec["reset"].append(env.reset_buf.numpy().copy())
            rec["progress"].append(env.progress_buf.numpy().copy())
            rec["root"].append(env.root_tensor.numpy().copy())
            rec["dof_pos"].append(env.dof_pos.numpy().copy())
            rec["dof_vel"].append(env.dof_vel.numpy().copy())
            rec["pot"].append(env.potentials.numpy().copy())
            rec["prev_pot"].append(env.prev_potentials.numpy().copy())
        out[variant + "_actions"] = acts
        for k, v in rec.items():
            out[variant + "_" + k] = np.stack(v)
    np.savez_compressed(os.path.join(HERE, "g3_step.npz"), **out)


def ref_step_lowgrav(env, actions):
    """Drive the reference's own pieces in flyLowGrav.py's order (its step() differs from
    fly.py's only by calling self.reset() after self.simulate(): diff lines 657-663)."""
    low = getattr(ref_step_lowgrav, "_mod", None)
    if low is None:
        import flyLowGrav as low  # the reference's second env file
        ref_step_lowgrav._mod = low
    low.Fly.step(env, actions)


def gen_net(rng):
    torch.manual_seed(0)
    net = ref_ppo.Net(NO, NA)
    x = torch.from_numpy(rng.normal(0, 1.5, (64, NO)).astype(np.float32))
    with torch.no_grad():
        pi, v = net.pi(x), net.v(x)
    out = {k: t.numpy() for k, t in net.state_dict().items()}
    out.update(x=x.numpy(), pi=pi.numpy(), v=v.numpy())
    np.savez_compressed(os.path.join(HERE, "g4_net.npz"), **out)
    return net


def gen_sample(rng):
    from torch.distributions import MultivariateNormal
    from torch.distributions.utils import _standard_normal
    n = 64
    mu = torch.from_numpy(rng.normal(0, 0.7, (n, NA)).astype(np.float32))
    out = {"mu": mu.numpy()}
    for tag, var0 in (("v02", 0.2), ("v001", 0.01), ("vmix", None)):
        var = torch.full((NA,), var0 if var0 else 0.0)
        if var0 is None:
            var = torch.from_numpy(rng.uniform(0.01, 0.2, NA).astype(np.float32))
        scale_tril = torch.cholesky(torch.diag(var))                      # ppo.py:215-216
        dist = MultivariateNormal(mu, scale_tril=scale_tril)
        torch.manual_seed(123)
        eps = _standard_normal(mu.shape, dtype=mu.dtype, device=mu.device)
        torch.manual_seed(123)
        action = dist.sample()                                            # ppo.py:218
        assert torch.allclose(action, mu + torch.sqrt(var) * eps, atol=1e-6)
        logp = dist.log_prob(action)                                      # ppo.py:219
        clipped = action.clip(-1, 1)                                      # ppo.py:220
        # log-prob of a stored (clipped) action under another variance: the update's use, ppo.py:189
        logp_clipped = dist.log_prob(clipped)
        out.update({tag + "_var": var.numpy(), tag + "_eps": eps.numpy(), tag + "_action": action.numpy(),
                    tag + "_logp": logp.numpy(), tag + "_clipped": clipped.numpy(),
                    tag + "_logp_clipped": logp_clipped.numpy()})
    np.savez_compressed(os.path.join(HERE, "g5_sample.npz"), **out)


def bare_ppo(net, T, N, mini_chunk):
    p = ref_ppo.PPO.__new__(ref_ppo.PPO)
    p.args = types.SimpleNamespace(num_envs=N, sim_device="cpu", testing=False, save=False)
    p.net = net
    p.epoch, p.lr, p.gamma, p.lmbda, p.clip = 5, 0.001, 0.99, 0.95, 0.2
    p.mini_chunk_size, p.rollout_size = mini_chunk, T
    p.optim_step = 0
    p.all_advantage = torch.zeros(T, N, 1)
    return p


def gen_gae_update(rng, net):
    out6 = {}
    for tag, T, N in (("a", 64, 16), ("b", 80, 96)):
        p = bare_ppo(net, T, N, T // 16)
        p.all_obs = torch.from_numpy(rng.normal(0, 1, (T, N, NO)).astype(np.float32))
        p.all_next_obs = torch.from_numpy(rng.normal(0, 1, (T, N, NO)).astype(np.float32))
        p.all_reward = torch.from_numpy(rng.normal(0.5, 1, (T, N, 1)).astype(np.float32))
        p.all_done = (1 - torch.from_numpy((rng.random(N) < 0.2).astype(np.int64))).unsqueeze(-1)  # ppo.py:230
        p.all_acts = torch.zeros(T, N, NA)
        p.all_log_prob = torch.zeros(T, N)
        with torch.no_grad():
            v = net.v(p.all_obs); vn = net.v(p.all_next_obs)
        _, _, _, target, adv = p.make_data()
        if tag == "a":      # keep the fixture small: only the small case carries the obs rows
            out6.update({tag + "_obs": p.all_obs.numpy(), tag + "_next_obs": p.all_next_obs.numpy()})
        out6.update({tag + "_reward": p.all_reward.numpy(), tag + "_done": p.all_done.numpy(),
                     tag + "_v": v.numpy(), tag + "_v_next": vn.numpy(),
                     tag + "_target": target.numpy(), tag + "_adv": adv.numpy().copy()})
    np.savez_compressed(os.path.join(HERE, "g6_gae.npz"), **out6)

    # g7: one full PPO.update on a tiny rollout (T=32, N=8, mini_chunk=2 -> 15 minibatches x 5 epochs)
    torch.manual_seed(1)
    net7 = ref_ppo.Net(NO, NA)
    T, N = 32, 8
    p = bare_ppo(net7, T, N, 2)
    p.all_obs = torch.from_numpy(rng.normal(0, 1, (T, N, NO)).astype(np.float32))
    p.all_next_obs = torch.from_numpy(rng.normal(0, 1, (T, N, NO)).astype(np.float32))
    p.all_reward = torch.from_numpy(rng.normal(0.5, 1, (T, N, 1)).astype(np.float32))
    p.all_done = (1 - torch.from_numpy((rng.random(N) < 0.2).astype(np.int64))).unsqueeze(-1)
    p.all_acts = torch.from_numpy(rng.uniform(-1, 1, (T, N, NA)).astype(np.float32))
    p.all_log_prob = torch.from_numpy(rng.normal(-2, 1, (T, N)).astype(np.float32))
    p.action_var = torch.full((NA,), 0.15)
    p.optim = torch.optim.Adam(net7.parameters(), lr=p.lr)
    out7 = {"w0_" + k: t.numpy().copy() for k, t in net7.state_dict().items()}
    out7.update(obs=p.all_obs.numpy(), next_obs=p.all_next_obs.numpy(), reward=p.all_reward.numpy(),
                done=p.all_done.numpy(), acts=p.all_acts.numpy(), log_prob=p.all_log_prob.numpy(),
                action_var=p.action_var.numpy())
    # first-minibatch loss and clipped-gradient norm, from the same code path (ppo.py:184-198)
    obs, action, old_lp, target, adv = p.make_data()
    import torch.nn.functional as F
    from torch.distributions import MultivariateNormal
    mu = net7.pi(obs[0:2])
    dist = MultivariateNormal(mu, scale_tril=torch.cholesky(torch.diag(p.action_var)))
    ratio = torch.exp(dist.log_prob(action[0:2]) - old_lp[0:2]).unsqueeze(-1)
    loss = -torch.min(ratio * adv[0:2], torch.clamp(ratio, 0.8, 1.2) * adv[0:2]) + \
        F.smooth_l1_loss(net7.v(obs[0:2]), target[0:2])
    net7.zero_grad()
    loss.mean().backward()
    gn = torch.nn.utils.clip_grad_norm_(net7.parameters(), 1.0)
    out7.update(loss0=loss.mean().detach().numpy(), gradnorm0=gn.numpy(),
                target=target.numpy(), adv=adv.numpy().copy())
    out7.update({"g0_" + k: t.grad.numpy().copy() for k, t in net7.named_parameters()})
    net7.zero_grad()
    p.update()                                                            # the reference's own loop
    assert p.optim_step == 75
    out7.update({"w1_" + k: t.numpy().copy() for k, t in net7.state_dict().items()})
    np.savez_compressed(os.path.join(HERE, "g7_update.npz"), **out7)


def gen_dqn(rng):
    """The reference's own DQN.act and DQN.update driven through a bare object (no env, a stub
    replay that returns fixed tensors).  num_obs is 73 here (the stale default 84 does not match
    the current Fly)."""
    import UselessFiles.dqn as ref_dqn
    n, A = 96, 18
    torch.manual_seed(5)
    d = ref_dqn.DQN.__new__(ref_dqn.DQN)
    d.args = types.SimpleNamespace(num_envs=n, sim_device="cpu")
    d.act_space, d.discount, d.mini_batch_size = A, 0.99, 4
    d.batch_size = n * d.mini_batch_size
    d.tau = 0.995
    d.q = ref_dqn.Net(num_obs=NO, num_act=A)
    d.q_target = ref_dqn.Net(num_obs=NO, num_act=A)
    ref_dqn.soft_update(d.q, d.q_target, tau=0.0)
    d.optimizer = torch.optim.Adam(d.q.parameters(), lr=3e-4)
    out = {"q_" + k: v.numpy().copy() for k, v in d.q.state_dict().items()}
    obs = torch.from_numpy(rng.normal(0, 1, (n, NO)).astype(np.float32))
    for tag, eps in (("e08", 0.8), ("e001", 0.01)):
        torch.manual_seed(77)
        coin_u = torch.rand(n); rand_u = torch.rand(n)              # the two draws of dqn.py:90-92, in order
        torch.manual_seed(77)
        act = d.act(obs, eps)
        with torch.no_grad():
            out[tag + "_q"] = d.q(obs).numpy()
        out.update({tag + "_coin_u": coin_u.numpy(), tag + "_rand_u": rand_u.numpy(), tag + "_act": act.numpy()})
    out["obs"] = obs.numpy()
    B = d.batch_size
    b_obs = torch.from_numpy(rng.normal(0, 1, (B, NO)).astype(np.float32))
    b_next = torch.from_numpy(rng.normal(0, 1, (B, NO)).astype(np.float32))
    b_act = torch.from_numpy((rng.integers(0, A, B) / (A - 1) * 2 - 1).astype(np.float32))
    b_rew = torch.from_numpy(rng.normal(0.5, 1.5, B).astype(np.float32))
    b_done = torch.from_numpy((rng.random(B) < 0.9).astype(np.float32))
    d.replay = types.SimpleNamespace(sample=lambda m: (b_obs, b_act, b_rew, b_next, b_done))
    grabbed = {}
    with torch.no_grad():
        q_table = d.q(b_obs).numpy().copy()
        q_next = d.q_target(b_next).numpy().copy()
    def _grab(mod, inp, outp):          # gradient of the loss w.r.t. the Q table (returns None: output unchanged)
        outp.register_hook(lambda g: grabbed.__setitem__("dq", g.clone()))
    handle = d.q.net[-1].register_forward_hook(_grab)
    loss = d.update()
    handle.remove()
    out.update(b_act=b_act.numpy(), b_rew=b_rew.numpy(), b_done=b_done.numpy(), q_table=q_table, q_next=q_next,
               loss=loss.detach().numpy(), dq=grabbed["dq"].numpy())
    out.update({"q1_" + k: v.numpy().copy() for k, v in d.q.state_dict().items()})
    out.update({"qt1_" + k: v.numpy().copy() for k, v in d.q_target.state_dict().items()})
    out.update(b_obs=b_obs.numpy(), b_next=b_next.numpy())
    np.savez_compressed(os.path.join(HERE, "g8_dqn.npz"), **out)


def main():
    rng = np.random.default_rng(20250202)
    O.build()
    gen_obs_reward(rng)
    gen_step(rng)
    net = gen_net(rng)
    gen_sample(rng)
    gen_gae_update(rng, net)
    gen_dqn(rng)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
