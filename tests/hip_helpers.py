"""Helpers shared by the GPU parity tests: oracle <-> product buffer conversion."""
import types

import numpy as np
import torch

from oracle import oracle as O


def make_args(n, **kw):
    d = dict(sim_device="cuda:0", num_envs=n, headless=True, testing=False, save=False, load=False,
             record=False, save_freq=100, save_path=None, load_path=None, seed=0, rank=0, world_size=1,
             variant="bigGrav", reward="standing")
    d.update(kw)
    return types.SimpleNamespace(**d)


def make_env(n, variant="bigGrav", **kw):
    from fly_bproject_amd.fly import Fly
    return Fly(make_args(n, variant=variant, **kw))


def push_state(env, s):
    """oracle EnvState (numpy AoS) -> product buffers."""
    dev = env.device
    n = s.n
    env.root_tensor.copy_(torch.from_numpy(s.root).to(dev))
    ds = np.stack([s.dof_pos, s.dof_vel], axis=-1).reshape(n, 36)
    env.dof_states.copy_(torch.from_numpy(ds).to(dev))
    env.actions.copy_(torch.from_numpy(s.targets.reshape(-1, 1)).to(dev))
    env.force_tensor.copy_(torch.from_numpy(s.contact.reshape(n * 11, 3)).to(dev))
    env.potentials.copy_(torch.from_numpy(s.pot).to(dev))
    env.prev_potentials.copy_(torch.from_numpy(s.prev_pot).to(dev))
    env.obs_buf.copy_(torch.from_numpy(s.obs).to(dev))
    env.reward_buf.copy_(torch.from_numpy(s.reward).to(dev))
    env.reset_buf.copy_(torch.from_numpy(s.reset).to(dev))
    env.progress_buf.copy_(torch.from_numpy(s.progress).to(dev))


def pull_state(env):
    """product buffers -> oracle-shaped EnvState."""
    n = env.args.num_envs
    torch.cuda.synchronize()
    s = O.EnvState(n)
    s.root[:] = env.root_tensor.cpu().numpy()
    ds = env.dof_states.cpu().numpy().reshape(n, 18, 2)
    s.dof_pos[:] = ds[..., 0]
    s.dof_vel[:] = ds[..., 1]
    s.targets[:] = env.actions.cpu().numpy().reshape(n, 18)
    s.contact[:] = env.force_tensor.cpu().numpy().reshape(n, 11, 3)
    s.pot[:] = env.potentials.cpu().numpy()
    s.prev_pot[:] = env.prev_potentials.cpu().numpy()
    s.obs[:] = env.obs_buf.cpu().numpy()
    s.reward[:] = env.reward_buf.cpu().numpy()
    s.reset[:] = env.reset_buf.cpu().numpy()
    s.progress[:] = env.progress_buf.cpu().numpy()
    return s


def pose_actions(cfg, n):
    lo = np.array(cfg.dof_lo[:]); hi = np.array(cfg.dof_hi[:]); pose = np.array(cfg.dof_pose[:])
    return np.tile(((2 * pose - hi - lo) / (hi - lo)).astype(np.float32), (n, 1))


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
