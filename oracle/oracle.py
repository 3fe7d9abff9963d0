"""TEST INFRASTRUCTURE: numpy-facing wrapper over libflyoracle.so (see fly_oracle.h).

Arrays are row-major AoS: root [N,13], dof_pos/dof_vel/targets [N,18], contact [N,11,3],
obs [N,73].
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .params import NCON, NDOF, NOBS, OrcConfig, default_config  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libflyoracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_reset_masked.restype = C.c_int64
    return _LIB


def set_threads(n):
    lib().orc_set_threads(C.c_int(int(n)))


def _p(a, ty=C.c_float):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


class EnvState:
    """One batch of environments, AoS, as the reference's buffers (fly.py:169-179, :88-100)."""

    def __init__(self, n):
        self.n = n
        self.root = np.zeros((n, 13), np.float32)
        self.dof_pos = np.zeros((n, NDOF), np.float32)
        self.dof_vel = np.zeros((n, NDOF), np.float32)
        self.targets = np.zeros((n, NDOF), np.float32)
        self.contact = np.zeros((n, NCON, 3), np.float32)
        self.pot = np.full((n,), np.float32(-1000.0 / (1.0 / 60.0)), np.float32)   # fly.py:121
        self.prev_pot = self.pot.copy()
        self.obs = np.zeros((n, NOBS), np.float32)
        self.reward = np.zeros((n,), np.float32)
        self.reset = np.ones((n,), np.int64)                                        # fly.py:175
        self.progress = np.zeros((n,), np.int64)

    def copy(self):
        o = EnvState.__new__(EnvState)
        o.n = self.n
        for k, v in self.__dict__.items():
            if isinstance(v, np.ndarray):
                setattr(o, k, v.copy())
        return o


def scale_actions(cfg, actions):
    a = _f32(actions)
    out = np.empty_like(a)
    lib().orc_scale_actions(C.byref(cfg), _p(a), _p(out), C.c_int64(a.shape[0]))
    return out


def reset_masked(cfg, s):
    return lib().orc_reset_masked(C.byref(cfg), _p(s.root), _p(s.dof_pos), _p(s.dof_vel), _p(s.pot),
                                  _p(s.prev_pot), _p(s.reset, C.c_int64), _p(s.progress, C.c_int64),
                                  C.c_int64(s.n))


def physics_step(cfg, s):
    lib().orc_physics_step(C.byref(cfg), _p(s.root), _p(s.dof_pos), _p(s.dof_vel), _p(s.targets),
                           _p(s.contact), C.c_int64(s.n))


def physics_step_f64(cfg, root, dof_pos, dof_vel, targets):
    root, dof_pos, dof_vel, targets = [np.ascontiguousarray(a, np.float64).copy()
                                       for a in (root, dof_pos, dof_vel, targets)]
    contact = np.zeros((root.shape[0], NCON, 3), np.float64)
    lib().orc_physics_step_f64(C.byref(cfg), _p(root, C.c_double), _p(dof_pos, C.c_double),
                               _p(dof_vel, C.c_double), _p(targets, C.c_double),
                               _p(contact, C.c_double), C.c_int64(root.shape[0]))
    return root, dof_pos, dof_vel, contact


def pack_obs(cfg, s, want_vecs=False):
    up = np.zeros((s.n, 3), np.float32) if want_vecs else None
    hd = np.zeros((s.n, 3), np.float32) if want_vecs else None
    lib().orc_pack_obs(C.byref(cfg), _p(s.root), _p(s.dof_pos), _p(s.dof_vel), _p(s.targets),
                       _p(s.contact), _p(s.pot), _p(s.prev_pot), _p(s.obs), _p(up), _p(hd),
                       C.c_int64(s.n))
    return up, hd


def pack_reward(cfg, s):
    lib().orc_pack_reward(C.byref(cfg), _p(s.obs), _p(s.targets), _p(s.root), _p(s.contact),
                          _p(s.pot), _p(s.prev_pot), _p(s.progress, C.c_int64), _p(s.reward),
                          _p(s.reset, C.c_int64), C.c_int64(s.n))


REWARD_TERMS = ("heading_reward", "alive_reward", "up_reward", "orient_reward", "actions_cost", "electricity_cost",
                "dof_at_limit_cost", "progress_reward", "leg_reward")


def reward_terms(cfg, s):
    """fly.py:504-546 (the viewer's P-key dump) on the state's current buffers -> {name: f32 [n]}."""
    t = np.zeros((s.n, 9), np.float32)
    lib().orc_reward_terms(C.byref(cfg), _p(s.obs), _p(s.targets), _p(s.root), _p(s.contact), _p(s.pot), _p(s.prev_pot),
                           _p(t), C.c_int64(s.n))
    return {k: t[:, i].copy() for i, k in enumerate(REWARD_TERMS)}


def env_step(cfg, s, actions):
    a = _f32(actions)
    assert a.shape == (s.n, NDOF)
    lib().orc_env_step(C.byref(cfg), _p(a), _p(s.root), _p(s.dof_pos), _p(s.dof_vel), _p(s.targets),
                       _p(s.contact), _p(s.pot), _p(s.prev_pot), _p(s.obs), _p(s.reward),
                       _p(s.reset, C.c_int64), _p(s.progress, C.c_int64), C.c_int64(s.n))


def sample_logprob(mu, var, eps):
    mu, var, eps = _f32(mu), _f32(var), _f32(eps)
    act = np.empty_like(mu)
    logp = np.empty((mu.shape[0],), np.float32)
    lib().orc_sample_logprob(_p(mu), _p(var), _p(eps), _p(act), _p(logp), C.c_int64(mu.shape[0]))
    return act, logp


def td_gae(reward, v, v_next, done, gamma=0.99, lmbda=0.95, mode_flags=0):
    reward, v, v_next, done = _f32(reward), _f32(v), _f32(v_next), _f32(done)
    T, N = reward.shape[0], reward.shape[1]
    target = np.empty((T, N), np.float32)
    adv = np.empty((T, N), np.float32)
    lib().orc_td_gae(_p(reward), _p(v), _p(v_next), _p(done), C.c_float(gamma), C.c_float(lmbda),
                     C.c_int64(T), C.c_int64(N), _p(target), _p(adv), C.c_int(mode_flags))
    return target, adv


NET_KEYS = ["shared_net.0", "shared_net.2", "to_mean.0", "to_mean.2", "to_value.0", "to_value.2"]


def net_forward(state_dict, x, head):
    """state_dict: name -> ndarray with the reference's key names (ppo.py:18-38)."""
    x = _f32(x)
    ws = [_f32(state_dict[k + ".weight"]) for k in NET_KEYS]
    bs = [_f32(state_dict[k + ".bias"]) for k in NET_KEYS]
    PP = C.POINTER(C.c_float) * 6
    w = PP(*[_p(a) for a in ws])
    b = PP(*[_p(a) for a in bs])
    out = np.empty((x.shape[0], NDOF if head == 0 else 1), np.float32)
    lib().orc_net_forward(w, b, _p(x), C.c_int64(x.shape[0]), C.c_int(head), _p(out))
    return out


def dqn_eps_greedy(q, coin_u, rand_u, epsilon):
    q, coin_u, rand_u = _f32(q), _f32(coin_u), _f32(rand_u)
    out = np.empty((q.shape[0],), np.float32)
    lib().orc_dqn_eps_greedy(_p(q), _p(coin_u), _p(rand_u), C.c_float(epsilon), C.c_int(q.shape[1]), _p(out),
                             C.c_int64(q.shape[0]))
    return out


def dqn_huber_td(q_table, act, reward, q_next, done, discount=0.99):
    q_table, act, reward, q_next, done = _f32(q_table), _f32(act), _f32(reward), _f32(q_next), _f32(done)
    dq = np.empty_like(q_table)
    loss = C.c_float()
    lib().orc_dqn_huber_td(_p(q_table), _p(act), _p(reward), _p(q_next), _p(done), C.c_float(discount),
                           C.c_int(q_table.shape[1]), C.c_int64(q_table.shape[0]), _p(dq), C.byref(loss))
    return dq, loss.value
