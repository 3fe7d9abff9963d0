#!/usr/bin/env python3
"""Rollouts launched as ONE kernel against step-by-step launches, at env counts around one tile per CU: where (step, env,
column) do the rows first differ?  Usage: debug_rollout_multi.py [train] [n ...]   (train: variance decays + updates)"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import make_args  # noqa: E402
from fly_bproject_amd.ppo import PPO  # noqa: E402

args = sys.argv[1:]
train = bool(args) and args[0] == "train"
if train:
    args = args[1:]


def snap(agent, T):
    return {"obs": agent._obs_ring.clone(), "act": agent.all_acts.clone(), "logp": agent.all_log_prob.clone(),
            "rew": agent.all_reward.clone(), "v": agent._v_ring[:T].clone(), "P": agent.policy.P.clone(),
            "var": agent._action_var.clone(), "adv": agent.all_advantage.clone(), "tgt": agent._target.clone()}


for n in [int(a) for a in args] or [8192, 8224, 16384]:
    res = {}
    for persistent in (False, True):
        torch.manual_seed(0)
        snaps = []
        with contextlib.redirect_stdout(io.StringIO()):
            agent = PPO(make_args(n, persistent_rollout=persistent, testing=not train))
            T = agent.rollout_size
            for r in range(4):
                for t in range(T):
                    agent.run()
                torch.cuda.synchronize()
                snaps.append(("after rollout %d (+ update)" % r, snap(agent, T)))
        res[persistent] = snaps
        agent.exit()
    print("n = %d (T = %d, %d tiles), train=%s, FLY_ROLLOUT_FS=%s" % (n, T, (n + 31) // 32, train, os.environ.get("FLY_ROLLOUT_FS")))
    done = False
    for (name, sa), (_, sb) in zip(res[False], res[True]):
        for k in sa:
            a, b = sa[k], sb[k]
            if torch.equal(a, b):
                continue
            d = (a != b)
            idx = torch.nonzero(d)
            msg = "   %s: %-5s DIFFERS: %d elements, first at %s" % (name, k, int(d.sum()), idx[0].tolist())
            if idx.shape[1] > 1 and a.dim() >= 2 and a.shape[1] == n:
                envs = torch.unique(idx[:, 1])
                tiles = torch.unique(envs // 32)
                msg += "; %d envs in %d tiles (min %d max %d); steps %d..%d" % (envs.numel(), tiles.numel(), int(tiles.min()), int(tiles.max()),
                                                                               int(idx[:, 0].min()), int(idx[:, 0].max()))
            print(msg)
            done = True
        if done:
            break
    if not done:
        print("   everything equal at every checkpoint")
