#!/usr/bin/env python3
"""Rollout-only time per env step (ONE launch per rollout, HIP events) for the library in $FLYHIP_LIB: ab_rollout.py [n] [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import RolloutAllHarness, _time_launches, make_args  # noqa: E402
from fly_bproject_amd.fly import Fly  # noqa: E402
from fly_bproject_amd.policy import PackedPolicy  # noqa: E402
from fly_bproject_amd.ppo import Net  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T = 16 * (40960 // n)
torch.manual_seed(0)
env = Fly(make_args(n))
pol = PackedPolicy(Net(73, 18).to("cuda:0"), "cuda:0")
var = torch.full((18,), 0.2, device="cuda:0")
h = RolloutAllHarness(env, pol, T, var)
t = _time_launches(h.launch, reps)
sp = h.phase_split() and h.phase_split()
print("%s: n=%d T=%d  %.1f us per rollout = %.3f us per env step; policy %.2f / env %.2f us" % (
    os.environ.get("FLYHIP_LIB", "default"), n, T, t * 1e6, t * 1e6 / T, t * 1e6 / T * sp["policy_frac"], t * 1e6 / T * sp["env_frac"]))
print("    policy sub-phases (shader cycles, wave 0): %s; policy %d env %d" % (
    ", ".join("%s %d" % (k, v) for k, v in sp["policy_sub_cycles"].items()), sp["policy_cycles"], sp["env_cycles"]))
env.exit()
