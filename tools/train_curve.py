#!/usr/bin/env python3
"""Does the loop learn?  Runs ITERS PPO iterations at N envs and prints, per iteration, the mean
per-step reward of the rollout and the episode statistics (mean return / length of finished
episodes).  Usage: python tools/train_curve.py [ITERS] [N] [backend] [variant] [gemm: f16x2 (default) | bf16x3 | f32]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import make_args  # noqa: E402
from fly_bproject_amd.ppo import PPO  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
backend = sys.argv[3] if len(sys.argv) > 3 else "hip"
variant = sys.argv[4] if len(sys.argv) > 4 else "bigGrav"
gemm = sys.argv[5] if len(sys.argv) > 5 else None
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    agent = PPO(make_args(n, update_backend=backend, variant=variant))
    if gemm:
        agent.policy.gemm = gemm
print("arithmetic: gemm=%s step_gemm=%s (%s)" % (agent.policy.gemm, agent.policy.step_gemm, agent.policy.update_path()))
t0 = time.perf_counter()
for it in range(iters):
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(agent.rollout_size):
            agent.run()
    if it % max(1, iters // 20) == 0 or it == iters - 1:
        mr, ml, cnt = agent.env.episode_stats(reset=True)
        print("iter %4d  mean step reward %.4f  episodes %6d  mean return %8.3f  mean length %7.2f  var %.4f  t=%.1fs" %
              (it, float(agent.all_reward.mean()), cnt, mr, ml, float(agent.action_var[0]), time.perf_counter() - t0), flush=True)
print("optimizer steps %d (device counter %d); updates in which an fp16x2 step was refused and redone in bf16x3: %d"
      % (agent.optim_step, int(agent.policy.step), agent.policy.h2_overflows))
agent.exit()
