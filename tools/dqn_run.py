#!/usr/bin/env python3
"""DQN variant sanity run: python tools/dqn_run.py [STEPS] [N] [MINI_BATCH_STEPS]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import make_args  # noqa: E402
from fly_bproject_amd.dqn import DQN  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
mb = int(sys.argv[3]) if len(sys.argv) > 3 else 32
torch.manual_seed(0)
agent = DQN(make_args(n, dqn_mini_batch_size=mb, replay_steps=max(4 * mb, 64)))
print("replay capacity (steps):", agent.replay.capacity, "batch rows per update:", agent.batch_size)
t0 = time.perf_counter()
for _ in range(steps):
    agent.run()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("steps %d envs %d: %.1f env-steps/s, last TD loss %.5f, finite %s" %
      (steps, n, steps * n / dt, float(agent.last_loss), all(torch.isfinite(p).all().item() for p in agent.q.parameters())))
print("episode stats:", agent.env.episode_stats())
agent.exit()
