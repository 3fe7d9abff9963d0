#!/usr/bin/env python3
"""Host-enqueue time vs GPU time of the rollout (80 env steps at 8192 envs) and of one update."""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import make_args  # noqa: E402
from fly_bproject_amd.ppo import PPO  # noqa: E402

N = int(os.environ.get("PROF_ENVS", "8192"))
with contextlib.redirect_stdout(io.StringIO()):
    agent = PPO(make_args(N))
    agent.args.testing = True          # rollout only
    for _ in range(agent.rollout_size):
        agent.run()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(agent.rollout_size):
            agent.run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("rollout: host enqueue %.2f ms, until GPU done %.2f ms (%d steps)" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, agent.rollout_size))
agent.args.testing = False
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agent.update()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("update: host enqueue %.2f ms, until GPU done %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
agent.exit()
