#!/usr/bin/env python3
"""Rollout step, eager against hipGraph replay (PPO(graph=True)): wall time per env step over whole rollouts
(no updates: testing=True), at N envs.  Run it under `rocprofv3 --kernel-trace --stats` to see the device side:
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/graph_vs_eager.py graph
usage: graph_vs_eager.py [eager|graph|both] [N]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import make_args  # noqa: E402
from fly_bproject_amd.ppo import PPO  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
for graph in ([False, True, "persistent"] if mode == "both" else [{"graph": True, "eager": False}.get(mode, "persistent")]):
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        agent = PPO(make_args(n, graph=graph is True, persistent_rollout=graph == "persistent", testing=True))
        T = agent.rollout_size
        for _ in range(2 * T):                      # rollout 1 eager (captures happen in rollout 2 for graph=True)
            agent.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10 * T):
            agent.run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # host-only cost of issuing one step (no sync in between): the GPU queue absorbs it when the step is GPU-bound
        t1 = time.perf_counter()
        for _ in range(T):
            agent.run()
        host = time.perf_counter() - t1
        torch.cuda.synchronize()
    print("%s  %d envs: %.2f us per env step (wall, 10 rollouts); host issue time %.2f us per step; launches per step: %s"
          % ({True: "graph", False: "eager"}.get(graph, "one launch per rollout"), n, dt / (10 * T) * 1e6, host / T * 1e6,
             {True: "one graph replay per ROLLOUT (T rollout_step nodes)", False: "1 (rollout_step, bookkeeping deferred)"}.get(graph, "1 / T (ppo_rollout_all)")))
    agent.exit()
