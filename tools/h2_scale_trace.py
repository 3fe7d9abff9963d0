#!/usr/bin/env python3
"""Per optimizer step of real PPO updates: the largest |value| of every tensor class of the fp16x2 step (x, h1..h3, dz4..dz1), from the
maxima the kernel tracks -- how far a class maximum moves from one minibatch to the next is what the scale headroom has to cover.
   python tools/h2_scale_trace.py [iterations] [num_envs]"""
import contextlib
import io
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fly_bproject_amd.policy import PackedPolicy  # noqa: E402
from fly_bproject_amd.ppo import PPO  # noqa: E402
from tests.hip_helpers import make_args  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    agent = PPO(make_args(n))
    agent.policy.gemm = "f16x2"
trace = []
orig = PackedPolicy.minibatch_grad


def patched(self, *a, **kw):
    live = kw.get("fuse_norm") and self.h2_live() and not self.h2_suspended
    if live:
        s_old = self.h2_scales[:8].clone()
    r = orig(self, *a, **kw)
    if live:
        m = self.h2_scales[32:40].clone()
        trace.append((m / s_old).cpu().tolist() + [int(self.h2_overflow.item())])
    return r


PackedPolicy.minibatch_grad = patched
with contextlib.redirect_stdout(io.StringIO()):
    for it in range(iters):
        for _ in range(agent.rollout_size):
            agent.run()
torch.cuda.synchronize()
names = ["x", "h1", "h2", "h3", "dz4", "dz3", "dz2", "dz1"]
print("updates: %d, traced fp16x2 steps: %d, updates with an overflow: %d" % (iters, len(trace), agent.policy.h2_overflows))
print("step  " + "  ".join("%9s" % nm for nm in names) + "  ovf")
worst = [0.0] * 8
prev = None
for i, row in enumerate(trace):
    if prev is not None:
        for c in range(8):
            if row[c] > 0 and prev[c] > 0:
                worst[c] = max(worst[c], abs(math.log2(row[c] / prev[c])))
    if i < 160 or row[8]:
        print("%4d  " % i + "  ".join("%9.3g" % v for v in row[:8]) + "  %d" % row[8])
    prev = row
print("largest step-to-step move of a class maximum, in binades: " + "  ".join("%s %.1f" % (nm, w) for nm, w in zip(names, worst)))
