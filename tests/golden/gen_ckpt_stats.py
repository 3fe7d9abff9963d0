#!/usr/bin/env python3
"""Build-container only: per-layer weight statistics of a few of the reference's current-architecture checkpoints
(/root/reference/saves/save9_1_23/*.pth, save8_bigGrav/*.pth; loaded weights-only, never unpickling code) and the size of the reference
Net's outputs on seeded observations (ppo.py:147-153) -> the constants of tests/test_trained_stats_gpu.py.  NUMBERS ONLY travel:
no checkpoint and no reference source leaves this container.
    python tests/golden/gen_ckpt_stats.py  > /tmp/stats.txt"""
import glob
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fly_bproject_amd.ppo import Net  # noqa: E402  (same keys and shapes as the reference's Net: tests/test_checkpoint_cpu.py)

files = sorted(glob.glob("/root/reference/saves/save9_1_23/*.pth"))
files = [files[0], files[len(files) // 2], files[-1]] + sorted(glob.glob("/root/reference/saves/save8_bigGrav/*.pth"))[-1:]
x = torch.randn(4096, 73, generator=torch.Generator().manual_seed(0))
out = []
for f in files:
    sd = torch.load(f, map_location="cpu", weights_only=True)
    net = Net(73, 18)
    net.load_state_dict(sd)
    with torch.no_grad():
        mu, v = net.pi(x), net.v(x)
    rec = {"file": os.path.basename(os.path.dirname(f)) + "/" + os.path.basename(f), "layers": {},
           "mu_abs_mean": round(float(mu.abs().mean()), 4), "mu_abs_max": round(float(mu.abs().max()), 3),
           "v_abs_mean": round(float(v.abs().mean()), 3), "v_abs_max": round(float(v.abs().max()), 2)}
    for k, t in sd.items():
        rec["layers"][k] = {"std": round(float(t.std()), 5), "max": round(float(t.abs().max()), 4), "mean": round(float(t.mean()), 5)}
    out.append(rec)
print(json.dumps(out, indent=1))
