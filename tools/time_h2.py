#!/usr/bin/env python3
"""A/B of the fp16x2 optimizer step for the library in $FLYHIP_LIB: gradient (fused launch + reduction) and gradient + Adam, HIP events,
alternating with nothing else; prints one line.   FLYHIP_LIB=build_ab/libflyhip_X.so python tools/time_h2.py [rows] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import _time_launches  # noqa: E402
from fly_bproject_amd.policy import PackedPolicy  # noqa: E402
from fly_bproject_amd.ppo import Net  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
torch.manual_seed(0)
net = Net(73, 18).to("cuda:0")
pol = PackedPolicy(net, "cuda:0")
pol.init_training(rows)
pol.gemm = "f16x2"
x = torch.randn(rows, 73, device="cuda:0")
act = torch.rand(rows, 18, device="cuda:0") * 2 - 1
olp = torch.randn(rows, device="cuda:0") - 20
adv = torch.randn(rows, device="cuda:0"); tgt = torch.randn(rows, device="cuda:0")
var = torch.full((18,), 0.2, device="cuda:0")
pol.calibrate_h2(x, act, olp, adv, tgt, var, 0.2)
out = []
for _ in range(3):
    t = _time_launches(lambda: pol.minibatch_grad(x, act, olp, adv, tgt, var, 0.2, fuse_norm=True), reps)
    t2 = _time_launches(lambda: (pol.minibatch_grad(x, act, olp, adv, tgt, var, 0.2, fuse_norm=True), pol.adam_step(norm_ready=True)), reps)
    out.append("%.1f / %.1f" % (t * 1e6, t2 * 1e6))
assert int(pol.h2_overflow) == 0
print("%-34s gradient / gradient + adam (us): %s" % (os.path.basename(os.environ.get("FLYHIP_LIB", "in-tree")), "   ".join(out)))
