#!/usr/bin/env python3
"""Time one optimizer step's gradient at 40 960 rows: the fused step (mlp_fused_grad = persistent launch + slab reduction)
against the three-launch bf16x3 path (mlp_forward_backward + mlp_grad_w + reduce) and the fp32 path, HIP events, warm."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import _time_launches  # noqa: E402
from fly_bproject_amd.policy import PackedPolicy  # noqa: E402
from fly_bproject_amd.ppo import Net  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
net = Net(73, 18).to("cuda:0")
pol = PackedPolicy(net, "cuda:0")
pol.init_training(rows)
x = torch.randn(rows, 73, device="cuda:0")
act = torch.rand(rows, 18, device="cuda:0") * 2 - 1
olp = torch.randn(rows, device="cuda:0") - 20
adv = torch.randn(rows, device="cuda:0"); tgt = torch.randn(rows, device="cuda:0")
var = torch.full((18,), 0.2, device="cuda:0")
for gemm, fused in (("f16x2", True), ("bf16x3", True), ("f16x2", True), ("bf16x3", True), ("bf16x3", False), ("f32", False)):
    pol.gemm = gemm
    pol.step_gemm = "f16x2" if gemm == "f16x2" else "bf16x3"
    pol.fused_step = fused
    if gemm == "f16x2":
        print("fp16x2 scales settled in %d launches" % pol.calibrate_h2(x, act, olp, adv, tgt, var, 0.2))
    t = _time_launches(lambda: pol.minibatch_grad(x, act, olp, adv, tgt, var, 0.2, fuse_norm=True), reps)
    t2 = _time_launches(lambda: (pol.minibatch_grad(x, act, olp, adv, tgt, var, 0.2, fuse_norm=True), pol.adam_step(norm_ready=True)), reps)
    print("%-7s fused_step=%-5s gradient %.1f us   gradient + adam %.1f us   (%s)" % (gemm, fused, t * 1e6, t2 * 1e6, pol.update_path()))
    if gemm == "f16x2":
        assert int(pol.h2_overflow) == 0
