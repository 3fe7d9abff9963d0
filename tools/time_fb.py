#!/usr/bin/env python3
"""Time mlp_forward + mlp_backward_dx (two launches) against mlp_forward_backward (one launch) at the
update size, HIP events around back-to-back repetitions."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fly_bproject_amd.policy import PackedPolicy  # noqa: E402
from fly_bproject_amd.ppo import Net  # noqa: E402

rows = int(os.environ.get("ROWS", "40960"))
reps = int(os.environ.get("REPS", "100"))
net = Net(73, 18).to("cuda:0")
pol = PackedPolicy(net, "cuda:0")
pol.init_training(rows)
x = torch.randn(rows, 73, device="cuda:0")
act = torch.rand(rows, 18, device="cuda:0") * 2 - 1
olp = torch.randn(rows, device="cuda:0") - 20
adv = torch.randn(rows, device="cuda:0")
tgt = torch.randn(rows, device="cuda:0")
var = torch.full((18,), 0.2, device="cuda:0")
for fuse in (False, True, False, True):
    pol.fuse_fwd_bwd = fuse
    for _ in range(5):
        pol.minibatch_grad(x, act, olp, adv, tgt, var, 0.2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        pol.minibatch_grad(x, act, olp, adv, tgt, var, 0.2)
    e1.record()
    torch.cuda.synchronize()
    print("fused" if fuse else "two launches", "forward+backward+dW+reduce: %.1f us per minibatch" % (e0.elapsed_time(e1) * 1e3 / reps),
          "err", int(pol.tile_wait_error.item()))
