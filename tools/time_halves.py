#!/usr/bin/env python3
"""Does splitting a minibatch's forward/backward + dW into row halves (so that what one launch writes is still in the
256 MB Infinity Cache when the next reads it) pay?  Times minibatch_grad on 40960 rows against 2 x 20480 and 4 x 10240
(gradient accumulation ignored: timing only).  GEMM=bf16x3 selects that arithmetic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fly_bproject_amd.policy import PackedPolicy  # noqa: E402
from fly_bproject_amd.ppo import Net  # noqa: E402

rows, reps = 40960, 100
net = Net(73, 18).to("cuda:0")
pol = PackedPolicy(net, "cuda:0")
pol.init_training(rows)
pol.gemm = os.environ.get("GEMM", "f32")
x = torch.randn(rows, 73, device="cuda:0"); act = torch.rand(rows, 18, device="cuda:0") * 2 - 1
olp = torch.randn(rows, device="cuda:0") - 20; adv = torch.randn(rows, device="cuda:0"); tgt = torch.randn(rows, device="cuda:0")
var = torch.full((18,), 0.2, device="cuda:0")


def run(parts):
    n = rows // parts
    for i in range(parts):
        sl = slice(i * n, (i + 1) * n)
        pol.minibatch_grad(x[sl], act[sl], olp[sl], adv[sl], tgt[sl], var, 0.2)


for rnd in range(2):
    for parts in (1, 2, 4):
        for _ in range(10):
            run(parts)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps):
            run(parts)
        e1.record(); torch.cuda.synchronize()
        print("%s  %d part(s): %.1f us per 40960 rows (fused + dW + reduce)" % (pol.gemm, parts, e0.elapsed_time(e1) * 1e3 / reps), flush=True)
