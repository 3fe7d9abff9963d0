#!/usr/bin/env python3
"""One-launch forward+backward against the two launches for awkward row counts (1 .. 40992, ragged last tiles) in
both GEMM modes: bit-identical gradients / saved tensors, finite results, no lost tile flag."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fly_bproject_amd.policy import PackedPolicy, untile
from fly_bproject_amd.ppo import Net
torch.manual_seed(0)
net = Net(73, 18).to("cuda:0")
pol = PackedPolicy(net, "cuda:0")
pol.init_training(40992)
ok = True
for mode in ("f32", "bf16x3"):
    pol.gemm = mode
    for n in (1, 2, 31, 32, 33, 63, 64, 65, 255, 257, 2047, 4097, 8191, 40959, 40961, 40992):
        g = torch.Generator(device="cuda:0").manual_seed(n)
        x = torch.randn(n, 73, device="cuda:0", generator=g)
        act = (torch.rand(n, 18, device="cuda:0", generator=g) * 2 - 1)
        olp = torch.randn(n, device="cuda:0", generator=g) - 18
        adv = torch.randn(n, device="cuda:0", generator=g); tgt = torch.randn(n, device="cuda:0", generator=g)
        var = torch.full((18,), 0.2, device="cuda:0")
        res = []
        for fuse in (False, True):
            pol.fuse_fwd_bwd = fuse
            pol.G.fill_(float("nan"))
            pol.minibatch_grad(x, act, olp, adv, tgt, var, 0.2)
            torch.cuda.synchronize()
            res.append((pol.G.clone(), untile(pol.dz["dz1"], n, 256), untile(pol.saves["h1"], n, 256), float(pol.loss_value(n))))
        same = all(torch.equal(a, b) for a, b in zip(res[0][:3], res[1][:3])) and res[0][3] == res[1][3]
        fin = bool(torch.isfinite(res[1][0]).all())
        if not (same and fin):
            ok = False
        print(mode, n, "same" if same else "DIFF", "finite" if fin else "NONFINITE", "err", int(pol.tile_wait_error.item()))
print("ALL OK" if ok else "FAILURES")
