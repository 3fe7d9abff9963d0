#!/usr/bin/env python3
"""A/B of the tile hand-off inside mlp_forward_backward: "sc1" (write-through stores + L1-bypassing loads)
against "xcd" (plain accesses, same-XCD placement), alternating, HIP events around back-to-back launches
of the fused kernel ALONE and of the whole minibatch (fused launch + dW + reduce)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fly_bproject_amd import _lib  # noqa: E402
from fly_bproject_amd.policy import PackedPolicy  # noqa: E402
from fly_bproject_amd.ppo import Net  # noqa: E402

rows = int(os.environ.get("ROWS", "40960"))
reps = int(os.environ.get("REPS", "200"))
lib = _lib.load()
net = Net(73, 18).to("cuda:0")
pol = PackedPolicy(net, "cuda:0")
pol.init_training(rows)
pol.gemm = os.environ.get("GEMM", "f32")
x = torch.randn(rows, 73, device="cuda:0")
act = torch.rand(rows, 18, device="cuda:0") * 2 - 1
olp = torch.randn(rows, device="cuda:0") - 20
adv = torch.randn(rows, device="cuda:0")
tgt = torch.randn(rows, device="cuda:0")
var = torch.full((18,), 0.2, device="cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
s, d = pol.saves, pol.dz


def fb(coh):
    pol._epoch += 1
    lib.mlp_forward_backward(p(pol.P), p(pol.PF), p(pol.PT), p(x), rows, p(s["out"]), p(s["h1"]), p(s["h2"]), p(s["h3"]),
                             p(act), p(olp), p(adv), p(tgt), p(var), 1.0 / rows, 0.2, p(d["dz4"]), p(d["dz3"]), p(d["dz2"]),
                             p(d["dz1"]), p(pol.loss_part), p(pol._tile_flags), pol._epoch, p(pol.tile_wait_error),
                             pol.pb_ptr(), pol.ptb_ptr(), coh, _lib.stream_ptr())


def timeit(fn):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for rnd in range(3):
    for mode, coh in (("sc1", 1), ("xcd", 0)):
        t = timeit(lambda: fb(coh))
        pol.handoff = mode
        t2 = timeit(lambda: pol.minibatch_grad(x, act, olp, adv, tgt, var, 0.2))
        print("round %d  %s: fused launch %.1f us   minibatch (fused + dW + reduce) %.1f us   err %d"
              % (rnd, mode, t, t2, int(pol.tile_wait_error.item())), flush=True)
