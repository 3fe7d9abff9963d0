#!/usr/bin/env python3
"""Time mlp_grad_w (+reduce) alone at the update size (FLYHIP_GW_SPLIT overrides the workgroup split)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402
import torch  # noqa: E402

from fly_bproject_amd import _lib  # noqa: E402
from fly_bproject_amd.policy import PackedPolicy  # noqa: E402
from fly_bproject_amd.ppo import Net  # noqa: E402

rows, reps = 40960, int(os.environ.get("REPS", "200"))
lib = _lib.load()
pol = PackedPolicy(Net(73, 18).to("cuda:0"), "cuda:0")
pol.init_training(rows)
x = torch.randn(rows, 73, device="cuda:0")
for t in list(pol.saves.values()) + list(pol.dz.values()):
    t.normal_()
p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
s, d = pol.saves, pol.dz
fn = lambda: lib.mlp_grad_w(p(x), p(s["h1"]), p(s["h2"]), p(s["h3"]), p(d["dz1"]), p(d["dz2"]), p(d["dz3"]), p(d["dz4"]), rows,   # noqa: E731
                            p(pol.workspace), p(pol.G), None, None, None, None, int(os.environ.get('B3', '0')), None)
for _ in range(5):
    fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(reps):
    fn()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("FLYHIP_GW_SPLIT", "default"), "dW + reduce: %.1f us" % (e0.elapsed_time(e1) * 1e3 / reps))
