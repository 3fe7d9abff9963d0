#!/usr/bin/env python3
"""Per-workgroup timeline of mlp_forward_backward at the update size: where do the CU's three
workgroup slots spend the launch?  (diagnostic instantiation with s_memrealtime stamps)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from fly_bproject_amd import _lib  # noqa: E402
from fly_bproject_amd.policy import PackedPolicy  # noqa: E402
from fly_bproject_amd.ppo import Net  # noqa: E402

rows = int(os.environ.get("ROWS", "40960"))
lib = _lib.load()
net = Net(73, 18).to("cuda:0")
pol = PackedPolicy(net, "cuda:0")
pol.init_training(rows)
x = torch.randn(rows, 73, device="cuda:0")
act = torch.rand(rows, 18, device="cuda:0") * 2 - 1
olp = torch.randn(rows, device="cuda:0") - 20
adv = torch.randn(rows, device="cuda:0")
tgt = torch.randn(rows, device="cuda:0")
var = torch.full((18,), 0.2, device="cuda:0")
tiles = (rows + 31) // 32
pad = (tiles + 7) & ~7
stamps = torch.zeros((pad + tiles) * 4, dtype=torch.int64, device="cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
s, d = pol.saves, pol.dz
fn = lib.flyhip_debug_mlp_fwd_bwd_stamped
fn.argtypes = [C.c_void_p] * 4 + [C.c_int64] + [C.c_void_p] * 9 + [C.c_float, C.c_float] + [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 3
for ep in range(1, 4):
    fn(p(pol.P), p(pol.PF), p(pol.PT), p(x), rows, p(s["out"]), p(s["h1"]), p(s["h2"]), p(s["h3"]), p(act), p(olp), p(adv),
       p(tgt), p(var), 1.0 / rows, 0.2, p(d["dz4"]), p(d["dz3"]), p(d["dz2"]), p(d["dz1"]), p(pol.loss_part),
       p(pol._tile_flags), 100 + ep, p(pol.tile_wait_error), p(stamps), None)
torch.cuda.synchronize()
raw = stamps.cpu().numpy().reshape(-1, 4)
kind = np.array([0] * pad + [1] * tiles)
valid = raw[:, 1] != 0
raw, kind = raw[valid], kind[valid]
cu = (raw[:, 0] >> 48) & 0xfff
t0 = (raw[:, 0] & ((1 << 48) - 1)).min()
start = ((raw[:, 0] & ((1 << 48) - 1)) - t0) / 100.0
end = ((raw[:, 1] & ((1 << 48) - 1)) - t0) / 100.0
go = np.where(kind == 1, ((raw[:, 2] & ((1 << 48) - 1)) - t0) / 100.0, start)
print("err", int(pol.tile_wait_error.item()), "workgroups", len(raw), "launch span %.1f us" % end.max())
for k, nm in ((0, "forward"), (1, "backward")):
    m = kind == k
    print("%-8s n %4d  duration mean %.1f us (p10 %.1f p90 %.1f)  start mean %.1f  end max %.1f  flag wait mean %.2f us max %.2f" %
          (nm, m.sum(), (end - start)[m].mean(), *np.percentile((end - start)[m], [10, 90]), start[m].mean(), end[m].max(),
           (go - start)[m].mean(), (go - start)[m].max()))
# slot occupancy per CU over time
grid = np.arange(0, end.max(), 0.5)
occ = np.zeros_like(grid)
for a, b in zip(start, end):
    occ += (grid >= a) & (grid < b)
occ /= len(np.unique(cu))
print("mean resident workgroups per CU over the launch: %.2f;  by 10 us window: %s" %
      (occ.mean(), " ".join("%.2f" % occ[(grid >= w) & (grid < w + 10)].mean() for w in range(0, int(end.max()), 10))))
busy = (end - start).sum() / len(np.unique(cu))
print("sum of workgroup durations per CU %.1f us over 3 slots = %.1f us if gap-free; span %.1f us" % (busy, busy / 3, end.max()))
for c in np.unique(cu)[:3]:
    m = cu == c
    o = np.argsort(start[m])
    print("CU %03x:" % c, " ".join("%s[%.0f-%.0f]" % ("fb"[k], a, b) for k, a, b in zip(kind[m][o], start[m][o], end[m][o])))
