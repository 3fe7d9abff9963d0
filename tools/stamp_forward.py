#!/usr/bin/env python3
"""Where does a workgroup of mlp_forward_kernel spend its cycles?  Runs the stamped diagnostic
instantiation at the update size (40960 rows) and prints the mean cycles per phase (wave 0)."""
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
grid = (rows + 31) // 32
stamps = torch.zeros(grid * 16, dtype=torch.int64, device="cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
s = pol.saves
fn = lib.flyhip_debug_mlp_forward_stamped
fn.argtypes = [C.c_void_p] * 3 + [C.c_int64] + [C.c_void_p] * 6 + [C.c_int]
GRID = int(os.environ.get("GRID", "0"))        # > 0: persistent workgroups walking tiles b, b+GRID, ...
for _ in range(3):
    fn(p(pol.P), p(pol.PF), p(x), rows, p(s["out"]), p(s["h1"]), p(s["h2"]), p(s["h3"]), p(stamps), None, GRID)
torch.cuda.synchronize()
raw = stamps.cpu().numpy().reshape(grid, 16).astype(np.int64)
raw = raw[raw[:, 13] > 0]                      # workgroups that stamped (persistent grids are smaller)
cu = (raw[:, 14] >> 48) & 0xfff                # xcc(4) | se(3) sh(1) cu(4)
raw[:, 14] &= (1 << 48) - 1
grid = raw.shape[0]
st = raw[:, :14]
real = (raw[:, 15] - raw[:, 14]).astype(np.float64)          # 100 MHz ticks
clk = (st[:, 13] - st[:, 0]) / np.maximum(real, 1) * 100e6
print("in-kernel clock (s_memtime / s_memrealtime): median %.3f GHz, p10 %.3f, p90 %.3f" %
      (np.median(clk) / 1e9, np.percentile(clk, 10) / 1e9, np.percentile(clk, 90) / 1e9))
d = np.diff(st, axis=1)
names = ["x stage+barrier", "L1a gemm", "L1a epilogue", "barrier", "L2a + L1b gemms", "barrier + L1b epilogue", "barrier",
         "L2b gemm", "L2 epilogue + barrier", "L3 gemm", "L3 epilogue", "barrier", "L4 + out"]
tot = (st[:, 13] - st[:, 0])
print("workgroups", grid, "mean total cycles per WG (s_memtime ticks = shader cycles)", tot.mean(), "median", np.median(tot))
for i, nm in enumerate(names):
    print("%-22s mean %8.0f  median %8.0f  (%4.1f %%)" % (nm, d[:, i].mean(), np.median(d[:, i]), 100 * d[:, i].mean() / tot.mean()))
rt = raw[:, 15].max() - raw[:, 14].min()
print("first start -> last end (s_memrealtime, global): %.2f us" % (rt / 100.0))
t0 = raw[:, 14].min()
start = (raw[:, 14] - t0) / 100.0
end = (raw[:, 15] - t0) / 100.0
first = start < 0.5 * np.median(end - start)
print("first-round workgroups %d: mean duration %.2f us; later workgroups %d: mean duration %.2f us, start mean %.2f us (p10 %.2f, p90 %.2f)" %
      (first.sum(), (end - start)[first].mean(), (~first).sum(), (end - start)[~first].mean(), start[~first].mean(),
       np.percentile(start[~first], 10), np.percentile(start[~first], 90)))
ncu = len(np.unique(cu))
per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
print("distinct CUs seen %d; workgroups per CU: min %d max %d; histogram %s" % (ncu, per_cu.min(), per_cu.max(), np.bincount(per_cu).tolist()))
last_end = np.array([end[cu == c].max() for c in np.unique(cu)])
print("per-CU finish time: mean %.2f us, p10 %.2f, p90 %.2f, max %.2f" % (last_end.mean(), np.percentile(last_end, 10), np.percentile(last_end, 90), last_end.max()))
xcc = cu >> 8
print("workgroups per XCC:", np.bincount(xcc).tolist())
dur = end - start
print("first-round duration percentiles (us): p5 %.1f p25 %.1f p50 %.1f p75 %.1f p95 %.1f" % tuple(np.percentile(dur[first], [5, 25, 50, 75, 95])))
print("later-round duration percentiles (us): p5 %.1f p25 %.1f p50 %.1f p75 %.1f p95 %.1f" % tuple(np.percentile(dur[~first], [5, 25, 50, 75, 95])))
for c in np.unique(cu)[:6]:
    m = cu == c
    o = np.argsort(start[m])
    print("CU %03x:" % c, " ".join("[%.1f-%.1f]" % (a, b) for a, b in zip(start[m][o], end[m][o])))
