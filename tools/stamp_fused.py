#!/usr/bin/env python3
"""Per-phase timeline of mlp_fused_step_kernel (diagnostic STAMP instantiation): s_memtime of wave 0 at the phase boundaries of
every tile, averaged over workgroups and tiles, in shader cycles and as a share of the tile."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from fly_bproject_amd import _lib  # noqa: E402
from fly_bproject_amd.policy import PackedPolicy  # noqa: E402
from fly_bproject_amd.ppo import Net  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
net = Net(73, 18).to("cuda:0")
pol = PackedPolicy(net, "cuda:0")
pol.init_training(rows)
pol.gemm = "bf16x3"
H2 = len(sys.argv) > 2 and sys.argv[2] == "f16x2"       # stamp_fused.py ROWS f16x2: the fp16x2 kernel (same stamp slots)
if H2:
    pol.step_gemm = "f16x2"
x = torch.randn(rows, 73, device="cuda:0")
act = torch.rand(rows, 18, device="cuda:0") * 2 - 1
olp = torch.randn(rows, device="cuda:0") - 20
adv = torch.randn(rows, device="cuda:0"); tgt = torch.randn(rows, device="cuda:0")
var = torch.full((18,), 0.2, device="cuda:0")
if H2:
    pol.calibrate_h2(x, act, olp, adv, tgt, var, 0.2)
for _ in range(30):
    pol.minibatch_grad(x, act, olp, adv, tgt, var, 0.2)
torch.cuda.synchronize()
grid = 256
SLOTS = 32
stamps = torch.zeros(grid * 64 * SLOTS, dtype=torch.int64, device="cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
arr = (C.c_void_p * 8)(stamps.data_ptr(), None, None, None, None, None, None, None)
ws = torch.empty(int(max(_lib.load().mlp_fused_workspace_floats(), _lib.load().mlp_fused_h2_workspace_floats())), device="cuda:0")
for _ in range(3):
    if H2:
        _lib.check(_lib.load().mlp_fused_grad_h2(p(pol.P), p(pol.PH), p(pol.PTH), p(pol.h2_scales), p(pol.h2_overflow), 1, p(x), rows, p(act),
                                                 p(olp), p(adv), p(tgt), p(var), C.c_float(1.0 / rows), C.c_float(0.2), p(ws), p(pol.G),
                                                 None, None, None, p(pol.loss_part), arr, None), "stamp")
    else:
        _lib.check(_lib.load().mlp_fused_grad(p(pol.P), p(pol.PB), p(pol.PTB), p(x), rows, p(act), p(olp), p(adv), p(tgt), p(var),
                                              C.c_float(1.0 / rows), C.c_float(0.2), p(ws), p(pol.G), None, None, None, p(pol.loss_part),
                                              arr, None), "stamp")
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(grid, 64, SLOTS).astype(np.int64)
tiles = (rows + 31) // 32
per = (tiles + grid - 1) // grid
names = ["P0 x_store", "P1 L1 (2 tiles) work", "P1 barrier wait", "P2 L2", "P3 L3", "P4 L4 split-K", "P5 loss", "P6 dA3+dW4+dz3",
         "P7 dA2+dW3+dz2", "P8 dA1+dW2+dW1 work", "P8 end barrier"]
d = []
for b in range(grid):
    for t in range(per):
        if s[b, t, 11] > 0:
            d.append(np.diff(s[b, t, :12]))
d = np.array(d)
tot = d.sum(1).mean()
print("tiles stamped: %d   cycles per tile (wave 0): mean %.0f  min %d  max %d" % (len(d), tot, d.sum(1).min(), d.sum(1).max()))
for i, nm in enumerate(names):
    print("  %-26s %8.0f cycles  %5.1f %%" % (nm, d[:, i].mean(), 100 * d[:, i].mean() / tot))
gaps = np.array([s[b, t + 1, 0] - s[b, t, 11] for b in range(grid) for t in range(per - 1) if s[b, t + 1, 11] > 0])
print("between tiles (stamp 11 of tile i -> stamp 0 of tile i+1): mean %.0f cycles" % gaps.mean())
print("loop entry -> first tile's stamp 0: mean %.0f; last tile's stamp 11 -> loop exit: mean %.0f" %
      ((s[:, 0, 0] - s[:, 62, 2]).mean(), (s[:, 63, 0] - s[:, per - 1, 11]).mean()))
sub = np.array([[s[b, t, 12] - s[b, t, 9], s[b, t, 13] - s[b, t, 12], s[b, t, 14] - s[b, t, 13], s[b, t, 15] - s[b, t, 14], s[b, t, 10] - s[b, t, 15]]
                for b in range(grid) for t in range(per) if s[b, t, 11] > 0])
print("inside P8: tile 0: dA1 GEMM (48 MFMA) %.0f | x DMA issue + dW2 with the dZ1 epilogue in its gaps (48) %.0f | dW1 tile 0 (36) %.0f | tile 1: GEMM + dW2/epilogue %.0f | dW1 tile 1 %.0f"
      % tuple(sub.mean(0)))
fine = np.array([[s[b, t, 18] - s[b, t, 12], s[b, t, 17] - s[b, t, 18], s[b, t, 13] - s[b, t, 17], s[b, t, 16] - s[b, t, 14], s[b, t, 15] - s[b, t, 16]]
                 for b in range(grid) for t in range(per) if s[b, t, 11] > 0])
print("   column tile 0: x DMA issue + H1/dZ2 fragment reads + dW2 c = 0..3 %.0f | c = 4..7 %.0f | fragment assembly, db2, drain, head of the second GEMM %.0f || column tile 1: GEMM %.0f | dW2 + epilogue %.0f"
      % tuple(fine.mean(0)))
if len(sys.argv) > 2 and sys.argv[2] == "f16x2":      # finer inside column tile 0's dW2 (the fp16x2 instantiation stamps them)
    ff = np.array([[s[b, t, 19] - s[b, t, 12], s[b, t, 20] - s[b, t, 19], s[b, t, 21] - s[b, t, 20], s[b, t, 22] - s[b, t, 21], s[b, t, 23] - s[b, t, 22],
                    s[b, t, 18] - s[b, t, 23]] for b in range(grid) for t in range(per) if s[b, t, 11] > 0 and s[b, t, 19] > 0])
    if len(ff):
        print("   column tile 0, finer: x DMA issue %.0f | fragment reads issued %.0f | c = 0: its three products %.0f | c = 0: db2 products + settle %.0f | c = 1 %.0f | c = 2, 3 %.0f"
              % tuple(ff.mean(0)))
tail = s[:, 63, 1] - s[:, 63, 0]
print("epilogue (slab write + bias sums) per workgroup: mean %.0f cycles" % tail.mean())
pro = s[:, 62, 2] - s[:, 62, 0]
print("prologue (accumulator clear, X clear, biases, first x load) per workgroup: mean %.0f cycles" % pro.mean())
whole = s[:, 63, 1] - s[:, 62, 0]
rt0, rt1 = s[:, 62, 1] & 0xffffffffffff, s[:, 63, 2] & 0xffffffffffff
real_us = (rt1 - rt0) / 100.0
print("kernel entry -> slab written per workgroup: mean %.0f cycles = %.1f us real (min %.1f, max %.1f) -> in-kernel clock %.2f GHz"
      % (whole.mean(), real_us.mean(), real_us.min(), real_us.max(), (whole / real_us).mean() / 1e3))
print("first workgroup start -> last workgroup end (100 MHz real-time counter): %.1f us" % ((rt1.max() - rt0.min()) / 100.0))
