#!/usr/bin/env python3
"""Run the fused DQN gradient repeatedly on the same inputs and report which blocks of the packed gradient / which tiles' loss sums
differ between runs (a race would show here): debug_dqn_determinism.py [rows] [chunks] [runs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tests.test_dqn import _bare_dqn  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
torch.manual_seed(3)
d = _bare_dqn(rows=n, fused=True)
with torch.no_grad():
    for p_ in d.q_target.parameters():
        p_.add_(0.05 * torch.randn_like(p_))
d.packed.refresh()
chunks = [(torch.randn(n, 73, device="cuda:0"), torch.rand(n, device="cuda:0") * 2 - 1, torch.randn(n, device="cuda:0") * 2,
           torch.randn(n, 73, device="cuda:0"), (torch.rand(n, device="cuda:0") > 0.1).float()) for _ in range(parts)]
blocks = {"W1": (0, 20480), "b1": (20480, 20736), "W2": (20736, 86272), "b2": (86272, 86528), "W3": (86528, 94720), "b3": (94720, 94752)}
ref = None
for r in range(runs):
    loss_part = d._update_fused(chunks, 1.0 / (n * parts))
    torch.cuda.synchronize()
    G, lp = d.packed.G.clone(), loss_part.clone()
    ws = d._fu_ws.clone()
    if ref is None:
        ref = (G, lp, ws)
        continue
    msg = []
    for k, (a, b) in blocks.items():
        dd = (G[a:b] != ref[0][a:b]).sum().item()
        if dd:
            msg.append("%s: %d of %d differ (max %.3g)" % (k, dd, b - a, float((G[a:b] - ref[0][a:b]).abs().max())))
    bad = torch.nonzero(lp != ref[1]).view(-1)
    if bad.numel():
        msg.append("loss tiles differ: %d (first %s)" % (bad.numel(), bad[:8].tolist()))
    wd = torch.nonzero(ws != ref[2]).view(-1)
    if wd.numel():
        msg.append("slab floats differ: %d (first at %s)" % (wd.numel(), wd[:6].tolist()))
    print("run %d: %s" % (r, "; ".join(msg) if msg else "identical"))
    db2 = (G[86272:86528] != ref[0][86272:86528]).view(8, 32).sum(1).tolist()
    dW2 = (G[20736:86272] != ref[0][20736:86272]).view(8, 32, 256).sum((1, 2)).tolist()
    db1 = (G[20480:20736] != ref[0][20480:20736]).view(8, 32).sum(1).tolist()
    print("   differing entries per 32-column tile of dZ2: b2 %s, W2 rows %s; of dZ1: b1 %s" % (db2, dW2, db1))

# which part of the saved plane images differs between two runs: [tile][H1 | dZ2][plane][row][256 (swizzled)]
imgs = []
for r in range(2):
    d._update_fused(chunks, 1.0 / (n * parts))
    torch.cuda.synchronize()
    imgs.append(d._fu_images.clone().view(-1, 2, 3, 32, 256))
diff = (imgs[0] != imgs[1])
print("image words differing: H1 %d, dZ2 %d of %d each" % (diff[:, 0].sum().item(), diff[:, 1].sum().item(), diff[:, 0].numel()))
if diff[:, 1].any():
    idx = torch.nonzero(diff[:, 1])
    print("first differing dZ2 words (tile, plane, row, physical col):", idx[:12].tolist())
    r = idx[:, 2]; c = idx[:, 3]
    key = ((r & 3) << 2) | ((r >> 2) & 3)
    logical = (((c >> 3) ^ key) << 3) | (c & 7)
    print("logical column histogram by 32-column tile:", torch.bincount(logical >> 5, minlength=8).tolist())
    print("plane histogram:", torch.bincount(idx[:, 1], minlength=3).tolist(), "tiles affected:", idx[:, 0].unique().numel())
