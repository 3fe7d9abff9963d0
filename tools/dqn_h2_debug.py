#!/usr/bin/env python3
"""Where does the fp16x2 DQN update's gradient differ from float64?  Per parameter tensor and, for layer 1, per 32-column tile of the
hidden layer; with the lagged scales as calibrated and with single classes moved by 2^k (frozen).  dqn_h2_debug.py [parts] [n] [unambiguous]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tests.test_dqn import _bare_dqn  # noqa: E402
from tests.test_dqn_h2_gpu import _batch, _grad64, _perturb_target, _restore, _state, _unambiguous, _views  # noqa: E402

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
torch.manual_seed(2)
chunks = _batch(parts, n, 21)
d = _bare_dqn(rows=n, fused=True, gemm="f16x2")
_perturb_target(d)
st = _state(d)
if len(sys.argv) > 3:           # any third argument: rows with a hidden unit within 1e-5 of LeakyReLU's kink replaced (the tests' batches)
    chunks, replaced = _unambiguous(d, chunks)
    print("rows replaced:", replaced)
want, loss64 = _grad64(d, chunks)


def report(tag):
    errs = [float((g.double() - w).abs().max() / w.abs().max()) for g, w in zip(_views(d.packed.G), want)]
    print("%-34s %s" % (tag, " ".join("%.2e" % e for e in errs)), "refused", d.h2_overflows)
    g1, w1 = _views(d.packed.G)[0].double(), want[0]
    gb, wb = _views(d.packed.G)[1].double(), want[1]
    per = [(float((g1[32 * t:32 * t + 32] - w1[32 * t:32 * t + 32]).abs().max() / w1.abs().max()),
            float((gb[32 * t:32 * t + 32] - wb[32 * t:32 * t + 32]).abs().max() / wb.abs().max())) for t in range(8)]
    print("    dW1 / db1 per n tile:", " ".join("%.1e/%.1e" % p for p in per))
    perk = [float((g1[:, 32 * k:32 * k + 32] - w1[:, 32 * k:32 * k + 32]).abs().max() / w1.abs().max()) for k in range(3)]
    print("    dW1 per k tile:", " ".join("%.1e" % p for p in perk))


print("columns: dW1 db1 dW2 db2 dW3 db3 (max |g - g64| / max |g64|)")
d.update(chunks); torch.cuda.synchronize(); report("first update (calibrates)")
print("    scales:", d.packed.h2_scales[:14].cpu().numpy(), "\n    maxima:", d.packed.h2_scales[32:46].cpu().numpy())
_restore(d, st)
d.update(chunks); torch.cuda.synchronize(); report("second update, lagged scales")
print("    maxima:", d.packed.h2_scales[32:46].cpu().numpy())
_restore(d, st)
d.h2_freeze = True
base = d.packed.h2_scales.clone()
for cls, name in ((6, "dZ1"), (5, "dZ2"), (1, "H1"), (0, "X")):
    for k in (-4, 4, 8):
        d.packed.h2_scales.copy_(base)
        d.packed.h2_scales[cls] *= 2.0 ** k
        d.packed.h2_scales[16 + cls] /= 2.0 ** k
        d.update(chunks); torch.cuda.synchronize(); report("%s scale x 2^%d" % (name, k))
        _restore(d, st)
d.packed.h2_scales.copy_(base)
for gemm in ("bf16x3",):
    d.update_gemm = gemm
    d.update(chunks); torch.cuda.synchronize(); report(gemm)
    _restore(d, st)
