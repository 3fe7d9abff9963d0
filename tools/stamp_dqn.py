#!/usr/bin/env python3
"""Per-phase shader cycles of dqn_chain_kernel (the second tile of every workgroup, thread 0) and HIP-event times of one fused DQN
update at 32768 envs x S sampled steps: stamp_dqn.py [S] [f16x2 (default: dqn_chain_h2_kernel, coarse phases only) | bf16x3]"""
import contextlib
import ctypes as C
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import make_args  # noqa: E402
from fly_bproject_amd import _lib  # noqa: E402
from fly_bproject_amd.dqn import DQN  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
gemm = sys.argv[2] if len(sys.argv) > 2 else "f16x2"
n = 32768
with contextlib.redirect_stdout(io.StringIO()):
    agent = DQN(make_args(n, dqn_mini_batch_size=S, replay_steps=max(2 * S, 8), dqn_gemm=gemm))
    for _ in range(S + 2):
        agent.run()
torch.cuda.synchronize()
lib = _lib.load()
stamps = torch.zeros(256 * 64, dtype=torch.int64, device="cuda:0")
lib.flyhip_debug_set_dqn_stamps.argtypes = [C.c_void_p]
lib.flyhip_debug_set_dqn_stamps.restype = None
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.flyhip_debug_set_dqn_stamps(C.c_void_p(stamps.data_ptr()) if rep == 2 else None)
    e0.record()
    agent.update()
    e1.record()
    torch.cuda.synchronize()
    print("update of %d sampled steps x %d rows: %.2f ms (%.1f us per sampled step)" % (S, n, e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e3 / S))
lib.flyhip_debug_set_dqn_stamps(None)
s = stamps.cpu().numpy().reshape(256, 64)
names = ["target: input planes", "target L1", "target L2", "target L3 + partials", "online input planes + target max, barrier", "(online pass begins)",
         "online L1", "online L2", "online L3 + partials", "loss + dZ3", "dA2, dZ2, dW3, db", "dA1, dZ1, dW1", "image copy-out", ]
d = np.diff(s[:, :14], axis=1).astype(np.float64)
ok = (s[:, 13] > 0)
print("%s: workgroups stamped: %d; cycles per tile %.0f; updates refused: %d" % (gemm, ok.sum(), d[ok].sum(1).mean(), agent.h2_overflows))
for i, nm in enumerate(names):
    print("  %-30s %8.0f" % (nm, d[ok][:, i].mean()))
if gemm != "bf16x3":
    agent.exit()
    sys.exit(0)
fine = ["L1 GEMM a", "L1 GEMM b (+ L1 epilogue a)", "barrier", "L2 GEMM a (+ L1 epilogue b, barrier inside)", "L2 GEMM b (+ L2 epilogue a)",
        "head requests", "L3 GEMM (+ L2 epilogue b)", "partials + barrier"]
for ps, (base, c0) in enumerate(((16, 1), (32, 6))):
    t = s[ok].astype(np.float64)
    seq = [t[:, c0], t[:, base + 0], t[:, base + 1], t[:, c0 + 1], t[:, base + 3], t[:, base + 4], t[:, c0 + 2], t[:, base + 6], t[:, c0 + 3]]
    print("  inside the %s pass (wave 0):" % ("target", "online")[ps])
    for i, nm in enumerate(fine):
        print("    %-44s %8.0f" % (nm, (seq[i + 1] - seq[i]).mean()))
t = s[ok].astype(np.float64)
seq = [t[:, 11], t[:, 48], t[:, 49], t[:, 50], t[:, 51], t[:, 12]]
print("  inside dA1 / dZ1 / dW1 (wave 0):")
for i, nm in enumerate(["GEMM a (+ dZ2 epilogue b, barrier inside), db2 b", "GEMM b (+ dZ1 epilogue a)", "dZ2 image out", "dW1 a (+ dZ1 epilogue b)", "dW1 b"]):
    print("    %-50s %8.0f" % (nm, (seq[i + 1] - seq[i]).mean()))
agent.exit()
