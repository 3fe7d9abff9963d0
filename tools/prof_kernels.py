#!/usr/bin/env python3
"""Launch each hot kernel a fixed number of times at the bench sizes (for rocprofv3 kernel-trace /
PMC passes).  Usage on the GPU box:
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/prof_kernels.py
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/prof_kernels.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import RolloutAllHarness, make_args  # noqa: E402
from fly_bproject_amd import _lib  # noqa: E402
from fly_bproject_amd.fly import Fly  # noqa: E402
from fly_bproject_amd.policy import PackedPolicy  # noqa: E402
from fly_bproject_amd.ppo import Net  # noqa: E402

REPS = int(os.environ.get("PROF_REPS", "20"))
N = int(os.environ.get("PROF_ENVS", "8192"))
lib = _lib.load()
env = Fly(make_args(N))
a = torch.zeros(N, 18, device="cuda:0").uniform_(-1, 1)
for _ in range(REPS):
    env.step(a)
rows = (40960 // N) * N
net = Net(73, 18).to("cuda:0")
pol = PackedPolicy(net, "cuda:0")
pol.init_training(rows)
x = torch.randn(rows, 73, device="cuda:0")
act = torch.rand(rows, 18, device="cuda:0") * 2 - 1
olp = torch.randn(rows, device="cuda:0") - 20
adv = torch.randn(rows, device="cuda:0")
tgt = torch.randn(rows, device="cuda:0")
var = torch.full((18,), 0.2, device="cuda:0")
if pol.h2_live():          # the default optimizer step (fp16x2): its scales are measured on the data first, as PPO._update_hip does
    pol.calibrate_h2(x, act, olp, adv, tgt, var, 0.2)
for _ in range(REPS):
    pol.minibatch_grad(x, act, olp, adv, tgt, var, 0.2)
    pol.adam_step()
if pol.h2_live():
    assert int(pol.h2_overflow.item()) == 0
xs = torch.randn(N, 73, device="cuda:0")
eps = torch.randn(N, 18, device="cuda:0")
a_o = torch.empty(N, 18, device="cuda:0"); lp_o = torch.empty(N, device="cuda:0"); v_o = torch.empty(N, device="cuda:0")
pp = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
for _ in range(REPS):      # the loop's one-launch env step
    lib.ppo_rollout_step(env._handle, C.byref(env._bufs), pp(pol.P), pp(pol.PF), pp(xs), pp(eps), pp(var), 0, 0.0, 0.0,
                         pp(a_o), pp(lp_o), pp(v_o), pol.infer_pb_ptr(), None, None)
with torch.no_grad():
    for _ in range(REPS):
        pol.forward(xs, want_mu=True, want_v=False)
torch.cuda.synchronize()
# the loop's rollout: ONE launch for the T steps of every tile (rollout_all_fs_kernel at 8192 envs; PROF_T steps per launch)
T_ROLL = int(os.environ.get("PROF_T", str(16 * (40960 // N))))
harness = RolloutAllHarness(env, pol, T_ROLL, var)
for _ in range(max(2, REPS // 4)):
    harness.launch()
torch.cuda.synchronize()
del harness
env.exit()
if os.environ.get("PROF_DQN", "1") != "0":          # configs[4]: DQN updates of PROF_DQN_MB sampled replay steps x 32768 rows
    from fly_bproject_amd.dqn import DQN
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        mb = int(os.environ.get("PROF_DQN_MB", "4"))        # sampled steps per update (summarize_pmc.py reads the same variable)
        agent = DQN(make_args(32768, dqn_mini_batch_size=mb, replay_steps=2 * mb))
        for _ in range(mb + 2):
            agent.run()
        for gemm in ("f16x2", "bf16x3"):                    # the default's kernels and the A/B's
            agent.update_gemm = gemm
            for _ in range(REPS // 4 + 1):
                agent.update()
    torch.cuda.synchronize()
    agent.exit()
print("done")
