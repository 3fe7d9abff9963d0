"""GPU (-m gpu): `mlp_fused_grad` -- the whole minibatch gradient (ppo.py:184-197) in ONE persistent launch with the
activations in LDS and dW accumulated in registers -- against the three-launch bf16x3 path it replaces (chain values bit
for bit), against fp64 (dW / db), and for run-to-run determinism.  The reference-golden and autograd checks of this kernel
are the `[bf16x3]` parametrisations of tests/test_mlp_train_gpu.py (the fused step is what `minibatch_grad` runs in that
mode)."""
import numpy as np
import pytest
import torch

from tests.test_mlp_train_gpu import _setup

pytestmark = pytest.mark.gpu
WIDTH = {"out": 32, "h1": 256, "h2": 128, "h3": 128, "dz4": 32, "dz3": 128, "dz2": 128, "dz1": 256}


def _chain(pol, n):
    from fly_bproject_amd.policy import untile
    return {k: untile((pol.saves if k in pol.saves else pol.dz)[k], n, w).clone() for k, w in WIDTH.items()}


def _poison(pol):
    for t in list(pol.saves.values()) + list(pol.dz.values()) + [pol.loss_part, pol.G]:
        t.fill_(float("nan"))


@pytest.mark.parametrize("n", [16, 33, 4099, 40960])
def test_fused_step_equals_three_launch_path(n):
    """Same operands, same MFMA order, same epilogue arithmetic: every value of the chain (h1, h2, h3, out, dz4 .. dz1) and the
    per-tile loss sums equal the three-launch bf16x3 path bit for bit; the gradient differs only in summation order."""
    net, ref, pol, (x, action, old_logp, adv, target, var) = _setup(n, 13, gemm="bf16x3")
    pol.fused_step = False
    _poison(pol)
    pol.minibatch_grad(x, action, old_logp, adv, target, var, 0.2)
    torch.cuda.synchronize()
    want, want_loss, want_G = _chain(pol, n), pol.loss_part[: (n + 31) // 32].clone(), pol.G.clone()
    pol.fused_step = True
    _poison(pol)
    pol.minibatch_grad(x, action, old_logp, adv, target, var, 0.2, dump=True)
    torch.cuda.synchronize()
    got, got_loss, got_G = _chain(pol, n), pol.loss_part[: (n + 31) // 32].clone(), pol.G.clone()
    for k in WIDTH:
        assert torch.isfinite(got[k]).all(), k
        assert torch.equal(got[k], want[k]), (k, float((got[k] - want[k]).abs().max()))
    assert torch.equal(got_loss, want_loss)
    assert torch.isfinite(got_G).all()
    m = pol.grad_mask > 0
    scale = float(want_G[m].abs().max())
    bad = torch.nonzero(((got_G - want_G).abs() > 2e-5 * scale + 1e-12) & m).view(-1)
    assert bad.numel() == 0, (bad[:8].tolist(), got_G[bad[:8]].tolist(), want_G[bad[:8]].tolist())


@pytest.mark.parametrize("n", [4099, 40960])
def test_fused_step_gradient_against_fp64(n):
    """dW = dZ^T A and db = colsum(dZ) of the fused launch against an fp64 evaluation on the chain values it dumped: inside the
    bar the separate dW kernels are held to (tests/test_mlp_train_gpu.py::test_grad_w_kernels_against_fp64)."""
    net, ref, pol, (x, action, old_logp, adv, target, var) = _setup(n, 31, gemm="bf16x3")
    pol.fused_step = True
    pol.minibatch_grad(x, action, old_logp, adv, target, var, 0.2, dump=True)
    torch.cuda.synchronize()
    c = _chain(pol, n)
    G = pol.G.clone()
    a = [x.double(), c["h1"].double(), c["h2"].double(), c["h3"].double()]
    dz = [c["dz1"].double(), c["dz2"].double(), c["dz3"].double(), c["dz4"].double()]
    views = [(G[:256 * 80].view(256, 80)[:, :73], G[20480:20736]), (G[20736:53504].view(128, 256), G[53504:53632]),
             (G[53632:70016].view(128, 128), G[70016:70144]), (G[70144:74240].view(32, 128), G[74240:74272])]
    worst = 0.0
    for (W, b), A, Z in zip(views, a, dz):
        W64, b64 = Z.T @ A, Z.sum(0)
        worst = max(worst, float((W.double() - W64).abs().max()) / (float(W64.abs().max()) + 1e-30),
                    float((b.double() - b64).abs().max()) / (float(b64.abs().max()) + 1e-30))
    assert worst <= 2e-5, worst
    # the padding columns of W1 (73..79) see x == 0: exactly zero, and element 76 (the "invalid gradient" mark) is 0
    assert torch.all(G[:256 * 80].view(256, 80)[:, 73:] == 0)


def test_fused_step_is_deterministic_and_grid_independent(monkeypatch):
    """Two launches on the same inputs leave the same gradient bit for bit; a different number of workgroups (7 instead of one
    per CU: every workgroup then walks ~183 tiles, the accumulators live through all of them) leaves the same chain values
    and a gradient equal up to summation order."""
    n = 40960 + 19
    net, ref, pol, (x, action, old_logp, adv, target, var) = _setup(n, 5, gemm="bf16x3")
    pol.fused_step = True
    pol.minibatch_grad(x, action, old_logp, adv, target, var, 0.2, dump=True)
    torch.cuda.synchronize()
    g1, c1 = pol.G.clone(), _chain(pol, n)
    _poison(pol)
    pol.minibatch_grad(x, action, old_logp, adv, target, var, 0.2, dump=True)
    torch.cuda.synchronize()
    assert torch.equal(pol.G, g1)
    import ctypes as C
    from fly_bproject_amd import _lib
    lib = _lib.load()
    lib.flyhip_debug_set_fused_grid.argtypes = [C.c_int]
    lib.flyhip_debug_set_fused_grid.restype = None
    lib.flyhip_debug_set_fused_grid(7)
    try:
        _poison(pol)
        pol.minibatch_grad(x, action, old_logp, adv, target, var, 0.2, dump=True)
        torch.cuda.synchronize()
    finally:
        lib.flyhip_debug_set_fused_grid(0)
    c7 = _chain(pol, n)
    for k in WIDTH:
        assert torch.equal(c7[k], c1[k]), k
    m = pol.grad_mask > 0
    assert float((pol.G[m] - g1[m]).abs().max()) <= 2e-5 * float(g1[m].abs().max())


def test_fused_step_whole_update_matches_three_launch_update():
    """One PPO iteration (75 optimizer steps at 4096 envs) through the fused step and through the three-launch path from the same
    seed: the two end far closer to each other than either moved (rounding of the gradient's summation order is all that differs)."""
    import contextlib
    import io
    from fly_bproject_amd.ppo import PPO
    from tests.hip_helpers import make_args
    out, init = {}, None
    for fused in (True, False):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            agent = PPO(make_args(4096))
            agent.policy.gemm = "bf16x3"
            agent.policy.fused_step = fused
            init = agent.policy.P.clone()
            for _ in range(agent.rollout_size):
                agent.run()
        torch.cuda.synchronize()
        assert agent.optim_step == 75 and int(agent.policy.step.item()) == 75
        assert ("mlp_fused_grad" in agent.policy.update_path()) == fused
        out[fused] = agent.policy.P.clone()
        agent.exit()
    assert torch.isfinite(out[True]).all()
    moved = float((out[False] - init).norm())
    apart = float((out[True] - out[False]).norm())
    assert moved > 0 and apart <= 0.15 * moved, (apart, moved)     # 75 Adam steps amplify summation-order rounding (cf. test_ppo_hip_and_torch_updates_agree)


@pytest.mark.parametrize("n,off", [(4099, 1), (40960, 3), (33, 2)])
def test_fused_step_on_a_misaligned_x(n, off):
    """x that is only 4-byte aligned (a minibatch slice of the rollout ring starts k * num_envs * 73 floats in: odd num_envs):
    the LDS-DMA of 16-byte pieces needs 16-byte alignment, so such an x takes the guarded loads -- chain values and the
    gradient equal those of the same rows at an aligned address bit for bit."""
    net, ref, pol, (x, action, old_logp, adv, target, var) = _setup(n, 17, gemm="bf16x3")
    pol.fused_step = True
    _poison(pol)
    pol.minibatch_grad(x, action, old_logp, adv, target, var, 0.2, dump=True)
    torch.cuda.synchronize()
    want, want_G, want_loss = _chain(pol, n), pol.G.clone(), pol.loss_part[: (n + 31) // 32].clone()
    buf = torch.full((n * 73 + 8,), float("nan"), device=x.device)
    xo = buf[off:off + n * 73].view(n, 73)
    xo.copy_(x)
    assert xo.data_ptr() % 16 != 0 and xo.is_contiguous()
    _poison(pol)
    pol.minibatch_grad(xo, action, old_logp, adv, target, var, 0.2, dump=True)
    torch.cuda.synchronize()
    got = _chain(pol, n)
    for k in WIDTH:
        assert torch.equal(got[k], want[k]), k
    assert torch.equal(pol.G, want_G) and torch.equal(pol.loss_part[: (n + 31) // 32], want_loss)


def test_ppo_with_an_odd_env_count_fused_vs_three_launch():
    """num_envs = 4095 -> mini_chunk 10: every second minibatch slice of the ring is 4-byte aligned only.  One PPO iteration
    through the fused step and through the three-launch path end close together (as test_fused_step_whole_update...)."""
    import contextlib
    import io
    from fly_bproject_amd.ppo import PPO
    from tests.hip_helpers import make_args
    out, init = {}, None
    for fused in (True, False):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            agent = PPO(make_args(4095))
            agent.policy.fused_step = fused
            init = agent.policy.P.clone()
            for _ in range(agent.rollout_size):
                agent.run()
        torch.cuda.synchronize()
        assert agent.optim_step == 75 and int(agent.policy.step.item()) == 75
        out[fused] = agent.policy.P.clone()
        agent.exit()
    assert torch.isfinite(out[True]).all()
    moved = float((out[False] - init).norm())
    apart = float((out[True] - out[False]).norm())
    assert moved > 0 and apart <= 0.15 * moved, (apart, moved)
