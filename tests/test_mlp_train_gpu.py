"""GPU (-m gpu): the MFMA training kernels (mlp_forward / mlp_backward_dx / mlp_grad_w /
mlp_adam_step, called through the C ABI) against torch fp32 autograd + torch.optim.Adam on the
same inputs, and against the reference's own 75-step PPO.update recorded in g7."""
import contextlib
import io
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bare_agent(net, var):
    from fly_bproject_amd.ppo import PPO
    p = PPO.__new__(PPO)
    p.net, p.action_var, p.clip = net, var, 0.2
    return p


# every arithmetic is held to autograd and to the reference's g7 golden at the SAME tolerances ("f16x2" = bf16x3 everywhere with the
# fused optimizer-step gradient in the two-term fp16 arithmetic, csrc/mlp_fused_h2.inc)
GEMMS = ["f32", "bf16x3", "f16x2"]


def _setup(n, seed, sd=None, gemm=None):
    from fly_bproject_amd.policy import PackedPolicy
    from fly_bproject_amd.ppo import Net, diag_gauss_logprob
    torch.manual_seed(seed)
    net = Net(73, 18).to(DEV)
    if sd is not None:
        net.load_state_dict(sd)
    ref = Net(73, 18).to(DEV)
    ref.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
    pol = PackedPolicy(net, DEV)
    pol.init_training(max(n, 32))
    pol.gemm = gemm if gemm is not None else "f32"      # tests that do not name an arithmetic were written for the fp32 MFMA kernels
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(n, 73, device=DEV, generator=g)
    var = torch.full((18,), 0.15, device=DEV)
    with torch.no_grad():
        mu = ref.to_mean(ref.shared_net(x))
        action = (mu + 0.4 * torch.randn(n, 18, device=DEV, generator=g)).clamp(-1, 1)
        old_logp = diag_gauss_logprob(mu, action, var) + 0.3 * torch.randn(n, device=DEV, generator=g)
    adv = torch.randn(n, device=DEV, generator=g)
    target = torch.randn(n, device=DEV, generator=g) * 1.5
    if gemm == "f16x2":             # the per-class scales of the fp16x2 step are measured on the data (PPO._update_hip does this too)
        pol.calibrate_h2(x, action, old_logp, adv, target, var, 0.2)
    return net, ref, pol, (x, action, old_logp, adv, target, var)


@pytest.mark.parametrize("gemm", GEMMS)
@pytest.mark.parametrize("n", [16, 4099, 40960])
def test_minibatch_gradient_matches_autograd(n, gemm):
    net, ref, pol, (x, action, old_logp, adv, target, var) = _setup(n, 3, gemm=gemm)
    pol.minibatch_grad(x, action, old_logp, adv, target, var, 0.2)
    loss_hip = float(pol.loss_value(n))
    agent = _bare_agent(ref, var)
    loss = agent.minibatch_loss(x, action, old_logp, target.unsqueeze(-1), adv.unsqueeze(-1))
    loss.backward()
    assert abs(loss_hip - float(loss)) <= 2e-5 * max(1.0, abs(float(loss)))
    # ratios must straddle the clip range for the test to mean anything
    G = pol.G
    for name, view in pol.views.items():
        idx = torch.arange(G.numel(), device=DEV).as_strided(view.shape, view.stride(), view.storage_offset())
        got = G[idx]
        want = dict(ref.named_parameters())[name].grad
        scale = float(want.abs().max()) + 1e-12
        err = float((got - want).abs().max())
        assert err <= 2e-4 * scale + 1e-9, (name, err, scale)
    # padding / structural zeros carry whatever the GEMM produced but are masked in the step:
    assert int(pol.grad_mask.sum()) == 69587


def test_adam_clip_step_matches_torch():
    net, ref, pol, batch = _setup(4096, 5)
    x, action, old_logp, adv, target, var = batch
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    agent = _bare_agent(ref, var)
    for it in range(3):
        pol.minibatch_grad(x, action, old_logp, adv * (50.0 if it == 1 else 1.0), target, var, 0.2)
        pol.adam_step()
        loss = agent.minibatch_loss(x, action, old_logp, target.unsqueeze(-1), (adv * (50.0 if it == 1 else 1.0)).unsqueeze(-1))
        opt.zero_grad()
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt.step()
        np.testing.assert_allclose(float(pol.grad_norm), float(gn), rtol=2e-4)
        # Adam's first steps move every weight by ~lr*sign(g): an element whose gradient is at rounding
        # level may legitimately land anywhere within +-lr per step, so the bar is: 99.9 % of the
        # elements within fp32 tolerance, every element within (it+1)*lr.
        for k, p in net.state_dict().items():
            q = ref.state_dict()[k]
            err = (p - q).abs()
            tight = err <= 2e-6 + 2e-4 * q.abs()
            assert float(tight.float().mean()) >= 0.999, (k, float(tight.float().mean()))
            assert float(err.max()) <= 1.05e-3 * (it + 1), (k, float(err.max()))
    assert int(pol.step) == 3
    # the fragment-ordered copies follow the master weights
    pf, pt = pol.PF.clone(), pol.PT.clone()
    pol.refresh()
    assert torch.equal(pf, pol.PF) and torch.equal(pt, pol.PT)
    # structural zeros stayed zero
    assert torch.all(pol.W1[:, 73:] == 0) and torch.all(pol.W4[19:] == 0) and torch.all(pol.W4[:18, 64:] == 0)
    assert torch.all(pol.W4[18, :64] == 0) and torch.all(pol.b4[19:] == 0)


@pytest.mark.parametrize("gemm", GEMMS)
def test_update_matches_reference_golden(golden, gemm):
    """75 optimizer steps of the reference's PPO.update (g7: T=32, N=8, mini_chunk 2) from the same
    initial weights end at the same weights (fp32 tolerance of the CPU-oracle test); the first minibatch's loss,
    gradient norm and every parameter's gradient equal the reference's recorded autograd values."""
    g = golden("g7_update")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)   # noqa: E731
    sd = {k[3:]: t(g[k]) for k in g.files if k.startswith("w0_")}
    net, ref, pol, _ = _setup(16, 0, sd=sd, gemm=gemm)
    obs, acts, logp = t(g["obs"]), t(g["acts"]), t(g["log_prob"])
    target, adv, var = t(g["target"]), t(g["adv"]), t(g["action_var"])
    T, N, mc = 32, 8, 2
    rows = mc * N
    steps = 0
    first = None
    if gemm == "f16x2":
        pol.calibrate_h2(obs[0:mc].reshape(rows, 73), acts[0:mc].reshape(rows, 18), logp[0:mc].reshape(rows), adv[0:mc].reshape(rows),
                         target[0:mc].reshape(rows), var, 0.2)
    for _ in range(5):
        k = 0
        for j in range(mc, T, mc):
            pol.minibatch_grad(obs[k:j].reshape(rows, 73), acts[k:j].reshape(rows, 18), logp[k:j].reshape(rows),
                               adv[k:j].reshape(rows), target[k:j].reshape(rows), var, 0.2)
            if first is None:
                first = (float(pol.loss_value(rows)), pol.G.clone())
            pol.adam_step()
            if steps == 0:
                np.testing.assert_allclose(float(pol.grad_norm), float(g["gradnorm0"]), rtol=2e-4)
            steps += 1
            k = j
    assert steps == 75 and int(pol.step) == 75 and int(pol.h2_overflow) == 0        # (no step of the fp16x2 run was refused)
    np.testing.assert_allclose(first[0], float(g["loss0"]), rtol=2e-5)
    coef = min(1.0, 1.0 / (float(g["gradnorm0"]) + 1e-6))      # g0_* were recorded after clip_grad_norm_ scaled them in place
    for name, view in pol.views.items():         # the reference's own gradients of the first minibatch (ppo.py:197-198)
        idx = torch.arange(first[1].numel(), device=DEV).as_strided(view.shape, view.stride(), view.storage_offset())
        want = g["g0_" + name]
        np.testing.assert_allclose(first[1][idx].cpu().numpy() * coef, want, rtol=0, atol=2e-4 * float(np.abs(want).max()) + 1e-9, err_msg=name)
    for k, p in net.state_dict().items():
        np.testing.assert_allclose(p.cpu().numpy(), g["w1_" + k], rtol=2e-3, atol=2e-4, err_msg=k)


def test_ppo_hip_and_torch_updates_agree():
    """Same rollout, one PPO.update through each backend: same parameters afterwards."""
    from fly_bproject_amd.ppo import PPO
    from tests.hip_helpers import make_args
    outs, init, fn = {}, None, {}
    for backend in ("hip", "torch"):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            agent = PPO(make_args(4096, update_backend=backend))
            agent.policy.gemm = "f32"
            init = {k: v.clone() for k, v in agent.net.state_dict().items()}
            with torch.no_grad():
                probe = torch.randn(512, 73, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
                fn["init"] = torch.cat([agent.net.pi(probe), agent.net.v(probe)], dim=1)
            for _ in range(agent.rollout_size):
                agent.run()
            with torch.no_grad():
                fn[backend] = torch.cat([agent.net.pi(probe), agent.net.v(probe)], dim=1)
        assert agent.optim_step == 75
        outs[backend] = {k: v.clone() for k, v in agent.net.state_dict().items()}
        agent.exit()
    # 75 Adam steps amplify rounding on elements whose gradient hovers around zero (each step moves a
    # weight by ~lr*sign(g)), so the bar is on the trajectory: the two backends end much closer to
    # each other than either moved from the start, in weight space and in function space.  The exact
    # checks are the gradient / Adam-step / 75-step golden tests above.
    for k in outs["hip"]:
        moved = float((outs["torch"][k] - init[k]).norm())
        apart = float((outs["hip"][k] - outs["torch"][k]).norm())
        assert moved > 0 and apart <= 0.3 * moved, (k, apart, moved)
    moved = float((fn["torch"] - fn["init"]).norm())
    apart = float((fn["hip"] - fn["torch"]).norm())
    assert apart <= 0.2 * moved, (apart, moved)


def test_fused_norm_step_matches_two_launch_step():
    """Single-rank fast path (clip-norm partials produced by the gradient reduce, one optimizer
    launch) vs the general path (separate norm launch): same norm, same parameters."""
    outs = []
    for mode in ("two_launch", "fused_norm", "self_norm"):
        net, ref, pol, (x, action, old_logp, adv, target, var) = _setup(4099, 11)
        for it in range(3):
            pol.minibatch_grad(x, action, old_logp, adv * (30.0 if it == 1 else 1.0), target, var, 0.2,
                               fuse_norm=mode == "fused_norm")
            # "self_norm": the data-parallel form -- ONE launch sums the (all-reduced) gradient itself, scales it by
            # 1 / world (here: the gradient doubled first, scale 0.5) and ping-pongs the step counter
            if mode == "self_norm":
                pol.G.mul_(2.0)
                pol.adam_step(grad_scale=0.5, self_norm=True)
            else:
                pol.adam_step(norm_ready=mode == "fused_norm")
        outs.append((pol.P.clone(), float(pol.grad_norm), int(pol.step), pol.PF.clone()))
    assert outs[0][2] == outs[1][2] == outs[2][2] == 3
    for o in outs[1:]:
        np.testing.assert_allclose(outs[0][1], o[1], rtol=1e-5)
        torch.testing.assert_close(outs[0][0], o[0], rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(outs[0][3], o[3], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n", [16, 4099, 40960])
def test_one_launch_forward_backward_equals_two_launches(n):
    """mlp_forward_backward (backward workgroups wait on per-tile flags of the forward workgroups of
    the same launch) leaves bit for bit what mlp_forward followed by mlp_backward_dx leave, call
    after call on the same flag words, and never reports a lost flag."""
    net, ref, pol, (x, action, old_logp, adv, target, var) = _setup(n, 11)
    res = {}
    assert pol.fuse_fwd_bwd, "the start-up probe found forward/backward workgroups on different XCDs"
    e0 = pol._epoch
    for fuse in (False, True, True):
        pol.fuse_fwd_bwd = fuse
        for t in list(pol.saves.values()) + list(pol.dz.values()) + [pol.loss_part, pol.G]:
            t.fill_(float("nan"))
        pol.minibatch_grad(x, action, old_logp, adv, target, var, 0.2)
        torch.cuda.synchronize()
        from fly_bproject_amd.policy import untile
        width = {"out": 32, "h1": 256, "h2": 128, "h3": 128, "dz4": 32, "dz3": 128, "dz2": 128, "dz1": 256}
        got = [untile(pol.saves[k], n, width[k]) for k in ("out", "h1", "h2", "h3")] + \
              [untile(pol.dz[k], n, width[k]) for k in ("dz4", "dz3", "dz2", "dz1")] + \
              [pol.loss_part[: (n + 31) // 32].clone(), pol.G.clone()]
        assert all(torch.isfinite(t).all() for t in got)
        if fuse:
            for a, b in zip(res[False], got):
                assert torch.equal(a, b)
        else:
            res[False] = got
    assert int(pol.tile_wait_error.item()) == 0 and pol._epoch == e0 + 2


@pytest.mark.parametrize("n", [4099, 40960])
def test_grad_w_kernels_against_fp64(n):
    """dW = dZ^T A of both arithmetics (fp32 MFMA chain; bf16x3 on the bf16 pipe, six terms into ONE fp32 accumulator
    per tile) against an fp64 evaluation on the same saved tensors: the bf16x3 kernel's error must stay within 2x of the
    fp32 kernel's (its licence to replace it) and both inside the suite's gradient tolerance."""
    from fly_bproject_amd.policy import untile
    net, ref, pol, (x, action, old_logp, adv, target, var) = _setup(n, 31)
    pol.gemm = "f32"
    pol.minibatch_grad(x, action, old_logp, adv, target, var, 0.2)
    torch.cuda.synchronize()
    g = {"f32": pol.G.clone()}
    width = {"h1": 256, "h2": 128, "h3": 128, "dz4": 32, "dz3": 128, "dz2": 128, "dz1": 256}
    a = [x.double()] + [untile(pol.saves[k], n, width[k]).double() for k in ("h1", "h2", "h3")]
    dz = [untile(pol.dz[k], n, width[k]).double() for k in ("dz1", "dz2", "dz3", "dz4")]
    # the same saved tensors through the bf16x3 dW kernel alone
    import ctypes as C
    from fly_bproject_amd import _lib
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    s_, d_ = pol.saves, pol.dz
    _lib.check(_lib.load().mlp_grad_w(p(x), p(s_["h1"]), p(s_["h2"]), p(s_["h3"]), p(d_["dz1"]), p(d_["dz2"]), p(d_["dz3"]), p(d_["dz4"]),
                                      C.c_int64(n), p(pol.workspace), p(pol.G), None, None, None, None, 1, None), "mlp_grad_w b3")
    torch.cuda.synchronize()
    g["bf16x3"] = pol.G.clone()
    views = lambda G: [(G[:256 * 80].view(256, 80)[:, :73], G[20480:20736]), (G[20736:53504].view(128, 256), G[53504:53632]),   # noqa: E731
                       (G[53632:70016].view(128, 128), G[70016:70144]), (G[70144:74240].view(32, 128), G[74240:74272])]
    err = {}
    for mode in g:
        worst = 0.0
        for (W, b), A, Z in zip(views(g[mode]), a, dz):
            W64, b64 = Z.T @ A, Z.sum(0)
            scale = float(W64.abs().max()) + 1e-30
            worst = max(worst, float((W.double() - W64).abs().max()) / scale, float((b.double() - b64).abs().max()) / (float(b64.abs().max()) + 1e-30))
        err[mode] = worst
    assert err["f32"] <= 2e-5 and err["bf16x3"] <= 2e-5, err
    assert err["bf16x3"] <= 2.0 * err["f32"] + 1e-7, err


def test_bf16x3_training_step_matches_fp32_path():
    """One minibatch gradient + Adam step with the bf16x3 GEMMs (forward, dX chain, dW) against the fp32-MFMA path and torch autograd: same loss and gradients to the suite's fp32
    tolerance, and the bf16 term planes the Adam kernel scatters equal a fresh split of the weights."""
    from fly_bproject_amd.policy import split_bf16x3, untile
    n = 4099
    net, ref, pol, (x, action, old_logp, adv, target, var) = _setup(n, 21)
    pol.fused_step = False            # this test is about the three-launch kernels (the fused step: tests/test_fused_step_gpu.py)
    res = {}
    for mode in ("f32", "bf16x3"):
        pol.gemm = mode
        for fuse in (False, True):
            pol.fuse_fwd_bwd = fuse
            pol.minibatch_grad(x, action, old_logp, adv, target, var, 0.2)
            torch.cuda.synchronize()
            res[(mode, fuse)] = (pol.G.clone(), float(pol.loss_value(n)), untile(pol.dz["dz1"], n, 256))
    assert torch.equal(res[("bf16x3", False)][0], res[("bf16x3", True)][0])       # one launch == two launches, bit for bit
    g32, l32, z32 = res[("f32", True)]
    g3, l3, z3 = res[("bf16x3", True)]
    assert abs(l3 - l32) <= 2e-6 * max(1.0, abs(l32))
    torch.testing.assert_close(g3, g32, rtol=2e-4, atol=2e-7)
    torch.testing.assert_close(z3, z32, rtol=2e-4, atol=1e-9)
    agent = _bare_agent(ref, var)
    loss = agent.minibatch_loss(x, action, old_logp, target.unsqueeze(-1), adv.unsqueeze(-1))
    assert abs(l3 - float(loss)) <= 2e-5 * max(1.0, abs(float(loss)))
    pol.adam_step()
    torch.cuda.synchronize()
    assert int(pol.tile_wait_error.item()) == 0
    for buf, src, dst in ((pol.PB, pol._src_fb, pol._dst_fb), (pol.PTB, pol._src_tb, pol._dst_tb)):
        for term, plane in enumerate(split_bf16x3(pol.P[src])):
            assert torch.equal(buf[dst + 512 * term], plane)


@pytest.mark.parametrize("handoff", ["sc1", "xcd"])
def test_fused_launch_handoff_modes_bit_equal_two_launches(handoff, monkeypatch):
    """Both hand-off forms of mlp_forward_backward (sc1 write-through stores + L1-bypassing loads, no
    placement assumption; plain accesses on one XCD) leave bit for bit what mlp_forward +
    mlp_backward_dx leave, repeated over 6 launches on changing weights (a consumer that read a stale
    line of the previous launch's activations would differ)."""
    monkeypatch.setenv("FLY_FWD_BWD_HANDOFF", handoff)
    n = 40960
    net, ref, pol, batch = _setup(n, 11)
    assert pol.handoff == handoff and pol.fuse_fwd_bwd
    x, action, old_logp, adv, target, var = batch
    for it in range(6):
        pol.fuse_fwd_bwd = True
        pol.minibatch_grad(x, action, old_logp, adv, target, var, 0.2)
        fused = {k: v.clone() for k, v in list(pol.dz.items()) + list(pol.saves.items())}
        g_fused = pol.G.clone()
        assert pol.check_fused_launch() == 0
        pol.fuse_fwd_bwd = False
        for v in pol.dz.values():
            v.fill_(float("nan"))
        pol.minibatch_grad(x, action, old_logp, adv, target, var, 0.2)
        for k, v in list(pol.dz.items()) + list(pol.saves.items()):
            assert torch.equal(v, fused[k]), (it, k)
        assert torch.equal(pol.G, g_fused), it
        pol.adam_step()                                   # the next launch runs on different weights and activations
        x = x + 0.01 * torch.randn_like(x)


def test_failed_fused_launch_is_refused_on_the_device_and_redone():
    """Force the failure the fused launch can have (consumer range shifted by one workgroup: in "xcd"
    mode every tile's backward lands on another XCD than its forward -> err 2, no dZ written):
      * the optimizer kernels refuse the step on the device: parameters, moments and the step counter do
        not move, for that minibatch and for every later one while err is set;
      * PPO.update notices the shortfall with its ONE host sync, switches to two launches, redoes exactly
        the refused minibatches, and ends bit-identical to an undisturbed update."""
    import contextlib
    import ctypes as C
    import io
    import os
    from fly_bproject_amd import _lib
    from fly_bproject_amd.ppo import PPO
    from tests.hip_helpers import make_args
    lib = _lib.load()
    lib.flyhip_debug_set_fwd_bwd_consumer_shift.argtypes = [C.c_int]
    lib.flyhip_debug_set_fwd_bwd_consumer_shift.restype = None
    os.environ["FLY_FWD_BWD_HANDOFF"] = "xcd"
    try:
        res = {}
        for tag in ("clean", "broken"):
            torch.manual_seed(0)
            out = io.StringIO()
            with contextlib.redirect_stdout(out):
                agent = PPO(make_args(4096))
                pol = agent.policy
                pol.gemm = "f32"                      # the flag hand-off inside mlp_forward_backward is the fp32 / three-launch path's
                assert pol.fuse_fwd_bwd and pol.handoff == "xcd"
                for _ in range(agent.rollout_size - 1):
                    agent.run()
                if tag == "broken":
                    # device-level check first: one poisoned minibatch + Adam leaves everything untouched
                    lib.flyhip_debug_set_fwd_bwd_consumer_shift(1)
                    before = (pol.P.clone(), pol.exp_avg.clone(), pol.exp_avg_sq.clone(), int(pol.step.item()))
                    rows = agent.mini_chunk_size * 4096
                    pol.minibatch_grad(agent.all_obs[:agent.mini_chunk_size].view(rows, 73),
                                       agent.all_acts[:agent.mini_chunk_size].view(rows, 18),
                                       agent.all_log_prob[:agent.mini_chunk_size].view(rows),
                                       agent.all_advantage[:agent.mini_chunk_size].view(rows),
                                       agent._target[:agent.mini_chunk_size].view(rows), agent._action_var, 0.2, fuse_norm=True)
                    pol.adam_step(norm_ready=True)
                    torch.cuda.synchronize()
                    assert int(pol.tile_wait_error.item()) == 2 and float(pol.G[76]) == 1.0
                    assert torch.equal(pol.P, before[0]) and torch.equal(pol.exp_avg, before[1])
                    assert torch.equal(pol.exp_avg_sq, before[2]) and int(pol.step.item()) == before[3]
                    pol.steps_issued -= 1                     # that call was this test's, not the update's
                    pol.tile_wait_error.zero_()
                    # now the real thing: the update's first fused launch fails, the device refuses all 75 steps
                agent.run()                                    # last env step of the rollout + update
                lib.flyhip_debug_set_fwd_bwd_consumer_shift(0)
            torch.cuda.synchronize()
            assert agent.optim_step == 75 and int(pol.step.item()) == 75
            if tag == "broken":
                assert "redoing them with two launches" in out.getvalue()
                assert not pol.fuse_fwd_bwd and pol.fused_launch_failures == 1
            else:
                assert pol.fuse_fwd_bwd
            res[tag] = (pol.P.clone(), pol.exp_avg.clone(), pol.exp_avg_sq.clone(), agent.all_advantage.clone())
            agent.exit()
        for a, b in zip(res["clean"], res["broken"]):
            assert torch.equal(a, b)
    finally:
        lib.flyhip_debug_set_fwd_bwd_consumer_shift(0)
        os.environ.pop("FLY_FWD_BWD_HANDOFF", None)


@pytest.mark.parametrize("mode", ["self_norm", "norm_ready", "two_launch"])
def test_adam_refuses_a_gradient_its_producer_marked_invalid(mode):
    """`grad_invalid` of mlp_adam_step (the err word of the peer-to-peer exchange: ANY of its workgroups sets it): nonzero ->
    parameters, moments, fragment copies and the step counter stay exactly as they were; zero -> the step is the usual one."""
    net, ref, pol, (x, action, old_logp, adv, target, var) = _setup(2048, 23)
    word = torch.zeros(1, dtype=torch.int32, device=x.device)

    def step():
        pol.minibatch_grad(x, action, old_logp, adv, target, var, 0.2, fuse_norm=mode == "norm_ready")
        pol.adam_step(norm_ready=mode == "norm_ready", self_norm=mode == "self_norm", grad_invalid=word)
    step()
    torch.cuda.synchronize()
    assert int(pol.step) == 1
    keep = [t.clone() for t in (pol.P, pol.PF, pol.PT, pol.PB, pol.exp_avg, pol.exp_avg_sq)]
    word.fill_(1)
    if mode != "norm_ready":            # (norm_ready: the gradient's own reduction advances the counter, as on a single rank)
        step()
        torch.cuda.synchronize()
        assert int(pol.step) == 1
    else:
        pol.adam_step(norm_ready=True, grad_invalid=word)
        torch.cuda.synchronize()
    for a, b in zip(keep, (pol.P, pol.PF, pol.PT, pol.PB, pol.exp_avg, pol.exp_avg_sq)):
        assert torch.equal(a, b)
    word.zero_()
    step()
    torch.cuda.synchronize()
    assert not torch.equal(keep[0], pol.P)
