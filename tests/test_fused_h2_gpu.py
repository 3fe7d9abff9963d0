"""GPU (-m gpu): `mlp_fused_grad_h2` -- the fused optimizer-step gradient (ppo.py:184-197) in the fp16x2 arithmetic (two fp16 terms
per operand, three products per k block, per-class power-of-two scales: csrc/mlp_fused_h2.inc) -- against fp64, against the bf16x3
kernel it stands beside, for determinism, and for what happens when a value does not fit fp16: the step is refused on the device and
redone on bf16x3, bit for bit an undisturbed bf16x3 step.  The reference-golden (g7) and autograd checks of this kernel are the
`[f16x2]` parametrisations of tests/test_mlp_train_gpu.py."""
import contextlib
import io

import pytest
import torch

from tests.test_fused_step_gpu import WIDTH, _chain, _poison
from tests.test_mlp_train_gpu import DEV, _setup

pytestmark = pytest.mark.gpu


def _fp64_chain(ref, x, action, old_logp, adv, target, var, clip=0.2):
    """The minibatch's forward, loss gradient and dX chain (ppo.py:184-197) in float64 torch: dict of the chain tensors."""
    sd = {k: v.double() for k, v in ref.state_dict().items()}
    elu = torch.nn.functional.elu
    x = x.double()
    n = x.shape[0]
    z1 = x @ sd["shared_net.0.weight"].T + sd["shared_net.0.bias"]
    h1 = elu(z1)
    h2 = elu(h1 @ sd["shared_net.2.weight"].T + sd["shared_net.2.bias"])
    a1 = elu(h2 @ sd["to_mean.0.weight"].T + sd["to_mean.0.bias"])
    c1 = elu(h2 @ sd["to_value.0.weight"].T + sd["to_value.0.bias"])
    h3 = torch.cat([a1, c1], 1)
    mu = elu(a1 @ sd["to_mean.2.weight"].T + sd["to_mean.2.bias"])
    v = c1 @ sd["to_value.2.weight"].T + sd["to_value.2.bias"]
    out = torch.zeros(n, 32, dtype=torch.float64, device=x.device)
    out[:, :18], out[:, 18:19] = mu, v
    leaves = [t.detach().requires_grad_(True) for t in (mu, v)]
    var64 = var.double()
    L = var64.sqrt()
    M = (((action.double() - leaves[0]) / L) ** 2).sum(-1)
    logp = -0.5 * (18 * 1.8378770664093453 + M) - L.log().sum()
    ratio = (logp - old_logp.double()).exp().unsqueeze(-1)
    A = adv.double().unsqueeze(-1)
    loss = (-torch.min(ratio * A, ratio.clamp(1 - clip, 1 + clip) * A)
            + torch.nn.functional.smooth_l1_loss(leaves[1], target.double().unsqueeze(-1))).mean()
    loss.backward()
    dmu, dv = leaves[0].grad, leaves[1].grad
    dz4 = torch.zeros_like(out)
    dz4[:, :18] = dmu * torch.where(mu > 0, torch.ones_like(mu), mu + 1)           # ELU' through the output
    dz4[:, 18:19] = dv
    W4 = torch.zeros(32, 128, dtype=torch.float64, device=x.device)
    W4[:18, :64], W4[18:19, 64:] = sd["to_mean.2.weight"], sd["to_value.2.weight"]
    W3 = torch.cat([sd["to_mean.0.weight"], sd["to_value.0.weight"]], 0)
    dz3 = (dz4 @ W4) * torch.where(h3 > 0, torch.ones_like(h3), h3 + 1)
    dz2 = (dz3 @ W3) * torch.where(h2 > 0, torch.ones_like(h2), h2 + 1)
    dz1 = (dz2 @ sd["shared_net.2.weight"]) * torch.where(h1 > 0, torch.ones_like(h1), h1 + 1)
    return {"out": out, "h1": h1, "h2": h2, "h3": h3, "dz4": dz4, "dz3": dz3, "dz2": dz2, "dz1": dz1}


def _errs(got, want):
    """Per tensor: max |got - want| relative to the tensor's largest |want| (the scale every test of this suite uses)."""
    return {k: float((got[k].double() - want[k]).abs().max()) / (float(want[k].abs().max()) + 1e-300) for k in WIDTH}


@pytest.mark.parametrize("n", [33, 4099, 40960])
def test_h2_chain_against_fp64_no_worse_than_the_fp32_mfma_chain(n):
    """Every GEMM class of the step -- forward-like (x, h1 .. h3 against weights) and gradient-like (dz4 .. dz1, |dz| ~ 1e-6 with
    decades of spread, against weights) -- held to float64: the fp16x2 chain's error stays inside the suite's bar (2e-5 of the tensor's
    largest value) and inside 2x the fp32-MFMA chain's on the same data (the bar bf16x3 was admitted under; it is usually smaller)."""
    from fly_bproject_amd.policy import untile
    net, ref, pol, batch = _setup(n, 13, gemm="f16x2")
    x, action, old_logp, adv, target, var = batch
    want = _fp64_chain(ref, *batch)
    errs = {}
    pol.minibatch_grad(*batch, 0.2, dump=True)
    torch.cuda.synchronize()
    assert int(pol.h2_overflow) == 0
    errs["f16x2"] = _errs(_chain(pol, n), want)
    pol.step_gemm = "bf16x3"
    pol.minibatch_grad(*batch, 0.2, dump=True)
    torch.cuda.synchronize()
    errs["bf16x3"] = _errs(_chain(pol, n), want)
    pol.gemm = "f32"                                            # the three-launch fp32-MFMA kernels leave the same tensors
    pol.minibatch_grad(*batch, 0.2)
    torch.cuda.synchronize()
    got32 = {k: untile((pol.saves if k in pol.saves else pol.dz)[k], n, w) for k, w in WIDTH.items()}
    errs["f32"] = _errs(got32, want)
    for k in WIDTH:
        assert errs["f16x2"][k] <= 2e-5, (k, errs)
        assert errs["f16x2"][k] <= 2.0 * errs["f32"][k] + 2e-7, (k, errs["f16x2"][k], errs["f32"][k], errs["bf16x3"][k])


@pytest.mark.parametrize("n", [4099, 40960])
def test_h2_gradient_against_fp64_and_bf16x3(n):
    """dW = dZ^T A and db = colsum(dZ) of the fp16x2 launch against float64 on the chain values it dumped (<= 2e-5 of each block's
    largest entry: the bar of the separate dW kernels), and the whole gradient against the bf16x3 fused step's."""
    net, ref, pol, batch = _setup(n, 31, gemm="f16x2")
    x = batch[0]
    pol.minibatch_grad(*batch, 0.2, dump=True)
    torch.cuda.synchronize()
    c, G = _chain(pol, n), pol.G.clone()
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
    assert torch.all(G[:256 * 80].view(256, 80)[:, 73:] == 0)          # padding columns of W1 see x == 0; element 76 (the mark) is 0
    pol.step_gemm = "bf16x3"
    pol.minibatch_grad(*batch, 0.2)
    torch.cuda.synchronize()
    m = pol.grad_mask > 0
    assert float((G[m] - pol.G[m]).abs().max()) <= 2e-5 * float(pol.G[m].abs().max())


def test_h2_is_deterministic_and_grid_independent():
    """Under the SAME scale table two launches leave the same gradient and chain bit for bit, and 7 workgroups instead of one per CU
    (accumulators live through ~183 tiles) leave the same chain and a gradient equal up to summation order."""
    import ctypes as C
    from fly_bproject_amd import _lib
    n = 40960 + 19
    net, ref, pol, batch = _setup(n, 5, gemm="f16x2")
    pol.h2_freeze = True
    pol.minibatch_grad(*batch, 0.2, dump=True)
    torch.cuda.synchronize()
    g1, c1, sc = pol.G.clone(), _chain(pol, n), pol.h2_scales[:32].clone()
    _poison(pol)
    pol.minibatch_grad(*batch, 0.2, dump=True)
    torch.cuda.synchronize()
    assert torch.equal(pol.G, g1) and torch.equal(pol.h2_scales[:32], sc)
    c2 = _chain(pol, n)
    for k in WIDTH:
        assert torch.equal(c2[k], c1[k]), k
    lib = _lib.load()
    lib.flyhip_debug_set_fused_grid.argtypes = [C.c_int]
    lib.flyhip_debug_set_fused_grid.restype = None
    lib.flyhip_debug_set_fused_grid(7)
    try:
        _poison(pol)
        pol.minibatch_grad(*batch, 0.2, dump=True)
        torch.cuda.synchronize()
    finally:
        lib.flyhip_debug_set_fused_grid(0)
    c7 = _chain(pol, n)
    for k in WIDTH:
        assert torch.equal(c7[k], c1[k]), k
    m = pol.grad_mask > 0
    assert float((pol.G[m] - g1[m]).abs().max()) <= 2e-5 * float(g1[m].abs().max())


def test_h2_scales_follow_the_data_and_the_planes_follow_the_weights():
    """After a launch every class's scale puts THAT launch's largest |value| into the class's window ([2^7, 2^8) for x / h1 .. h3,
    [2^2, 2^3) for dz4 .. dz1: csrc/mlp_fused_h2.inc, h2_target_exp); the
    weight planes mlp_adam_step maintains equal a fresh split of the weights under the scales it published, whose headroom over the
    current max |w| is 8x .. 64x (16x .. 32x when the scale was set; max |w| counted as 2^-4 at least)."""
    import math
    from fly_bproject_amd.policy import H2_INV, H2_W0, OFF_W1, OFF_W2, OFF_W3, OFF_W4, PACKED, split_f16x2
    n = 4099
    net, ref, pol, batch = _setup(n, 7, gemm="f16x2")
    x, action, old_logp, adv, target, var = batch
    for it in range(3):
        pol.minibatch_grad(x, action, old_logp, adv * (10.0 if it == 1 else 1.0), target, var, 0.2, fuse_norm=True, dump=(it == 2))
        before = pol.h2_scales.clone()
        torch.cuda.synchronize()
        # the table now holds the scales for the NEXT launch, made from the maxima (of |value * old scale|) this launch saw
        pol.adam_step(norm_ready=True)
    assert int(pol.h2_overflow) == 0 and int(pol.step) == 3
    sc = pol.h2_scales.cpu()
    c = _chain(pol, n)
    true_max = [float(x.abs().max()), *[float(c[k].abs().max()) for k in ("h1", "h2", "h3", "dz4", "dz3", "dz2", "dz1")]]
    for i, m in enumerate(true_max):
        scaled, e = m * float(sc[i]), (7 if i < 4 else 2)
        assert 2.0 ** e <= scaled * (1 + 1e-6) and scaled < 2.0 ** (e + 1) * (1 + 1e-6), (i, m, float(sc[i]), scaled)
        assert float(sc[i]) * float(sc[H2_INV + i]) == 1.0 and math.log2(float(sc[i])).is_integer()
    bounds = (OFF_W1, OFF_W2, OFF_W3, OFF_W4, PACKED)
    layer_scale = torch.ones(PACKED, device=DEV)
    for l in range(4):
        sel = pol._src_fb[(pol._src_fb >= bounds[l]) & (pol._src_fb < bounds[l + 1])]
        k = float(sc[H2_W0 + l])
        head = 65504.0 / (float(pol.P[sel].abs().max()) * k)
        assert 8.0 <= head <= 64.0, (l, head)
        layer_scale[bounds[l]:bounds[l + 1]] = k
    for buf, src, dst in ((pol.PH, pol._src_fb, pol._dst_fb), (pol.PTH, pol._src_tb, pol._dst_tb)):
        dh = (dst // 1536) * 1024 + dst % 1536
        for term, plane in enumerate(split_f16x2(pol.P[src], layer_scale[src])):
            assert torch.equal(buf[dh + 512 * term], plane), term
    del before


def test_h2_overflow_is_refused_on_the_device_and_the_redo_equals_a_bf16x3_step():
    """Scales far too large for the data (h1 scaled past 65504): the launch marks its gradient invalid, sets the sticky word and does
    not advance the step; mlp_adam_step refuses -- weights, moments, planes unchanged -- and goes on refusing later launches although
    THEIR values fit; after the host clears the word, the redo on the bf16x3 kernel leaves bit for bit what a bf16x3 step leaves from
    the same state."""
    from fly_bproject_amd.policy import ERR_SLOT, H2_INV
    n = 4099
    net, ref, pol, batch = _setup(n, 17, gemm="f16x2")
    netb, refb, polb, batchb = _setup(n, 17, gemm="bf16x3")
    for p_ in (pol, polb):                                        # one ordinary step on both (fp16x2 here, bf16x3 there) ...
        p_.minibatch_grad(*batch, 0.2, fuse_norm=True)
        p_.adam_step(norm_ready=True)
    polb.P.copy_(pol.P); polb.exp_avg.copy_(pol.exp_avg); polb.exp_avg_sq.copy_(pol.exp_avg_sq)   # ... then the same state
    polb.refresh()
    state = (pol.P.clone(), pol.exp_avg.clone(), pol.exp_avg_sq.clone(), pol.PH.clone(), pol.PB.clone())
    good = pol.h2_scales.clone()
    pol.h2_scales[1] = 2.0 ** 40
    pol.h2_scales[H2_INV + 1] = 2.0 ** -40
    pol.minibatch_grad(*batch, 0.2, fuse_norm=True)
    pol.adam_step(norm_ready=True)
    torch.cuda.synchronize()
    assert int(pol.h2_overflow) == 1 and float(pol.G[ERR_SLOT]) == 1.0 and int(pol.step) == 1
    pol.h2_scales.copy_(good)                                     # a launch whose values fit, while the word is still set: refused too
    pol.minibatch_grad(*batch, 0.2, fuse_norm=True)
    pol.adam_step(norm_ready=True)
    torch.cuda.synchronize()
    assert float(pol.G[ERR_SLOT]) == 1.0 and int(pol.step) == 1
    for a, b in zip(state, (pol.P, pol.exp_avg, pol.exp_avg_sq, pol.PH, pol.PB)):
        assert torch.equal(a, b)
    pol.steps_issued -= 2
    pol.h2_overflow.zero_()                                       # what PPO._update_hip does: clear, redo on bf16x3
    pol.h2_suspended = True
    pol.minibatch_grad(*batch, 0.2, fuse_norm=True)
    pol.adam_step(norm_ready=True)
    pol.h2_suspended = False
    polb.minibatch_grad(*batch, 0.2, fuse_norm=True)
    polb.adam_step(norm_ready=True)
    torch.cuda.synchronize()
    assert int(pol.step) == 2 == int(polb.step)
    assert torch.equal(pol.P, polb.P) and torch.equal(pol.exp_avg, polb.exp_avg) and torch.equal(pol.exp_avg_sq, polb.exp_avg_sq)
    assert torch.equal(pol.PB, polb.PB)


def test_h2_update_with_an_overflow_in_it_ends_where_the_mixed_update_ends(monkeypatch):
    """A whole PPO.update (75 steps, 4096 envs) in which the scales are knocked out of range before step 40: steps 40 .. 74 are refused
    on the device, PPO._update_hip finds the counter 35 short, redoes them on bf16x3 -- and ends bit for bit where an update ends that
    runs steps 0 .. 39 in fp16x2 and 40 .. 74 in bf16x3 by construction."""
    from fly_bproject_amd.policy import H2_INV, PackedPolicy
    from fly_bproject_amd.ppo import PPO
    from tests.hip_helpers import make_args
    outs = {}
    for mode in ("overflow", "mixed"):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            agent = PPO(make_args(4096))
            agent.policy.gemm = "f16x2"
            calls = {"n": 0}
            orig = PackedPolicy.minibatch_grad

            def patched(self, *a, _orig=orig, _calls=calls, _mode=mode, **kw):
                if kw.get("fuse_norm") and not self.h2_suspended:            # (calibration launches pass fuse_norm=False)
                    if _calls["n"] == 40:
                        if _mode == "overflow":
                            self.h2_scales[2] = 2.0 ** 40
                            self.h2_scales[H2_INV + 2] = 2.0 ** -40
                        else:
                            self.h2_suspended = "by construction"            # truthy: bf16x3 from here on, never cleared by the redo path
                    _calls["n"] += 1
                return _orig(self, *a, **kw)

            monkeypatch.setattr(PackedPolicy, "minibatch_grad", patched)
            for _ in range(agent.rollout_size):
                agent.run()
            torch.cuda.synchronize()
            monkeypatch.setattr(PackedPolicy, "minibatch_grad", orig)
        assert agent.optim_step == 75 and int(agent.policy.step) == 75
        assert agent.policy.h2_overflows == (1 if mode == "overflow" else 0)
        outs[mode] = (agent.policy.P.clone(), agent.policy.exp_avg.clone(), agent.policy.exp_avg_sq.clone())
        agent.exit()
    for a, b in zip(outs["overflow"], outs["mixed"]):
        assert torch.equal(a, b)
