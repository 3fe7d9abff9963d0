"""GPU: the DQN update's fp16x2 arithmetic (csrc/dqn_fused_h2.inc; reference UselessFiles/dqn.py:64-85) beyond the parity cases of
tests/test_dqn.py, which run every fused-update test in both arithmetics: its error against float64 beside bf16x3's, the scale
windows and the weight planes `dqn_fused_update_h2` leaves, and the refusal path (an update whose values do not fit fp16 under the
lagged scales is formed again by the bf16x3 launches: bit for bit what an undisturbed bf16x3 update leaves)."""
import math

import numpy as np
import pytest
import torch

from tests.test_dqn import _bare_dqn

pytestmark = pytest.mark.gpu


def _batch(parts, n, seed):
    g = torch.Generator(device="cuda:0"); g.manual_seed(seed)
    r = lambda *s: torch.randn(*s, device="cuda:0", generator=g)   # noqa: E731
    u = lambda *s: torch.rand(*s, device="cuda:0", generator=g)   # noqa: E731
    return [(r(n, 73), u(n) * 2 - 1, r(n) * 2, r(n, 73), (u(n) > 0.1).float()) for _ in range(parts)]


def _perturb_target(d):
    with torch.no_grad():
        for p_ in d.q_target.parameters():
            p_.add_(0.05 * torch.randn_like(p_))
    d.packed.refresh()


def _unambiguous(d, chunks, margin=1e-5):
    """`chunks` with every row whose float64 pre-activations come within `margin` of zero (any of the 512 hidden units of the online
    network) replaced by a copy of a row that does not.  LeakyReLU' jumps from 0.01 to 1 at zero: a unit at 1e-8 takes either side
    in ANY arithmetic (each is right to its own rounding), and one such unit moves a row's whole contribution -- 1 / sqrt(B) of the
    gradient's scale, 2.6e-3 at 32768 rows (tools/dqn_h2_debug.py found exactly one: hidden column tile 6, gone when H1's scale moved).
    About 0.7 % of the rows at this margin.  The target network's pass has no such jump (its output is used, not its derivative)."""
    w1, b1, w2, b2 = (t.detach().double() for t in (d.q.net[0].weight, d.q.net[0].bias, d.q.net[2].weight, d.q.net[2].bias))
    out, replaced = [], 0
    for obs, act, rew, nxt, done in chunks:
        z1 = obs.double() @ w1.T + b1
        z2 = torch.nn.functional.leaky_relu(z1) @ w2.T + b2
        bad = torch.minimum(z1.abs().min(1)[0], z2.abs().min(1)[0]) < margin
        row = obs[torch.nonzero(~bad)[0, 0]].clone()
        obs = obs.clone()
        obs[bad] = row
        replaced += int(bad.sum())
        out.append((obs, act, rew, nxt, done))
    return out, replaced


def _grad64(d, chunks):
    """The update's gradient by torch autograd in float64 on the same weights and rows."""
    import copy
    obs, act, rew, nxt, done = (torch.cat([c[i] for c in chunks]).double() for i in range(5))
    q, qt = copy.deepcopy(d.q).double(), copy.deepcopy(d.q_target).double()
    B = obs.shape[0]
    idx = torch.round(0.5 * (act.float() + 1) * 17).long()
    q_val = q(obs)[torch.arange(B), idx]
    with torch.no_grad():
        target = rew + 0.99 * qt(nxt).max(1)[0] * done
    loss = torch.nn.functional.smooth_l1_loss(q_val, target)
    return torch.autograd.grad(loss, list(q.parameters())), float(loss)


def _views(G):
    return [G[:256 * 80].view(256, 80)[:, :73], G[20480:20736], G[20736:86272].view(256, 256), G[86272:86528],
            G[86528:94720].view(32, 256)[:18], G[94720:94738]]


def _state(d):
    pk = d.packed
    return [t.clone() for t in (pk.P, pk.P_tgt, pk.exp_avg, pk.exp_avg_sq, pk.step)]


def _restore(d, st):
    pk = d.packed
    for dst, src in zip((pk.P, pk.P_tgt, pk.exp_avg, pk.exp_avg_sq, pk.step), st):
        dst.copy_(src)
    pk.refresh()


def test_h2_gradient_error_against_float64_beside_bf16x3():
    """4 x 8192 rows, none with a hidden unit within 1e-5 of LeakyReLU's kink (_unambiguous).  Per parameter tensor:
    max |g - g64| / max |g64| of the fp16x2 update <= 2e-5 and <= twice the fp32-MFMA per-step path's (the reference's numerics) + 2e-6;
    bf16x3 printed beside.  The loss within 2e-6."""
    torch.manual_seed(2)
    raw = _batch(4, 8192, 21)
    errs = {}
    for gemm, fused in (("f16x2", True), ("bf16x3", True), ("f32", False)):
        torch.manual_seed(2)
        d = _bare_dqn(rows=8192, fused=fused, gemm=gemm if fused else "f16x2")
        _perturb_target(d)
        chunks, replaced = _unambiguous(d, raw)             # (the same weights in all three: the same rows)
        assert 0 < replaced < 0.03 * 4 * 8192
        want, loss64 = _grad64(d, chunks)
        loss = d.update(chunks)
        torch.cuda.synchronize()
        assert abs(float(loss) - loss64) <= 2e-6 * abs(loss64) + 1e-9, gemm
        errs[gemm] = [float((g.double() - w).abs().max() / w.abs().max()) for g, w in zip(_views(d.packed.G), want)]
        assert d.h2_overflows == 0
    print("max |g - g64| / max |g64| per tensor:", {k: ["%.2e" % e for e in v] for k, v in errs.items()})
    for e_h2, e_f32 in zip(errs["f16x2"], errs["f32"]):
        assert e_h2 <= 2e-5 and e_h2 <= 2 * e_f32 + 2e-6, errs


def test_h2_scales_land_in_their_windows_and_planes_hold_the_weights():
    """After an update: the maxima of |scaled value| the launch recorded lie in their class's window once the scales have seen the same
    rows (activations [2^7, 2^8), gradients [2^2, 2^3), weights [2^11, 2^12)); the two-term planes reproduce every weight to 2^-21
    relative to its layer's maximum, online and target, forward and transposed."""
    from fly_bproject_amd import dqn as D
    torch.manual_seed(4)
    d = _bare_dqn(rows=4096, fused=True)
    _perturb_target(d)
    chunks = _batch(3, 4096, 5)
    st = _state(d)
    d.update(chunks)                # calibrates (two passes) + the update
    _restore(d, st)
    d.update(chunks)                # the same rows under the scales the first update left
    torch.cuda.synchronize()
    sc = d.packed.h2_scales.cpu().numpy()
    assert d.h2_overflows == 0 and int(d.packed.h2_overflow) == 0
    for c in (0, 1, 2):
        assert 2.0 ** 7 <= sc[32 + c] < 2.0 ** 8, (c, sc[32 + c])
    for c in (5, 6):
        assert 2.0 ** 2 <= sc[32 + c] < 2.0 ** 3, (c, sc[32 + c])
    np.testing.assert_array_equal(sc[:16] * sc[16:32], np.ones(16, np.float32))
    assert all(np.log2(s) == np.round(np.log2(s)) for s in sc[:14] if s > 0)          # powers of two
    # weights: s_l max|w_l| in [2^11, 2^12) (the planes were made from the weights in front of the LAST update: restore them)
    _restore(d, st)
    d.h2_freeze = True
    d.update(chunks)
    _restore(d, st)
    torch.cuda.synchronize()
    pk = d.packed
    layers = ((D.OFF_W1, D.OFF_B1), (D.OFF_W2, D.OFF_B2), (D.OFF_W3, D.OFF_B3))
    sc = pk.h2_scales.cpu().numpy()
    for which, (master, planes_f) in enumerate(((pk.P, pk.QH), (pk.P_tgt, pk.QH_tgt))):
        for l, (a, b) in enumerate(layers):
            s = float(sc[8 + 3 * which + l])
            m = float(master[a:b].abs().max())
            # max(max |w|, 2^-4) -> [2^11, 2^12): nn.Linear(256, .)'s init is U(-1/16, 1/16), its maximum sits just under the floor
            assert s == 2.0 ** (11 - math.floor(math.log2(max(m, 0.0625)))) and s * m < 2.0 ** 12
            assert float(sc[32 + 8 + 3 * which + l]) == m
            idx = pk.idx_fb[a:b].long()
            hf = (idx // 1536) * 1024 + idx % 1536
            got = (planes_f[hf].view(torch.float16).double() + planes_f[hf + 512].view(torch.float16).double()) / s
            assert float((got - master[a:b].double()).abs().max()) <= 2.0 ** -21 * m, (which, l)
    for l, (a, b) in enumerate(layers[1:], start=1):
        s = float(sc[8 + l])
        idx = pk.idx_tb[a:b].long()
        ht = (idx // 1536) * 1024 + idx % 1536
        got = (pk.QTH[ht].view(torch.float16).double() + pk.QTH[ht + 512].view(torch.float16).double()) / s
        assert float((got - pk.P[a:b].double()).abs().max()) <= 2.0 ** -21 * float(pk.P[a:b].abs().max()), l


@pytest.mark.parametrize("cls", [0, 2, 6])
def test_h2_overflow_is_refused_and_the_update_redone_in_bf16x3(cls):
    """A lagged scale 2^14 too large (class X, H2 or dZ1): the launch's maximum does not fit fp16, the device word is set, `DQN.update`
    forms the gradient again with the bf16x3 launches -- the packed gradient, the loss and the networks afterwards are bit for bit
    what a bf16x3 update leaves -- and the NEXT update calibrates again and runs in fp16x2."""
    torch.manual_seed(6)
    chunks = _batch(2, 4096, 9)
    ref = _bare_dqn(rows=4096, fused=True, gemm="bf16x3")
    _perturb_target(ref)
    st = _state(ref)
    l_ref = float(ref.update(chunks)); torch.cuda.synchronize()
    g_ref, p_ref, pt_ref = ref.packed.G.clone(), ref.packed.P.clone(), ref.packed.P_tgt.clone()

    d = _bare_dqn(rows=4096, fused=True, gemm="f16x2")
    _restore(d, st)
    d.update(chunks); torch.cuda.synchronize()               # calibrated
    assert d.h2_calibrated and d.h2_overflows == 0
    _restore(d, st)
    with torch.no_grad():
        d.packed.h2_scales[cls] *= 2.0 ** 14
        d.packed.h2_scales[16 + cls] /= 2.0 ** 14
    l = float(d.update(chunks)); torch.cuda.synchronize()
    assert d.h2_overflows == 1 and not d.h2_calibrated and int(d.packed.h2_overflow) == 0
    assert l == l_ref and torch.equal(d.packed.G, g_ref) and torch.equal(d.packed.P, p_ref) and torch.equal(d.packed.P_tgt, pt_ref)
    # the next update calibrates again and is an fp16x2 update
    g_b3_next = None
    l2 = float(d.update(chunks)); torch.cuda.synchronize()
    assert d.h2_overflows == 1 and d.h2_calibrated
    ref.update(chunks); torch.cuda.synchronize()
    g_b3_next = ref.packed.G
    m = d.packed.grad_mask > 0
    assert float((d.packed.G - g_b3_next)[m].abs().max()) <= 2e-5 * float(g_b3_next[m].abs().max()) and np.isfinite(l2)
    assert not torch.equal(d.packed.G, g_b3_next)           # (it really ran the other arithmetic)


def test_h2_update_is_independent_of_the_grid():
    """One workgroup per CU or a quarter of them: the partial slabs are summed in a fixed order per grid, so two grids differ only in
    summation order -- to 2e-6 of the gradient's scale -- and each is deterministic run to run.  (Smaller grid: fewer rows than CUs x 32.)"""
    torch.manual_seed(8)
    d = _bare_dqn(rows=2048, fused=True)
    _perturb_target(d)
    st = _state(d)
    big = _batch(8, 2048, 3)                 # 512 tiles on 256 workgroups
    d.update(big); _restore(d, st)
    d.h2_freeze = True
    d.update(big); torch.cuda.synchronize(); g_a = d.packed.G.clone(); _restore(d, st)
    d.update(big); torch.cuda.synchronize(); g_b = d.packed.G.clone(); _restore(d, st)
    assert torch.equal(g_a, g_b)
    # the same rows as 64-tile updates (grid = 64 workgroups), averaged
    acc = torch.zeros_like(g_a)
    for i in range(8):
        d.update(big[i:i + 1]); torch.cuda.synchronize(); acc += d.packed.G / 8; _restore(d, st)
    m = d.packed.grad_mask > 0
    assert float((acc - g_a)[m].abs().max()) <= 2e-5 * float(g_a[m].abs().max())
    assert d.h2_overflows == 0


def test_h2_overflow_in_the_training_loop_is_refused_on_the_device_and_settled_later():
    """`DQN.run()`'s own updates do not synchronise with the host (a blocking read per env step drains the launch queue): the optimizer
    launch takes the overflow word as `grad_invalid`, the host looks at a copy of it when a later update begins.  A lagged scale pushed
    2^14 too high between two env steps: the updates from there on are refused ON THE DEVICE (nothing moves, the step counter stays),
    the host finds out within the next steps, clears the word, forms as many updates in bf16x3 and goes on in fp16x2 -- at the end every
    env step has had its update and the networks are finite."""
    import contextlib
    import io
    from fly_bproject_amd.dqn import DQN
    from tests.hip_helpers import make_args
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        agent = DQN(make_args(256, dqn_mini_batch_size=8, replay_steps=16))
        for _ in range(14):
            agent.run()
    torch.cuda.synchronize()
    assert agent.update_gemm == "f16x2" and agent.h2_calibrated and agent.h2_overflows == 0
    issued0 = agent._updates_issued
    assert issued0 == 14 - 8 and int(agent.packed.step) == issued0
    p_before = agent.packed.P.clone()
    with torch.no_grad():
        agent.packed.h2_scales[0] *= 2.0 ** 14
        agent.packed.h2_scales[16] /= 2.0 ** 14
    with contextlib.redirect_stdout(io.StringIO()):
        agent.run()                                      # this update overflows: refused on the device
    torch.cuda.synchronize()
    assert int(agent.packed.h2_overflow) == 1 and int(agent.packed.step) == issued0 and torch.equal(agent.packed.P, p_before)
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(4):
            agent.run()                                  # the host notices, redoes the refused updates in bf16x3, calibrates, goes on
        agent._h2_poll(block=True)                       # (at the latest here)
        agent.run()                                      # and an fp16x2 update again: calibrates first
        agent._h2_poll(block=True)
    torch.cuda.synchronize()
    assert agent.h2_overflows >= 1 and int(agent.packed.h2_overflow) == 0 and agent.h2_calibrated
    assert agent._updates_issued == issued0 + 6 and int(agent.packed.step) == issued0 + 6
    assert not torch.equal(agent.packed.P, p_before) and all(torch.isfinite(q).all() for q in agent.q.parameters())
    agent.exit()


def test_h2_dw2_from_the_record_against_the_image_form_and_float64():
    """The default dW2 kernel rebuilds dZ2 from a 2.3 KB record per tile (dq s_z2 and the action per row, LeakyReLU' flags) instead of
    reading its 32 KB plane image: the layer-2 weight gradient of both forms against float64 (<= 2e-6 of its scale each: the rebuilt
    dZ2 is the correctly rounded fp32 product, the image a two-term fp16 split of the chain's) and against each other; every other
    block of the gradient bit-equal (only `dqn_dw2` differs); 5 x 4096 rows: workgroups walk tiles across chunk boundaries."""
    torch.manual_seed(12)
    raw = _batch(5, 4096, 31)
    d = _bare_dqn(rows=4096, fused=True)
    _perturb_target(d)
    chunks, _ = _unambiguous(d, raw)
    want, _ = _grad64(d, chunks)
    st = _state(d)
    d.update(chunks); _restore(d, st)                    # calibrated
    d.h2_freeze = True
    got = {}
    for recon in (True, False):
        d.dw2_recon = recon
        d.update(chunks); torch.cuda.synchronize()
        got[recon] = d.packed.G.clone()
        _restore(d, st)
    w2 = want[2]
    scale = float(w2.abs().max())
    for recon in (True, False):
        g2 = _views(got[recon])[2].double()
        assert float((g2 - w2).abs().max()) <= 2e-6 * scale, recon
    a, b = got[True].clone(), got[False].clone()
    assert float((_views(a)[2] - _views(b)[2]).abs().max()) <= 2e-6 * scale and not torch.equal(_views(a)[2], _views(b)[2])
    _views(a)[2].zero_(); _views(b)[2].zero_()
    assert torch.equal(a, b)
