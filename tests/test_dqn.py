"""DQN variant (reference UselessFiles/dqn.py, configs[4]): oracle vs the golden vectors recorded
from the reference's own DQN.act / DQN.update (CPU), HIP kernels vs both (GPU)."""
import contextlib
import io

import numpy as np
import pytest
import torch

from oracle import oracle as O


@pytest.mark.parametrize("tag,eps", [("e08", 0.8), ("e001", 0.01)])
def test_oracle_eps_greedy_matches_reference(golden, tag, eps):
    g = golden("g8_dqn")
    act = O.dqn_eps_greedy(g[tag + "_q"], g[tag + "_coin_u"], g[tag + "_rand_u"], eps)
    assert np.array_equal(act, g[tag + "_act"])
    if tag == "e08":
        assert 0.3 < (g[tag + "_coin_u"] < eps).mean() < 1.0


def test_oracle_huber_td_matches_reference(golden):
    g = golden("g8_dqn")
    dq, loss = O.dqn_huber_td(g["q_table"], g["b_act"], g["b_rew"], g["q_next"], g["b_done"])
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=1e-6)
    np.testing.assert_allclose(dq, g["dq"], rtol=1e-6, atol=1e-9)
    assert (np.abs(g["q_table"][:, 0] - g["b_rew"]) > 1).any()        # both Huber branches exercised


@pytest.mark.gpu
@pytest.mark.parametrize("tag,eps", [("e08", 0.8), ("e001", 0.01)])
def test_hip_eps_greedy_bit_exact(golden, tag, eps):
    from fly_bproject_amd import _lib
    from tests.hip_helpers import cuda
    g = golden("g8_dqn")
    q, cu, ru = cuda(g[tag + "_q"]), cuda(g[tag + "_coin_u"]), cuda(g[tag + "_rand_u"])
    out = torch.empty(q.shape[0], device="cuda:0")
    _lib.check(_lib.load().dqn_eps_greedy(q.data_ptr(), cu.data_ptr(), ru.data_ptr(), eps, 18, out.data_ptr(),
                                          q.shape[0], None), "eps")
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), g[tag + "_act"])
    # ties: the FIRST maximal entry wins (dqn.py:95)
    qt = torch.zeros(4, 18, device="cuda:0"); qt[1, 5] = qt[1, 9] = 2.0; qt[2, 17] = 1.0
    z = torch.ones(4, device="cuda:0")
    _lib.check(_lib.load().dqn_eps_greedy(qt.data_ptr(), z.data_ptr(), z.data_ptr(), 0.0, 18, out.data_ptr(), 4, None), "eps")
    np.testing.assert_allclose(out[:3].cpu().numpy(), [2 * (0 / 17 - 0.5), 2 * (5 / 17 - 0.5), 1.0], rtol=1e-6)


@pytest.mark.gpu
def test_hip_huber_td_vs_reference_and_oracle(golden):
    from fly_bproject_amd import _lib
    from tests.hip_helpers import cuda
    g = golden("g8_dqn")
    B = g["q_table"].shape[0]
    qt, act, rew, qn, dn = cuda(g["q_table"]), cuda(g["b_act"]), cuda(g["b_rew"]), cuda(g["q_next"]), cuda(g["b_done"])
    dq = torch.empty_like(qt); parts = torch.empty((B + 255) // 256, device="cuda:0")
    _lib.check(_lib.load().dqn_huber_td(qt.data_ptr(), act.data_ptr(), rew.data_ptr(), qn.data_ptr(), dn.data_ptr(),
                                        0.99, 18, B, dq.data_ptr(), parts.data_ptr(), None), "td")
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(parts.sum() / B), float(g["loss"]), rtol=2e-6)
    np.testing.assert_allclose(dq.cpu().numpy(), g["dq"], rtol=1e-6, atol=1e-9)
    dq2, loss2 = O.dqn_huber_td(g["q_table"], g["b_act"], g["b_rew"], g["q_next"], g["b_done"])
    np.testing.assert_allclose(dq.cpu().numpy(), dq2, rtol=3e-7, atol=0)     # d/B vs d*(1/B): one ulp


def _bare_dqn(golden_sd=None, rows=384, fused=False, gemm="f16x2"):
    """A DQN without an environment: the packed Q-network pair + workspace on cuda:0.  `gemm`: the fused update's arithmetic
    (f16x2, the default of DQN, or bf16x3)."""
    from fly_bproject_amd import _lib
    from fly_bproject_amd.dqn import DQN, Net, QNetPacked, soft_update
    d = DQN.__new__(DQN)
    d.device = torch.device("cuda:0")
    d.discount, d.mini_batch_size, d.tau, d.act_space, d.lr = 0.99, 4, 0.995, 18, 3e-4
    d.q, d.q_target = Net(73, 18).to("cuda:0"), Net(73, 18).to("cuda:0")
    if golden_sd is not None:
        d.q.load_state_dict(golden_sd)
    soft_update(d.q, d.q_target, tau=0.0)
    d._lib = _lib.load()
    d.packed = QNetPacked(d.q, d.q_target, "cuda:0")
    d._alloc_workspace(rows)
    d.fused_update = fused
    d.update_gemm, d.h2_calibrated, d.h2_freeze, d.h2_overflows = gemm, False, False, 0
    d._updates_issued, d._h2_use_b3, d._h2_guard = 0, False, False
    import os
    d.dw2_recon = os.environ.get("FLY_DQN_DW2_RECON", "1") != "0"      # (DQN.__init__'s switch: the suite runs under either setting)
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 33, 96, 4099])
def test_mfma_q_network_forward_matches_torch(golden, n):
    """dqn_forward (fp32 MFMA, packed fragment-ordered weights) against torch's fp32 evaluation of the same
    module, on the reference's golden weights; ragged last tile included."""
    g = golden("g8_dqn")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")   # noqa: E731
    d = _bare_dqn({k[2:]: t(g[k]) for k in g.files if k.startswith("q_") and "." in k})
    torch.manual_seed(n)
    x = torch.randn(n, 73, device="cuda:0") * 2
    if n == 96:
        x = t(g["obs"])
    got = d.q_values(x)
    with torch.no_grad():
        h = torch.nn.functional.leaky_relu(x.double() @ d.q.net[0].weight.double().T + d.q.net[0].bias.double())
        h = torch.nn.functional.leaky_relu(h @ d.q.net[2].weight.double().T + d.q.net[2].bias.double())
        ref = h @ d.q.net[4].weight.double().T + d.q.net[4].bias.double()
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-5)
    if n == 96:     # and the Q tables the reference recorded for these observations
        np.testing.assert_allclose(got.cpu().numpy(), g["e08_q"], rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("eps", [0.0, 0.8])
def test_fused_act_equals_forward_plus_eps_greedy(golden, eps):
    """dqn_act (forward + first-argmax + eps mix in one launch) == dqn_forward followed by the oracle's
    dqn.py:89-100 on that Q table, bit for bit; ties resolve to the FIRST maximal entry."""
    g = golden("g8_dqn")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")   # noqa: E731
    d = _bare_dqn({k[2:]: t(g[k]) for k in g.files if k.startswith("q_") and "." in k})
    n = 4099
    torch.manual_seed(3)
    x = torch.randn(n, 73, device="cuda:0")
    x[7] = x[8]                                                   # identical rows -> identical actions
    d._gen = torch.Generator(device="cuda:0"); d._gen.manual_seed(1)
    d._coin = torch.empty(n, device="cuda:0"); d._rand = torch.empty(n, device="cuda:0")
    a = d.act(x, eps)
    q = d.q_values(x)
    torch.cuda.synchronize()
    ref = O.dqn_eps_greedy(q.cpu().numpy(), d._coin.cpu().numpy(), d._rand.cpu().numpy(), eps)
    assert np.array_equal(a.cpu().numpy(), ref)
    # a constant last layer makes every Q row a 18-way tie: index 0 must win
    with torch.no_grad():
        d.q.net[4].weight.zero_(); d.q.net[4].bias.fill_(0.25)
    d.packed.refresh()
    a0 = d.act(x, 0.0)
    assert torch.all(a0 == -1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, "bf16x3", "f16x2"])
def test_dqn_update_matches_reference_step(golden, fused):
    """One DQN.update from the reference's weights on the reference's batch, all on the HIP kernels
    (dqn_td_step + dqn_grad_w + dqn_adam_soft_update): same loss, same Q-network and target network
    afterwards (Adam 3e-4 + soft update 0.995) -- as ONE chunk and as four chunks whose gradients accumulate."""
    g = golden("g8_dqn")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")   # noqa: E731
    batch = (t(g["b_obs"]), t(g["b_act"]), t(g["b_rew"]), t(g["b_next"]), t(g["b_done"]))
    B = batch[0].shape[0]
    assert B % 128 == 0                      # whole 32-row tiles in four chunks too: the fused launches take both splits
    for parts in (1, 4):
        d = _bare_dqn({k[2:]: t(g[k]) for k in g.files if k.startswith("q_") and "." in k}, fused=bool(fused), gemm=fused or "f16x2")
        step = B // parts
        chunks = [tuple(x[i * step:(i + 1) * step].contiguous() for x in batch) for i in range(parts)]
        loss = d.update(chunks)
        torch.cuda.synchronize()
        np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=2e-5)
        assert int(d.packed.step) == 1
        for k, v in d.q.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), g["q1_" + k], rtol=2e-4, atol=3e-5, err_msg=k)
        for k, v in d.q_target.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), g["qt1_" + k], rtol=2e-4, atol=1e-6, err_msg=k)
        # padding never moves, fragment copies follow the masters
        assert torch.all(d.packed.P[:256 * 80].view(256, 80)[:, 73:] == 0)
        pf, pt, pft = d.packed.PF.clone(), d.packed.PT.clone(), d.packed.PF_tgt.clone()
        qb, qtb, qbt = d.packed.QB.clone(), d.packed.QTB.clone(), d.packed.QB_tgt.clone()
        d.packed.refresh()
        assert torch.equal(pf, d.packed.PF) and torch.equal(pt, d.packed.PT) and torch.equal(pft, d.packed.PF_tgt)
        assert torch.equal(qb, d.packed.QB) and torch.equal(qtb, d.packed.QTB) and torch.equal(qbt, d.packed.QB_tgt)   # the bf16x3 planes too
        assert d.h2_overflows == 0 and d.h2_calibrated == (fused == "f16x2")


@pytest.mark.gpu
@pytest.mark.parametrize("fused,B", [(False, 1000), ("bf16x3", 1024), ("bf16x3", 16384), ("f16x2", 1024), ("f16x2", 16384)])
def test_dqn_td_gradient_matches_autograd(fused, B):
    """The packed gradient of one update batch (TD target from the target net, Huber, backward, dW) against
    torch autograd on the same module in fp64-free fp32; random weights; 1000 rows (ragged tile) through the per-step
    launches, whole tiles through the fused launches in both arithmetics (16384 rows: every workgroup walks two tiles)."""
    torch.manual_seed(5)
    d = _bare_dqn(rows=B, fused=bool(fused), gemm=fused or "f16x2")
    with torch.no_grad():
        for p_ in d.q_target.parameters():
            p_.add_(0.05 * torch.randn_like(p_))                 # target differs from online
    d.packed.refresh()
    obs = torch.randn(B, 73, device="cuda:0"); nxt = torch.randn(B, 73, device="cuda:0")
    act = torch.rand(B, device="cuda:0") * 2 - 1
    rew = torch.randn(B, device="cuda:0") * 2; done = (torch.rand(B, device="cuda:0") > 0.1).float()
    q_table = d.q(obs)
    idx = torch.round(0.5 * (act + 1) * 17).long()
    q_val = q_table[torch.arange(B), idx]
    with torch.no_grad():
        target = rew + 0.99 * d.q_target(nxt).max(1)[0] * done
    loss = torch.nn.functional.smooth_l1_loss(q_val, target)
    grads = torch.autograd.grad(loss, list(d.q.parameters()))
    before = d.packed.P.clone()
    got_loss = d.update([(obs, act, rew, nxt, done)])
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(got_loss), float(loss), rtol=2e-5)
    G = d.packed.G
    views = {"net.0.weight": G[:256 * 80].view(256, 80)[:, :73], "net.0.bias": G[20480:20736],
             "net.2.weight": G[20736:86272].view(256, 256), "net.2.bias": G[86272:86528],
             "net.4.weight": G[86528:94720].view(32, 256)[:18], "net.4.bias": G[94720:94738]}
    for (name, _), want in zip(d.q.named_parameters(), grads):
        got = views[name]
        scale = float(want.abs().max()) + 1e-12
        assert float((got - want).abs().max()) <= 2e-4 * scale + 1e-9, name
    assert not torch.equal(before, d.packed.P)
    assert d.h2_overflows == 0


@pytest.mark.gpu
@pytest.mark.parametrize("gemm", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("parts,n,skew", [(5, 4096, 0), (5, 4096, 1), (3, 96, 1), (1, 32, 0)])
def test_dqn_fused_update_over_chunks_and_misaligned_rows(parts, n, skew, gemm):
    """dqn_fused_update on several sampled steps: 5 x 4096 rows = 640 tiles, so workgroups walk up to three tiles across chunk
    boundaries (rows and scalars of the NEXT tile are requested a tile ahead); `skew` = 1 puts every row block one row into its
    allocation (292 bytes: not 16-byte aligned), which takes the plain-load path instead of LDS-DMA; 32 rows = one tile, one
    workgroup.  Against torch autograd on the concatenated batch."""
    torch.manual_seed(11)
    B = parts * n
    d = _bare_dqn(rows=B, fused=True, gemm=gemm)
    with torch.no_grad():
        for p_ in d.q_target.parameters():
            p_.add_(0.05 * torch.randn_like(p_))
    d.packed.refresh()
    chunks = []
    for _ in range(parts):
        obs = torch.randn(n + skew, 73, device="cuda:0")[skew:]; nxt = torch.randn(n + skew, 73, device="cuda:0")[skew:]
        act = (torch.rand(n + skew, device="cuda:0") * 2 - 1)[skew:]
        rew = (torch.randn(n + skew, device="cuda:0") * 2)[skew:]; done = (torch.rand(n + skew, device="cuda:0") > 0.1).float()[skew:]
        assert obs.is_contiguous() and (skew == 0) == (obs.data_ptr() % 16 == 0)
        chunks.append((obs, act, rew, nxt, done))
    obs, act, rew, nxt, done = (torch.cat([c[i] for c in chunks]) for i in range(5))
    idx = torch.round(0.5 * (act + 1) * 17).long()
    q_val = d.q(obs)[torch.arange(B), idx]
    with torch.no_grad():
        target = rew + 0.99 * d.q_target(nxt).max(1)[0] * done
    loss = torch.nn.functional.smooth_l1_loss(q_val, target)
    grads = torch.autograd.grad(loss, list(d.q.parameters()))
    got_loss = d.update(chunks)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(got_loss), float(loss), rtol=2e-5)
    G = d.packed.G
    views = {"net.0.weight": G[:256 * 80].view(256, 80)[:, :73], "net.0.bias": G[20480:20736],
             "net.2.weight": G[20736:86272].view(256, 256), "net.2.bias": G[86272:86528],
             "net.4.weight": G[86528:94720].view(32, 256)[:18], "net.4.bias": G[94720:94738]}
    for (name, _), want in zip(d.q.named_parameters(), grads):
        scale = float(want.abs().max()) + 1e-12
        assert float((views[name] - want).abs().max()) <= 2e-4 * scale + 1e-9, name
    assert d.h2_overflows == 0


@pytest.mark.gpu
@pytest.mark.parametrize("gemm", ["bf16x3", "f16x2"])
def test_dqn_fused_update_at_full_size_is_deterministic_and_linear(gemm):
    """BASELINE configs[4]'s row count per sampled step (32768), eight sampled steps: (i) two runs on the same inputs leave the same
    packed gradient and per-tile loss sums bit for bit (fixed-order reductions everywhere: no atomics); (ii) the gradient of the batch
    is the mean of the gradients of its halves (the update is a sum over rows scaled by 1 / B: any row lost or doubled at a tile,
    chunk or workgroup boundary breaks this), to summation-order rounding; (iii) it agrees with the per-step fp32-MFMA launches."""
    torch.manual_seed(3)
    n, parts = 32768, 8
    d = _bare_dqn(rows=n, fused=True, gemm=gemm)
    with torch.no_grad():
        for p_ in d.q_target.parameters():
            p_.add_(0.05 * torch.randn_like(p_))
    d.packed.refresh()
    chunks = [(torch.randn(n, 73, device="cuda:0"), torch.rand(n, device="cuda:0") * 2 - 1, torch.randn(n, device="cuda:0") * 2,
               torch.randn(n, 73, device="cuda:0"), (torch.rand(n, device="cuda:0") > 0.1).float()) for _ in range(parts)]
    p0 = [t.clone() for t in (d.packed.P, d.packed.P_tgt, d.packed.exp_avg, d.packed.exp_avg_sq, d.packed.step)]

    def restore():      # the update also steps Adam and the target: put the networks back
        for dst, src in zip((d.packed.P, d.packed.P_tgt, d.packed.exp_avg, d.packed.exp_avg_sq, d.packed.step), p0):
            dst.copy_(src)
        d.packed.refresh()

    def grad(cs, fused=True):
        restore()
        d.fused_update = fused
        loss = d.update(cs)
        torch.cuda.synchronize()
        return d.packed.G.clone(), float(loss)
    if gemm == "f16x2":     # the first update calibrates the lagged scales; from there on they stay as they are (they are a launch's INPUT:
        grad(chunks)        # two launches with different scales differ in rounding, two with the same scales must not differ at all)
        d.h2_freeze = True
    g1, l1 = grad(chunks)
    g2, l2 = grad(chunks)
    assert torch.equal(g1, g2) and l1 == l2 and d.h2_overflows == 0
    ga, la = grad(chunks[:parts // 2])
    gb, lb = grad(chunks[parts // 2:])
    m = d.packed.grad_mask > 0
    scale = float(g1[m].abs().max())
    assert float((0.5 * (ga + gb) - g1)[m].abs().max()) <= 2e-5 * scale
    assert abs(0.5 * (la + lb) - l1) <= 2e-5 * abs(l1)
    g3, l3 = grad(chunks, fused=False)
    assert float((g3 - g1)[m].abs().max()) <= 2e-4 * scale and abs(l3 - l1) <= 2e-5 * abs(l1)


@pytest.mark.gpu
def test_dqn_runs_end_to_end():
    from fly_bproject_amd.dqn import DQN
    from tests.hip_helpers import make_args
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        agent = DQN(make_args(256, dqn_mini_batch_size=8, replay_steps=16))
    assert agent.replay.capacity == 16
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(30):
            agent.run()
    torch.cuda.synchronize()
    assert agent.replay.size() == 16 and agent.last_loss is not None       # the ring wrapped
    assert int(agent.packed.step) == 30 - 8                                 # an update per step once size > 8
    assert torch.isfinite(agent.last_loss) and all(torch.isfinite(p).all() for p in agent.q.parameters())
    a = agent.act(agent.env.obs_buf, 0.0)
    assert a.shape == (256,) and float(a.abs().max()) <= 1.0
    agent.exit()


@pytest.mark.gpu
def test_dqn_fused_update_with_more_chunks_over_no_more_rows():
    """`DQN.update(chunks)` is a public entry that callers give varying splits: 2 x 4096 rows, then 8 x 1024 -- MORE chunks over no more
    rows.  The chunk tables have a capacity of their own (round 4's advisor: they used to be regrown only when the row count grew, and
    the second call walked DqnChunk records past the end of a table sized for two).  Both calls against torch autograd."""
    torch.manual_seed(5)
    d = _bare_dqn(rows=8192, fused=True)
    for parts, n in ((2, 4096), (8, 1024), (3, 2048)):
        chunks = [(torch.randn(n, 73, device="cuda:0"), torch.rand(n, device="cuda:0") * 2 - 1, torch.randn(n, device="cuda:0") * 2,
                   torch.randn(n, 73, device="cuda:0"), (torch.rand(n, device="cuda:0") > 0.1).float()) for _ in range(parts)]
        obs, act, rew, nxt, done = (torch.cat([c[i] for c in chunks]) for i in range(5))
        B = parts * n
        idx = torch.round(0.5 * (act + 1) * 17).long()
        q_val = d.q(obs)[torch.arange(B), idx]
        with torch.no_grad():
            target = rew + 0.99 * d.q_target(nxt).max(1)[0] * done
        loss = torch.nn.functional.smooth_l1_loss(q_val, target)
        got_loss = d.update(chunks)
        torch.cuda.synchronize()
        np.testing.assert_allclose(float(got_loss), float(loss), rtol=2e-5)
        assert d._fu_S >= parts and d._fu_tab[0][0].shape[0] >= parts
