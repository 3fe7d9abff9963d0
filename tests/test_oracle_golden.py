"""CPU: the oracle (oracle/) against the golden vectors recorded from the reference's own code
(tests/golden/gen_golden.py).  This is what pins the checker before it is trusted on the GPU."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from oracle import ppo_oracle as PO

NA = 18


def _state(g, p, n):
    s = O.EnvState(n)
    s.root[:] = g[p + "root"]
    s.dof_pos[:] = g[p + "dof_pos"]
    s.dof_vel[:] = g[p + "dof_vel"]
    s.targets[:] = g[p + "targets"]
    s.contact[:] = g[p + "contact"]
    s.pot[:] = g[p + "pot_in"]
    return s


@pytest.mark.parametrize("n", [16, 257])
def test_obs_pack_matches_reference(golden, n):
    g = golden("g1_obs")
    p = "n%d_" % n
    cfg = O.default_config(n)
    s = _state(g, p, n)
    up, hd = O.pack_obs(cfg, s, want_vecs=True)
    ref = g[p + "obs"]
    # exact copies / integer-valued columns are bit-exact
    for col in [0] + list(range(48, 66)) + list(range(67, 73)):
        assert np.array_equal(s.obs[:, col], ref[:, col]), col
    assert np.array_equal(s.prev_pot, g[p + "prev_pot"])
    # arithmetic columns: fp32 tolerance (torch's vectorised atan2/asin/bmm vs libm)
    np.testing.assert_allclose(s.obs, ref, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(s.pot, g[p + "pot"], rtol=1e-6)
    np.testing.assert_allclose(up, g[p + "up_vec"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(hd, g[p + "heading_vec"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("n", [16, 257])
@pytest.mark.parametrize("variant,ecs", [("big", 0.005), ("low", 1.0)])
def test_reward_done_matches_reference(golden, n, variant, ecs):
    g = golden("g2_reward")
    p = "n%d_" % n
    cfg = O.default_config(n)
    cfg.energy_cost_scale = ecs
    s = O.EnvState(n)
    s.obs[:] = g[p + "obs"]
    s.targets[:] = g[p + "targets"]
    s.root[:] = g[p + "root"]
    s.contact[:] = g[p + "contact"]
    s.pot[:] = g[p + "pot"]
    s.prev_pot[:] = g[p + "prev_pot"]
    s.progress[:] = g[p + "progress"]
    s.reset[:] = g[p + "reset_in"]
    O.pack_reward(cfg, s)
    assert np.array_equal(s.reset, g[p + variant + "_reset"])          # done mask: bit-exact
    np.testing.assert_allclose(s.reward, g[p + variant + "_reward"], rtol=2e-6, atol=2e-6)
    assert 0 < s.reset.sum() < n


@pytest.mark.parametrize("variant", ["bigGrav", "lowGrav"])
def test_step_orchestration_matches_reference(golden, variant):
    """Fly.step ordering, reset semantics and progress counting (fly.py:624-681, :446-480;
    flyLowGrav.py:657-663) with FlyDyn plugged in on both sides."""
    g = golden("g3_step")
    acts = g[variant + "_actions"]
    steps, n, _ = acts.shape
    cfg = O.default_config(n, variant)
    s = O.EnvState(n)
    resets = 0
    for t in range(steps):
        O.env_step(cfg, s, acts[t])
        if t == 20:
            s.progress[3] = 1497
        assert np.array_equal(s.root, g[variant + "_root"][t]), t       # same C physics both sides
        assert np.array_equal(s.dof_pos, g[variant + "_dof_pos"][t]), t
        assert np.array_equal(s.dof_vel, g[variant + "_dof_vel"][t]), t
        assert np.array_equal(s.reset, g[variant + "_reset"][t]), t
        assert np.array_equal(s.progress, g[variant + "_progress"][t]), t
        np.testing.assert_allclose(s.obs, g[variant + "_obs"][t], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(s.reward, g[variant + "_reward"][t], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(s.pot, g[variant + "_pot"][t], rtol=1e-6)
        np.testing.assert_allclose(s.prev_pot, g[variant + "_prev_pot"][t], rtol=1e-6)
        resets += int(s.reset.sum())
    assert resets > 0, "fixture must exercise the reset path"


def test_reset_state_values():
    """fly.py:446-480: reset pose, potentials -|(1000,0,0)|/dt, cleared flags."""
    cfg = O.default_config(5)
    s = O.EnvState(5)
    s.root[:] = 7.0
    s.dof_vel[:] = 3.0
    s.reset[:] = [1, 0, 1, 0, 0]
    s.progress[:] = 9
    assert O.reset_masked(cfg, s) == 2
    assert np.array_equal(s.root[0], np.array([0, 0, 2, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], np.float32))
    assert np.all(s.root[1] == 7.0)
    np.testing.assert_allclose(s.dof_pos[2], np.array(cfg.dof_pose[:], np.float32))
    assert np.all(s.dof_vel[0] == 0) and np.all(s.dof_vel[1] == 3.0)
    assert s.pot[0] == np.float32(-1000.0) / np.float32(1 / 60) and s.prev_pot[0] == s.pot[0]
    assert list(s.reset) == [0, 0, 0, 0, 0] and list(s.progress) == [0, 9, 0, 9, 9]


def test_net_forward_matches_reference(golden):
    g = golden("g4_net")
    sd = {k: g[k] for k in g.files if "." in k}
    pi = O.net_forward(sd, g["x"], 0)
    v = O.net_forward(sd, g["x"], 1)
    np.testing.assert_allclose(pi, g["pi"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(v, g["v"], rtol=1e-5, atol=1e-5)
    net = PO.OracleNet()
    net.load_state_dict({k: torch.from_numpy(a) for k, a in sd.items()})   # reference key names
    with torch.no_grad():
        np.testing.assert_allclose(net.pi(torch.from_numpy(g["x"])).numpy(), g["pi"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(net.v(torch.from_numpy(g["x"])).numpy(), g["v"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("tag", ["v02", "v001", "vmix"])
def test_sample_logprob_matches_reference(golden, tag):
    g = golden("g5_sample")
    act, logp = O.sample_logprob(g["mu"], g[tag + "_var"], g[tag + "_eps"])
    assert np.array_equal(act, g[tag + "_clipped"]) or np.allclose(act, g[tag + "_clipped"], atol=1e-7)
    np.testing.assert_allclose(logp, g[tag + "_logp"], rtol=2e-6, atol=2e-5)
    lp = PO.diag_gauss_logprob(torch.from_numpy(g["mu"]), torch.from_numpy(g[tag + "_clipped"]),
                               torch.from_numpy(g[tag + "_var"]))
    np.testing.assert_allclose(lp.numpy(), g[tag + "_logp_clipped"], rtol=2e-6, atol=2e-5)
    assert (np.abs(g[tag + "_action"]) > 1).any(), "fixture must exercise the clip"


@pytest.mark.parametrize("tag", ["a", "b"])
def test_td_gae_matches_reference(golden, tag):
    g = golden("g6_gae")
    r, v, vn, d = g[tag + "_reward"][..., 0], g[tag + "_v"][..., 0], g[tag + "_v_next"][..., 0], g[tag + "_done"][..., 0]
    target, adv = O.td_gae(r, v, vn, d.astype(np.float32))
    assert np.array_equal(target, g[tag + "_target"][..., 0])       # same fp32 ops in the same order
    assert np.array_equal(adv, g[tag + "_adv"][..., 0])


def test_make_data_torch_oracle(golden):
    g4, g = golden("g4_net"), golden("g6_gae")
    net = PO.OracleNet()
    net.load_state_dict({k: torch.from_numpy(g4[k]) for k in g4.files if "." in k})
    t = torch.from_numpy
    target, adv = PO.make_data(net, t(g["a_obs"]), t(g["a_next_obs"]), t(g["a_reward"]), t(g["a_done"]))
    np.testing.assert_allclose(target.numpy(), g["a_target"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(adv.numpy(), g["a_adv"], rtol=1e-5, atol=1e-5)


def test_ppo_update_matches_reference(golden):
    """75 optimizer steps of ppo.py:173-202 from the same initial weights end at the same weights."""
    g = golden("g7_update")
    t = torch.from_numpy
    net = PO.OracleNet()
    net.load_state_dict({k[3:]: t(g[k]) for k in g.files if k.startswith("w0_")})
    optim = torch.optim.Adam(net.parameters(), lr=1e-3)
    target, adv = PO.make_data(net, t(g["obs"]), t(g["next_obs"]), t(g["reward"]), t(g["done"]))
    np.testing.assert_allclose(target.numpy(), g["target"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(adv.numpy(), g["adv"], rtol=1e-5, atol=1e-5)
    seen = {}

    def on_step(i, loss, gn):
        if i == 0:
            seen["loss"], seen["gn"] = float(loss), float(gn)
            seen["grads"] = {k: p.grad.clone() for k, p in net.named_parameters()}
    n = PO.update(net, optim, t(g["obs"]), t(g["acts"]), t(g["log_prob"]), target, adv,
                  t(g["action_var"]), mini_chunk_size=2, rollout_size=32, on_step=on_step)
    assert n == 75                                                   # log.txt:48-49
    np.testing.assert_allclose(seen["loss"], float(g["loss0"]), rtol=1e-5)
    np.testing.assert_allclose(seen["gn"], float(g["gradnorm0"]), rtol=1e-4)
    for k, p in net.state_dict().items():
        np.testing.assert_allclose(p.numpy(), g["w1_" + k], rtol=2e-3, atol=2e-4, err_msg=k)


def test_physics_f32_vs_f64_tolerance():
    """FlyDyn is build-defined (parity unpinned vs PhysX).  Stated fp32 tolerance: one env step
    (15 substeps) from the same state stays within 2e-4 (abs+rel) of the fp64 evaluation for
    positions/quaternions/joint angles and 5e-3 for velocities (stiff contact amplifies rounding)."""
    n = 64
    rng = np.random.default_rng(3)
    cfg = O.default_config(n)
    s = O.EnvState(n)
    lo = np.array(cfg.dof_lo[:]); hi = np.array(cfg.dof_hi[:]); pose = np.array(cfg.dof_pose[:])
    a0 = ((2 * pose - hi - lo) / (hi - lo)).astype(np.float32)
    for t in range(30):
        a = np.clip(a0 + rng.normal(0, 0.4, (n, NA)), -1, 1).astype(np.float32)
        O.env_step(cfg, s, a)
        if t >= 10:
            s2 = s.copy()
            s2.targets[:] = O.scale_actions(cfg, a)
            r64, q64, qd64, c64 = O.physics_step_f64(cfg, s2.root, s2.dof_pos, s2.dof_vel, s2.targets)
            O.physics_step(cfg, s2)
            np.testing.assert_allclose(s2.root[:, :7], r64[:, :7], rtol=2e-4, atol=2e-4)
            np.testing.assert_allclose(s2.root[:, 7:], r64[:, 7:], rtol=5e-3, atol=5e-3)
            np.testing.assert_allclose(s2.dof_pos, q64, rtol=2e-4, atol=2e-4)
    assert np.isfinite(s.root).all()


# ---- property tests: the GAE recurrence and the DQN / replay host logic -------------------
def test_gae_recurrence_properties():
    """ppo.py:157-171 as the oracle restates it, against an independent float64 evaluation of the recurrence
    adv_t = gamma*lambda*adv_{t+1} + delta_t (adv_T = 0, NO reset at episode ends: Q2) on random shapes, plus its
    structural properties: linearity in the rewards and independence of the envs."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=40, deadline=None)
    @given(st.integers(1, 70), st.integers(1, 33), st.integers(0, 2 ** 31 - 1))
    def check(T, n, seed):
        rng = np.random.default_rng(seed)
        r = rng.normal(size=(T, n)).astype(np.float32)
        v = rng.normal(size=(T, n)).astype(np.float32)
        vn = rng.normal(size=(T, n)).astype(np.float32)
        done = (rng.random(n) > 0.3).astype(np.float32)                    # Q1: ONE [N] mask, broadcast over T
        target, adv = O.td_gae(r, v, vn, done)
        t64 = r.astype(np.float64) + 0.99 * vn.astype(np.float64) * done
        delta = t64 - v
        ref = np.zeros((T, n))
        acc = np.zeros(n)
        for t in range(T - 1, -1, -1):
            acc = 0.99 * 0.95 * acc + delta[t]
            ref[t] = acc
        np.testing.assert_allclose(target, t64, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(adv, ref, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(ref).max()))
        # envs are independent: a column of the batch alone gives that column
        j = int(rng.integers(0, n))
        t1, a1 = O.td_gae(r[:, j:j + 1].copy(), v[:, j:j + 1].copy(), vn[:, j:j + 1].copy(), done[j:j + 1].copy())
        assert np.array_equal(t1[:, 0], target[:, j]) and np.array_equal(a1[:, 0], adv[:, j])
        # linear in (r, v, v_next): doubling all three doubles both outputs exactly (powers of two are exact)
        t2, a2 = O.td_gae(2 * r, 2 * v, 2 * vn, done)
        assert np.array_equal(t2, 2 * target) and np.array_equal(a2, 2 * adv)
    check()


def test_dqn_packed_layout_maps():
    """Host-side layout logic of the DQN Q-network (fly_bproject_amd/dqn.py, csrc/dqn_layout.h): the fragment index
    maps are injective and cover their buffers exactly; layer 3's forward map is the split-K order the kernel reads."""
    from fly_bproject_amd import dqn as D
    idx_f, idx_t = D.build_index_maps()
    assert (idx_f >= 0).sum() == D.FRAG == 94208 and (idx_t >= 0).sum() == D.FRAG_T == 73728
    assert D.PACKED == 94752
    for idx, size in ((idx_f, D.FRAG), (idx_t, D.FRAG_T)):
        used = idx[idx >= 0]
        assert len(np.unique(used)) == len(used) and used.min() == 0 and used.max() == size - 1
    # biases have no fragment copy
    for off, n in ((D.OFF_B1, D.H), (D.OFF_B2, D.H), (D.OFF_B3, D.OUT)):
        assert np.all(idx_f[off:off + n] == -1) and np.all(idx_t[off:off + n] == -1)
    # W3[n][k] -> ((w*8 + kq)*64 + (h*32 + n))*4 + q with k = 64 w + 32 h + 4 kq + q
    n, k = 5, 64 * 2 + 32 * 1 + 4 * 3 + 2
    assert idx_f[D.OFF_W3 + n * D.H + k] == D.OFF_F3 + ((2 * 8 + 3) * 64 + (1 * 32 + n)) * 4 + 2


def test_replay_ring_sampling_logic():
    """replay.py:10-31 as a ring: capacity in STEPS, wrap-around, distinct sampled slots, host-side draw."""
    import torch
    from fly_bproject_amd.dqn import ReplayBuffer
    rb = ReplayBuffer(num_envs=4, num_obs=73, device="cpu", buffer_limit=5, seed=3)
    assert rb.capacity == 5 and rb.bytes == 5 * 4 * (2 * 73 + 3) * 4
    for i in range(8):
        o = torch.full((4, 73), float(i))
        rb.push(o, torch.full((4,), float(i)), torch.full((4,), float(i)), o + 0.5, torch.ones(4))
    assert rb.size() == 5 and rb.head == 3
    held = sorted(float(rb.action[s][0]) for s in range(5))
    assert held == [3.0, 4.0, 5.0, 6.0, 7.0]                                   # the five newest steps
    slots = rb.sample_slots(4)
    assert len(set(slots)) == 4 and all(0 <= s < 5 for s in slots)
    chunks = rb.sample(3)
    assert len(chunks) == 3 and all(c[0].shape == (4, 73) and c[0].data_ptr() >= rb.obs.data_ptr() for c in chunks)   # views, no copy
    big = ReplayBuffer(num_envs=32768, num_obs=73, device="meta", buffer_limit=None)
    assert big.capacity == 512 and abs(big.bytes / 1e9 - 10.0) < 0.01        # the stated default: 512 steps = 10.0 GB


def _g2_state(g, n):
    p = "n%d_" % n
    s = O.EnvState(n)
    s.obs[:] = g[p + "obs"]; s.targets[:] = g[p + "targets"]; s.root[:] = g[p + "root"]
    s.contact[:] = g[p + "contact"]; s.pot[:] = g[p + "pot"]; s.prev_pot[:] = g[p + "prev_pot"]
    s.progress[:] = g[p + "progress"]; s.reset[:] = g[p + "reset_in"]
    return s


@pytest.mark.parametrize("n", [16, 257])
def test_reward_term_dump_is_consistent_with_the_reference_reward(golden, n):
    """orc_reward_terms restates the viewer's P-key dump (fly.py:504-546), which no reference run can execute here (it lives in
    the Isaac Gym viewer's event loop).  What CAN be pinned: its electricity, dof-at-limit and leg terms are the reward's own
    (fly.py:733-744), so on g2's inputs the reference's recorded reward of every env that is not dead equals
    0.5 + up_r * orient_r - scale * electricity_cost - dof_at_limit_cost + leg_reward with those terms -- for both cost
    scales.  The dump's OWN up / orient terms differ from the reward's (no "below 2.1" branch, :512-513; threshold 0.92, :518,
    against 0.98, :728): checked against their definitions, with envs inside the 0.92 .. 0.98 band present."""
    g = golden("g2_reward")
    s = _g2_state(g, n)
    t = O.reward_terms(O.default_config(n), s)
    assert set(t) == set(O.REWARD_TERMS)
    z, ori = s.obs[:, 0], s.root[:, 5] ** 2 + s.root[:, 6] ** 2
    up_r = np.float32(0.75) * (z > 1.4) - np.float32(0.75) * (z < 2.1)
    orient_r = np.float32(0.75) * (ori > 0.98)
    for tag, ecs in (("big", 0.005), ("low", 1.0)):
        want = g["n%d_%s_reward" % (n, tag)]
        alive = want != -2.0
        assert alive.sum() >= 5
        pred = 0.5 + up_r * orient_r - np.float32(ecs) * t["electricity_cost"] - t["dof_at_limit_cost"] + t["leg_reward"]
        np.testing.assert_allclose(pred[alive], want[alive], rtol=3e-6, atol=3e-6)
    assert np.array_equal(t["up_reward"], np.where(z > 1.4, np.float32(0.75), np.float32(0)))
    assert np.array_equal(t["orient_reward"], np.where(ori > 0.92, np.float32(0.75), np.float32(0)))
    band = (ori > 0.92) & (ori <= 0.98)
    assert band.sum() >= 1 and np.all(t["orient_reward"][band] == 0.75) and np.all(orient_r[band] == 0)
    assert np.array_equal(t["alive_reward"], np.full(n, 0.5, np.float32))
    np.testing.assert_allclose(t["progress_reward"], s.pot - s.prev_pot, rtol=0, atol=0)
    np.testing.assert_allclose(t["actions_cost"], (s.targets.astype(np.float64) ** 2).sum(1), rtol=2e-6)
    hp = s.obs[:, 11]
    np.testing.assert_allclose(t["heading_reward"], np.where(hp > 0.8, 0.5, 0.5 * hp / 0.8), rtol=2e-6, atol=1e-7)
