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


@pytest.mark.gpu
def test_dqn_update_matches_reference_step(golden):
    """One DQN.update from the reference's weights on the reference's batch: same loss, same
    Q-network and target network afterwards (Adam 3e-4 + soft update 0.995)."""
    from fly_bproject_amd.dqn import DQN, Net, soft_update
    import types
    g = golden("g8_dqn")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")   # noqa: E731
    d = DQN.__new__(DQN)
    d.discount, d.mini_batch_size, d.tau, d.act_space = 0.99, 4, 0.995, 18
    d.q, d.q_target = Net(73, 18).to("cuda:0"), Net(73, 18).to("cuda:0")
    d.q.load_state_dict({k[2:]: t(g[k]) for k in g.files if k.startswith("q_") and "." in k})
    soft_update(d.q, d.q_target, tau=0.0)
    d.optimizer = torch.optim.Adam(d.q.parameters(), lr=3e-4)
    from fly_bproject_amd import _lib
    d._lib = _lib.load()
    batch = (t(g["b_obs"]), t(g["b_act"]), t(g["b_rew"]), t(g["b_next"]), t(g["b_done"]))
    d.replay = types.SimpleNamespace(sample=lambda m: batch)
    loss = d.update()
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=2e-5)
    for k, v in d.q.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g["q1_" + k], rtol=2e-4, atol=3e-5, err_msg=k)
    for k, v in d.q_target.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g["qt1_" + k], rtol=2e-4, atol=1e-6, err_msg=k)


@pytest.mark.gpu
def test_dqn_runs_end_to_end():
    from fly_bproject_amd.dqn import DQN
    from tests.hip_helpers import make_args
    torch.manual_seed(0)
    agent = DQN(make_args(256, dqn_mini_batch_size=8, replay_bytes=64 << 20))
    assert agent.replay.capacity >= 9
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(30):
            agent.run()
    torch.cuda.synchronize()
    assert agent.replay.size() == min(30, agent.replay.capacity) and agent.last_loss is not None
    assert torch.isfinite(agent.last_loss) and all(torch.isfinite(p).all() for p in agent.q.parameters())
    a = agent.act(agent.env.obs_buf, 0.0)
    assert a.shape == (256,) and float(a.abs().max()) <= 1.0
    agent.exit()
