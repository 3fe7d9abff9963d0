"""GPU (-m gpu): every configuration BASELINE.json lists, at its full size, through the product path.

  configs[0]  16 envs, T = 40 960 (the reference's own CPU-runnable shape) through the PRODUCT path             (here)
  configs[1]  4096 envs, fp32                         -> tests/test_ppo_gpu.py (two iterations)
  configs[2]  16384 envs, bf16(x3) MLP on the bf16 MFMA pipe + hipGraph-captured rollout step   (here)
  configs[3]  8 x 8192 envs, gradient all-reduce      -> 8192-env shard here; ranks in tests/test_dist_gpu.py
  configs[4]  DQN, 32768 envs, HBM replay ring, eps-greedy + Huber-TD kernels                   (here)
  log.txt     the reference's one recorded run: 300 envs -> mini_chunk 136, rollout 2176, 75 steps (here)
"""
import contextlib
import io

import numpy as np
import pytest
import torch

from tests.hip_helpers import make_args

pytestmark = pytest.mark.gpu


def _quiet():
    return contextlib.redirect_stdout(io.StringIO())


def _iterations(agent, k):
    with _quiet():
        for _ in range(k * agent.rollout_size):
            agent.run()
    torch.cuda.synchronize()


def _snapshot(agent):
    return dict(obs=agent._obs_ring.clone(), acts=agent.all_acts.clone(), rew=agent.all_reward.clone(),
                logp=agent.all_log_prob.clone(), adv=agent.all_advantage.clone(), P=agent.policy.P.clone(),
                m=agent.policy.exp_avg.clone(), progress=agent.env.progress_buf.clone(), root=agent.env.root_tensor.clone(),
                var=agent.action_var.clone())


def test_config2_16384_envs_bf16x3_with_captured_rollout():
    """configs[2]: 16384 envs (T = 32), EVERY MLP GEMM on the bf16 matrix pipe through three-term splits (rollout
    policy, critic pass, the update's forward, dX chain and dW), the rollout step replayed from captured hipGraphs.
      * graph + bf16x3 == eager + bf16x3 bit for bit over two full PPO iterations (rollout tensors, advantages,
        parameters, Adam moments, env state): capturing changes no value;
      * the bf16x3 policy launch against the fp32-MFMA one on the same inputs (step 0: all-zero observations, Q8):
        actions and log-probs within the suite's fp32 tolerance;
      * one whole update (75 optimizer steps of 40 960 rows) on the SAME rollout in both arithmetics: the two
        end points differ by < 10 % of the distance the update moved the parameters (single-step gradients are
        held to 2e-4 in tests/test_mlp_train_gpu.py)."""
    from fly_bproject_amd.ppo import PPO
    runs = {}
    for tag, graph in (("eager_b3", False), ("graph_b3", True)):
        torch.manual_seed(0)
        with _quiet():
            agent = PPO(make_args(16384, graph=graph, persistent_rollout=False))
        agent.policy.gemm = "bf16x3"
        assert agent.policy.gemm_infer == "bf16x3"
        assert agent.mini_chunk_size == 2 and agent.rollout_size == 32              # ppo.py:120-122
        _iterations(agent, 1)
        assert agent.optim_step == 75
        first = _snapshot(agent)
        _iterations(agent, 1)
        assert agent.optim_step == 150
        if graph:
            assert len(agent._graphs) == 1                                           # iteration 2 ran from the captured rollout
        runs[tag] = (first, _snapshot(agent))
        for k, v in runs[tag][1].items():
            assert torch.isfinite(v.float()).all(), (tag, k)
        agent.exit()
    for it in (0, 1):
        for k in runs["eager_b3"][it]:
            assert torch.equal(runs["eager_b3"][it][k], runs["graph_b3"][it][k]), (it, k)
    # fp32 default vs bf16x3: the policy launch on identical inputs, then the update on an identical rollout
    torch.manual_seed(0)
    with _quiet():
        agent = PPO(make_args(16384, persistent_rollout=False))
        agent.policy.gemm = "f32"
        for _ in range(agent.rollout_size - 1):
            agent.run()
        agent._launch_step(agent.rollout_size - 1)                                    # the last env step without the update
        agent._flush_bookkeeping()
    f32_first = (agent.all_acts[0].clone(), agent.all_log_prob[0].clone())
    np.testing.assert_allclose(runs["eager_b3"][0]["acts"][0].cpu().numpy(), f32_first[0].cpu().numpy(), rtol=0, atol=3e-5)
    pol = agent.policy
    saved = (pol.P.clone(), pol.exp_avg.clone(), pol.exp_avg_sq.clone(), pol.step.clone(), pol.steps_issued)
    out = {}
    for gemm in ("f32", "bf16x3"):
        with torch.no_grad():
            pol.P.copy_(saved[0]); pol.exp_avg.copy_(saved[1]); pol.exp_avg_sq.copy_(saved[2]); pol.step.copy_(saved[3])
        pol.steps_issued = saved[4]
        pol.refresh()
        pol.gemm = gemm
        with _quiet():
            agent.update()
        torch.cuda.synchronize()
        out[gemm] = pol.P.clone()
    # 75 Adam steps amplify rounding-level gradient differences (an element whose gradient sits at rounding level
    # moves by up to +-lr per step in either arithmetic): the bar is the deviation relative to the distance the
    # update moved the parameters, and a cap on any single element (75 steps x lr 1e-3 = 0.075 is the reach).
    moved = (out["f32"] - saved[0]).norm().item()
    dev = (out["bf16x3"] - out["f32"]).norm().item()
    assert moved > 0.1 and dev <= 0.10 * moved, (dev, moved)
    assert float((out["bf16x3"] - out["f32"]).abs().max()) <= 8e-3
    assert not torch.equal(out["f32"], out["bf16x3"])                                 # a different arithmetic did run
    agent.exit()


def test_config0_shape_16_envs_one_iteration():
    """configs[0]'s shape on the product path: 16 envs -> mini_chunk_size 2560, T = 40 960 (ppo.py:118-122).  One whole PPO
    iteration: 40 960 one-launch env steps, make_data with the wave-scan GAE the host selects for few envs x long rollouts
    (equal to the reference's sequential loop, restated by the oracle, on this very rollout, to the scan's stated 2e-5), then
    75 optimizer steps of 40 960 rows."""
    from fly_bproject_amd.ppo import PPO
    from oracle import oracle as O
    torch.manual_seed(0)
    with _quiet():
        agent = PPO(make_args(16))
    assert agent.mini_chunk_size == 2560 and agent.rollout_size == 40960
    T = agent.rollout_size
    with _quiet():
        for _ in range(T - 1):
            agent.run()
        agent._launch_step(T - 1)                      # the rollout's last env step without the update that run() would start
        agent._flush_bookkeeping()
    torch.cuda.synchronize()
    assert agent._v_have == T
    np.testing.assert_allclose(float(agent.action_var[0]), max(0.01, 0.2 - T * 1e-5), rtol=1e-4)    # 0.2 -> 0.01 floor at step 19 000
    _, _, _, target, adv = agent.make_data()
    torch.cuda.synchronize()
    values, done_f = agent._keep
    t_o, a_o = O.td_gae(agent.all_reward[..., 0].cpu().numpy(), values[:T, :, 0].cpu().numpy(), values[1:, :, 0].cpu().numpy(),
                        done_f.view(-1).cpu().numpy())
    assert np.array_equal(target[..., 0].cpu().numpy(), t_o)                           # the TD target has no recurrence: bit-exact
    np.testing.assert_allclose(adv[..., 0].cpu().numpy(), a_o, rtol=2e-5, atol=2e-5)   # scan mode re-associates the chunk carries
    with _quiet():
        agent.update()
    torch.cuda.synchronize()
    assert agent.optim_step == 75 and int(agent.policy.step.item()) == 75
    for k, v in _snapshot(agent).items():
        assert torch.isfinite(v.float()).all(), k
    assert int(agent.policy.tile_wait_error.item()) == 0
    agent.exit()


def test_config3_shard_8192_envs_one_iteration():
    """configs[3]'s per-GPU shard: 8192 envs, T = 80, 75 optimizer steps of 40 960 samples."""
    from fly_bproject_amd.ppo import PPO
    torch.manual_seed(0)
    with _quiet():
        agent = PPO(make_args(8192))
    assert agent.mini_chunk_size == 5 and agent.rollout_size == 80
    _iterations(agent, 1)
    assert agent.optim_step == 75
    s = _snapshot(agent)
    for k, v in s.items():
        assert torch.isfinite(v.float()).all(), k
    assert int(agent.policy.tile_wait_error.item()) == 0
    agent.exit()


def test_log_txt_shape_300_envs():
    """The reference's only recorded run (log.txt:24-25, :48-49): 300 envs -> mini_chunk_size 136,
    rollout_size 2176, minibatches of 40 800 rows, 'Training' then Opt Step 0075, the variance printed
    at step 2200 = start - 2200e-5 (log.txt:49: 0.1000 -> 0.0780)."""
    from fly_bproject_amd.ppo import PPO
    torch.manual_seed(0)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        agent = PPO(make_args(300))
        agent.action_var = torch.full((18,), 0.1, device="cuda:0")                  # the run in log.txt started at 0.1
        for _ in range(2201):
            agent.run()
        agent.flush_log()                   # score lines are written when their values have arrived (PPO._emit_score)
    torch.cuda.synchronize()
    out = buf.getvalue().splitlines()
    assert "mini_chunk_size:  136" in out and "rollout_size:  2176" in out          # log.txt:24-25
    assert agent.mini_chunk_size * 300 == 40800
    assert agent.optim_step == 75
    i_train = out.index("Training")
    assert out[i_train - 1].startswith("Steps: 2100 | Opt Step: 0000")             # log.txt:47-48
    assert out[i_train + 1].startswith("Steps: 2200 | Opt Step: 0075") and out[i_train + 1].endswith("Action Var 0.0780")
    assert torch.isfinite(agent.policy.P).all() and torch.isfinite(agent._obs_ring).all()
    agent.exit()


def test_action_var_assignment_reaches_the_rollout_kernels():
    """Drop-in code assigns `agent.action_var = ...` (the reference does, ppo.py:237).  The rollout
    launches hold a pointer to the variance tensor: the assignment must be seen by the very next
    sampling / log-prob launch, with pending lazy decays applied to the OLD value first.  (Step-by-step launches:
    with the default one launch per rollout the steps in progress are already sampled, the setter says so in a warning.)"""
    from fly_bproject_amd.ppo import PPO, diag_gauss_logprob
    torch.manual_seed(0)
    with _quiet():
        agent = PPO(make_args(2048, testing=False))
        assert agent.persistent_rollout
        agent.run()
        with pytest.warns(UserWarning, match="next rollout"):
            agent.action_var = torch.full((18,), 0.1, device="cuda:0")
        agent.exit()
        agent = PPO(make_args(2048, testing=False, persistent_rollout=False))
        for _ in range(3):
            agent.run()
        assert float(agent.action_var[0]) == pytest.approx(0.2 - 3e-5, rel=1e-6)
        agent.action_var = torch.full((18,), 0.05, device="cuda:0")
        t = agent.mini_batch_number
        agent.run()
    torch.cuda.synchronize()
    assert float(agent.action_var[0]) == pytest.approx(0.05 - 1e-5, rel=1e-5)       # one more decay after the step
    # the step at row t sampled and scored with variance 0.05: recompute its log-prob from the stored pieces
    with torch.no_grad():
        mu = agent.net.pi(agent._obs_ring[t])
    var = torch.full((18,), 0.05, device="cuda:0")
    unclipped = mu + var.sqrt() * agent._eps_all[t]
    ref = diag_gauss_logprob(mu, unclipped, var)
    np.testing.assert_allclose(agent.all_log_prob[t].cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-4)
    assert torch.equal(agent.all_acts[t], unclipped.clamp(-1, 1)) or \
        torch.allclose(agent.all_acts[t], unclipped.clamp(-1, 1), atol=1e-6)
    agent.exit()


def test_config4_dqn_32768_envs_hbm_replay():
    """configs[4]: 32768 envs, replay ring in HBM with a STATED capacity (64 steps = 1.26 GB), per-env
    eps-greedy and Huber-TD on the HIP kernels, 24 env steps with an update each (8 sampled steps =
    262 144 rows per update): finite loss and parameters, the ring wraps nothing yet, and `act` on the
    live Q table is bit-equal to the oracle's first-argmax / eps mix."""
    from oracle import oracle as O
    from fly_bproject_amd.dqn import DQN
    torch.manual_seed(0)
    n = 32768
    with _quiet():
        agent = DQN(make_args(n, dqn_mini_batch_size=8, replay_steps=64))
    assert agent.replay.capacity == 64
    assert agent.replay.bytes == 64 * n * (2 * 73 + 3) * 4                           # 1.25 GB, SURVEY §8(d)
    with _quiet():
        for _ in range(24):
            agent.run()
    torch.cuda.synchronize()
    assert agent.replay.size() == 24 and agent.last_loss is not None
    assert torch.isfinite(agent.last_loss)
    for p in agent.q_parameters():
        assert torch.isfinite(p).all()
    obs = agent.env.obs_buf.clone()
    for eps in (0.0, 0.3):
        a = agent.act(obs, eps)
        torch.cuda.synchronize()
        q = agent.q_values(obs).cpu().numpy()
        ref = O.dqn_eps_greedy(q, agent._coin.cpu().numpy(), agent._rand.cpu().numpy(), eps)
        assert np.array_equal(a.cpu().numpy(), ref), eps
    agent.exit()
