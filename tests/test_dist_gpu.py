"""GPU (-m gpu): the whole data-parallel PPO path with world_size 2 on ONE MI355X (both ranks on
cuda:0, gloo transport: RCCL refuses two ranks on one device).  Ranks own different envs and eps
streams; after one PPO iteration (75 all-reduced optimizer steps) the replicas are bit-identical."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _p2p_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from fly_bproject_amd.dist import P2PAllReduce
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    n = 74272
    ar = P2PAllReduce(n, "cuda:0")
    outs = []
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(100 + rank)
    for it in range(40):                                   # many epochs: both parities, back-to-back launches, no host sync
        g = torch.randn(n, device="cuda:0", generator=gen) * (1.0 + it)
        mine = g.clone()
        ar.allreduce_(g)
        if it % 3 == 0:
            torch.cuda._sleep(200000 * (rank + 1))         # uneven arrival
        outs.append((mine.cpu(), g.cpu()))
    assert ar.check()
    torch.save(outs, os.path.join(out_dir, "p2p%d.pt" % rank))
    dist.barrier()
    ar.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_p2p_allreduce_ranks_on_one_gpu(tmp_path, world):
    """dp_allreduce_p2p between `world` processes (all on cuda:0, windows exchanged through hipIpc): after every
    epoch every rank holds the sum of all inputs, added in rank order, bit for bit -- 40 epochs back to back
    with uneven arrival.  (Exercises the protocol and the IPC plumbing; the xGMI leg needs a multi-GPU node.)"""
    port = _free_port()
    mp.spawn(_p2p_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / ("p2p%d.pt" % r), weights_only=True) for r in range(world)]
    for it in range(len(outs[0])):
        want = torch.zeros_like(outs[0][it][0])
        for r in range(world):
            want = want + outs[r][it][0]                    # rank order, starting from 0
        for r in range(world):
            assert torch.equal(outs[r][it][1], want), (it, r)


def _worker(rank, world, port, out_dir, dp_mode="grad_allreduce"):
    sys.path.insert(0, REPO)
    import contextlib
    import io
    import torch.distributed as dist
    from fly_bproject_amd.dist import broadcast_policy
    from fly_bproject_amd.ppo import PPO
    from tests.hip_helpers import make_args
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.manual_seed(10 + rank)                      # different init per rank: the broadcast must fix it
    with contextlib.redirect_stdout(io.StringIO()):
        p2p = dp_mode == "p2p"
        agent = PPO(make_args(2048, rank=rank, world_size=world, seed=0, dp_mode="grad_allreduce" if p2p else dp_mode,
                              dp_allreduce="p2p" if p2p else "rccl"))
        broadcast_policy(agent)
        for _ in range(agent.rollout_size):
            agent.run()
    torch.cuda.synchronize()
    assert agent.optim_step == 75
    torch.save({"P": agent.policy.P.cpu(), "PF": agent.policy.PF.cpu(), "acts": agent.all_acts[0, :8].cpu(),
                "finite": bool(torch.isfinite(agent.policy.P).all())}, os.path.join(out_dir, "r%d.pt" % rank))
    agent.exit()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dp_mode", ["grad_allreduce", "param_average", "p2p"])
def test_two_ranks_one_gpu(tmp_path, dp_mode):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), dp_mode), nprocs=2, join=True)
    a = torch.load(tmp_path / "r0.pt", weights_only=True)
    b = torch.load(tmp_path / "r1.pt", weights_only=True)
    assert a["finite"] and b["finite"]
    assert torch.equal(a["P"], b["P"]) and torch.equal(a["PF"], b["PF"])     # replicas in lock step
    assert not torch.equal(a["acts"], b["acts"])                              # but different rollouts
    if dp_mode == "p2p":
        # the one-shot kernel sums in rank order like gloo's two-rank sum: the SAME run through the collective, made right
        # here (no dependence on another parametrisation or on test order), ends at the same parameters bit for bit
        ref_dir = tmp_path / "collective"
        ref_dir.mkdir()
        mp.spawn(_worker, args=(2, _free_port(), str(ref_dir), "grad_allreduce"), nprocs=2, join=True)
        ref = torch.load(ref_dir / "r0.pt", weights_only=True)
        assert torch.equal(a["P"], ref["P"]) and torch.equal(a["acts"], ref["acts"])


def test_bench_gpus_2_on_one_gpu():
    """bench.py's own rank spawning end to end on the one-GPU box: `--gpus 2` starts two ranks (both
    on cuda:0, gloo transport), each runs a full PPO iteration with the per-step gradient all-reduce,
    and rank 0 prints one line with n_gpus == 2 and twice the env-steps of one rank."""
    import json
    import subprocess
    env = dict(os.environ, FLY_SINGLE_GPU="1", FLY_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--num_envs", "2048", "--no_cpu_baseline", "--no_alt_gemm", "--no_dqn", "--p2p_variant", "--kernel_reps", "5"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["params_finite"]
    T = line["config"]["rollout_size"]
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 2 * 2048 * T) < 1e-3 * 2 * 2048 * T
    # the line proves who took part: one entry per rank with its device identity, the world size the process group reports,
    # the exchange that carried `value` (the collective) with its own HIP-event timing, and the labelled peer-to-peer variant
    cfg = line["config"]
    assert cfg["world_size_seen"] == 2 and len(cfg["devices"]) == 2 and {d["rank"] for d in cfg["devices"]} == {0, 1}
    assert all(d["name"] and d["pid"] for d in cfg["devices"]) and cfg["devices"][0]["pid"] != cfg["devices"][1]["pid"]
    assert cfg["grad_exchange"].startswith("torch.distributed.all_reduce")
    assert line["grad_exchange_us_per_step"] > 0 and "rccl" in line["grad_exchange_variants_us"]
    assert line["grad_exchange_variants_us"].get("p2p", 0) > 0 and line["p2p_selftest"] == "passed"      # --p2p_variant
    assert line["refused_steps"] == 0


def _late_peer_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    import contextlib
    import io
    import json
    import torch.distributed as dist
    from fly_bproject_amd import _lib
    from fly_bproject_amd.dist import broadcast_policy
    from fly_bproject_amd.ppo import PPO
    from tests.hip_helpers import make_args
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.manual_seed(0)
    res = {"error": None}
    with contextlib.redirect_stdout(io.StringIO()) as out:
        agent = PPO(make_args(2048, rank=rank, world_size=world, seed=0, dp_allreduce="p2p"))
        broadcast_policy(agent)
        agent.prepare()                                   # windows + self-test with the default poll budget
        assert agent._p2p is not None and agent.p2p_selftest == "passed"
        os.environ["FLY_P2P_POLL_LOG2"] = "4"             # from here on a rank gives up on a peer after 16 polls
        for _ in range(agent.rollout_size - 1):
            agent.run()
        torch.cuda.synchronize()
        dist.barrier()
        epochs_before = agent._p2p.epoch
        if rank == 1:
            torch.cuda._sleep(int(4e9))                   # the late peer: ~2 s before its update's first launch
        try:
            agent.run()                                   # the rollout's last step -> update -> 75 exchanges
        except _lib.FlyHipError as e:
            res["error"] = str(e)
        res["epochs"] = agent._p2p.epoch - epochs_before
        res["fuse_fwd_bwd"] = bool(agent.policy.fuse_fwd_bwd)
        res["stdout"] = out.getvalue()[-2000:]
    torch.cuda.synchronize()
    with open(os.path.join(out_dir, "late%d.json" % rank), "w") as f:
        json.dump(res, f)
    dist.barrier()
    agent._p2p.close()
    dist.destroy_process_group()


def test_p2p_late_peer_is_fatal_and_nothing_is_redone(tmp_path):
    """A peer that arrives after the bounded wait (FLY_P2P_POLL_LOG2=4): the rank that gave up raises the P2P error -- not the
    "device step counter is N steps behind" / "mlp_forward_backward ... refused" of the redo loop -- and issues NO further
    dp_allreduce_p2p launch (exactly the 75 of the update: a redo would pair with the peers' NEXT epochs and hand them wrong
    sums).  Its optimizer launches refused every un-reduced gradient on the device (fail closed)."""
    import json
    port = _free_port()
    mp.spawn(_late_peer_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = json.load(open(tmp_path / "late0.json"))
    assert r0["error"] is not None and "dp_allreduce_p2p" in r0["error"], r0
    assert r0["epochs"] == 75, r0
    assert "refused" not in r0["stdout"] and "steps behind" not in r0["error"]
