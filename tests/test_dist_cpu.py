"""CPU, world_size 2 over gloo: the data-parallel path (fly_bproject_amd/dist.py) — one flat
gradient all-reduce per optimizer step, before the clip — must reproduce single-process training
on the concatenated batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bare_agent(net, var):
    from fly_bproject_amd.ppo import PPO
    p = PPO.__new__(PPO)
    p.net, p.action_var, p.clip = net, var, 0.2
    return p


def _batch(seed, n):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(n, 73, generator=g), torch.rand(n, 18, generator=g) * 2 - 1,
            torch.randn(n, generator=g) - 20, torch.randn(n, 1, generator=g), torch.randn(n, 1, generator=g))


def _train(net, batches, flat=None):
    import torch.nn as nn
    agent = _bare_agent(net, torch.full((18,), 0.15))
    optim = torch.optim.Adam(net.parameters(), lr=1e-3)
    for b in batches:
        loss = agent.minibatch_loss(*b)
        optim.zero_grad(set_to_none=False)
        loss.backward()
        if flat is not None:
            flat.allreduce_mean()
        nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        optim.step()
    return [p.detach().clone() for p in net.parameters()]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from fly_bproject_amd.dist import FlatGradAllReduce, broadcast_parameters
    from fly_bproject_amd.ppo import Net
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                 # different init per rank: broadcast must fix it
    net = Net(73, 18)
    broadcast_parameters(net, src=0)
    flat = FlatGradAllReduce(net.parameters(), world)
    assert flat.flat.numel() == 69587             # one 278 KB buffer
    batches = []
    for step in range(4):
        full = _batch(step, 256)
        batches.append(tuple(t[rank * 128:(rank + 1) * 128] for t in full))
    params = _train(net, batches, flat)
    torch.save(params, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    for a, b in zip(r0, r1):
        assert torch.equal(a, b)                  # replicas stay bit-identical
    from fly_bproject_amd.ppo import Net
    torch.manual_seed(100)
    net = Net(73, 18)
    ref = _train(net, [_batch(step, 256) for step in range(4)])
    for a, b in zip(r0, ref):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=2e-4, atol=2e-5)


def test_flat_grad_aliasing_guard():
    from fly_bproject_amd.dist import FlatGradAllReduce
    from fly_bproject_amd.ppo import Net
    net = Net(73, 18)
    flat = FlatGradAllReduce(net.parameters(), 1)
    net.pi(torch.randn(4, 73)).sum().backward()
    assert flat.flat.abs().sum() > 0              # autograd accumulated straight into the flat buffer
    flat.allreduce_mean()
    torch.optim.SGD(net.parameters(), lr=0.1).zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError):
        flat.allreduce_mean()


def _run_bench(extra, env_extra=None, timeout=300):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + extra, env=env, capture_output=True,
                       text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`bench.py --gpus 2` with no external launcher must start two rank processes itself (before any
    GPU call), rendezvous them, take the max over ranks and print ONE line with n_gpus == 2."""
    r, lines = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "0", "--dry_run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["config"]["parallelism"] == "dp2"
    assert lines[0]["ms_per_step"] >= 19.0          # rank 1 sleeps 20 ms per step: the MAX over ranks is reported
    cfg = lines[0]["config"]                        # who took part: one identity per rank, gathered over the process group
    assert cfg["world_size_seen"] == 2 and cfg["backend"] == "gloo" and len(cfg["devices"]) == 2
    assert [d["rank"] for d in cfg["devices"]] == [0, 1] and cfg["devices"][0]["pid"] != cfg["devices"][1]["pid"]


def test_bench_dead_rank_fails_fast():
    """A rank that dies at start (here: exit code 3 before the rendezvous) must not leave its siblings waiting in a
    collective until some outer time limit: the parent polls all children, terminates the rest and returns non-zero
    within seconds, naming the rank."""
    import time
    t0 = time.time()
    r, lines = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--dry_run", "--fail_rank", "1"], timeout=120)
    took = time.time() - t0
    assert r.returncode == 3 and not lines, (r.returncode, r.stderr[-1000:])
    assert took < 30.0, took
    assert "rank 1 exited with code 3" in r.stderr


def test_bench_refuses_a_launcher_that_disagrees_with_gpus():
    r, lines = _run_bench(["--gpus", "2", "--dry_run"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and not lines


# ---- the PRODUCT's gradient layout through the same two-rank exchange -------------------------------------------------------
# PPO._update_hip does not all-reduce 69 587 autograd views: it all-reduces the PACKED gradient the kernels write (74 272 floats in
# the layout of csrc/mlp_layout.h: K padded to 80 for layer 1, the two heads stacked into one 32 x 128 last layer, structural
# zeros and padding masked), in this order: local reduce -> exchange (sum) -> x 1/world inside the optimizer launch -> the clip sees
# the GLOBAL gradient; element 76 (a masked padding column of W1) is the "this gradient is invalid" mark that must reach every rank.
def _packed_grad(pol, net):
    """The packed gradient mlp_grad_w / mlp_fused_grad would leave for `net`'s current .grad (masked positions carry junk there: 7.0)."""
    from fly_bproject_amd.policy import PACKED
    G = torch.full((PACKED,), 7.0)
    for name, view in pol.views.items():
        idx = torch.arange(PACKED).as_strided(view.shape, view.stride(), view.storage_offset())
        G[idx.reshape(-1)] = dict(net.named_parameters())[name].grad.reshape(-1)
    return G


def _packed_worker(rank, world, port, out_dir, mark_rank):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from fly_bproject_amd.policy import ERR_SLOT, PackedPolicy
    from fly_bproject_amd.ppo import Net
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.manual_seed(7)
    net = Net(73, 18)
    pol = PackedPolicy(net, "cpu")                   # layout tables only: views, grad_mask (no kernel is launched on the CPU)
    assert float(pol.grad_mask[ERR_SLOT]) == 0.0     # the mark lives in a masked element
    full = _batch(3, 256)
    mine = tuple(t[rank * 128:(rank + 1) * 128] for t in full)
    agent = _bare_agent(net, torch.full((18,), 0.15))
    agent.minibatch_loss(*mine).backward()           # local minibatch, inv_batch = 1 / local rows (as minibatch_grad)
    G = _packed_grad(pol, net)
    G[ERR_SLOT] = 1.0 if rank == mark_rank else 0.0  # what mlp_grad_reduce writes: 1 on the rank whose launch was refused
    dist.all_reduce(G, op=dist.ReduceOp.SUM)         # the exchange (RCCL on the GPUs), 297 KB
    scale = 1.0 / world                              # mlp_adam_step(grad_scale = 1 / world, self_norm): the clip sees the global gradient
    g = G * scale * pol.grad_mask
    norm = float(torch.sqrt((g.double() ** 2).sum()))
    torch.save({"G": G, "norm": norm, "mark": float(G[ERR_SLOT])}, os.path.join(out_dir, "packed%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mark_rank", [-1, 1])
def test_two_rank_gloo_packed_gradient_matches_single_process(tmp_path, mark_rank):
    """World size 2 over gloo on the packed 74 272-float gradient: both ranks end with the same buffer; its masked part times 1 / world
    equals the single-process gradient of the concatenated batch (and so does the clip norm); junk in masked positions never
    reaches the norm; an invalid-gradient mark set on ONE rank arrives non-zero on BOTH (mlp_adam_step then refuses the step on
    every rank alike), an unmarked exchange arrives as exactly 0."""
    sys.path.insert(0, REPO)
    from fly_bproject_amd.policy import ERR_SLOT, PACKED, PackedPolicy
    from fly_bproject_amd.ppo import Net
    port = _free_port()
    mp.spawn(_packed_worker, args=(2, port, str(tmp_path), mark_rank), nprocs=2, join=True)
    r = [torch.load(tmp_path / ("packed%d.pt" % k), weights_only=True) for k in range(2)]
    assert torch.equal(r[0]["G"], r[1]["G"]) and r[0]["norm"] == r[1]["norm"]
    assert r[0]["mark"] == r[1]["mark"] == (1.0 if mark_rank >= 0 else 0.0)
    torch.manual_seed(7)
    net = Net(73, 18)
    pol = PackedPolicy(net, "cpu")
    agent = _bare_agent(net, torch.full((18,), 0.15))
    agent.minibatch_loss(*_batch(3, 256)).backward()
    want = _packed_grad(pol, net) * pol.grad_mask
    got = r[0]["G"] * 0.5 * pol.grad_mask
    assert int(pol.grad_mask.sum()) == 69587 and got.numel() == PACKED == 74272
    # the Huber term enters as a scalar MEAN per minibatch (ppo.py:194, Q7): the mean of two half-batch means is the full-batch mean, so
    # the two-rank gradient is the single-process one up to fp32 summation order
    torch.testing.assert_close(got, want, rtol=1e-4, atol=2e-5 * float(want.abs().max()))
    np.testing.assert_allclose(r[0]["norm"], float(torch.sqrt((want.double() ** 2).sum())), rtol=1e-5)
    assert float(got[ERR_SLOT]) == 0.0               # masked: the mark itself never moves a weight or enters the norm
