#!/usr/bin/env python3
"""bench.py — env-steps/s of the whole PPO loop (rollout + update) on N MI355X.

One "step" (--steps K) is ONE PPO iteration on every rank: T = 16*(40960//num_envs) env steps of
`num_envs` envs (policy forward, sample, fused env step, rollout store) followed by the update
(critic pass + TD/GAE + 5 epochs x 15 minibatches of 40 960 samples, Adam).  At the default
8192 envs/GPU that is 80 env steps = 655 360 env-steps per rank per step (ppo.py:118-122).
Ranks shard envs with no data-path collective except the per-optimizer-step gradient all-reduce
(scaling: weak).  Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FUSED_STEP_BYTES_PER_ENV = 800     # fused step K1-K5, algorithmic bytes per env-step


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="PPO iterations timed")
    ap.add_argument("--warmup", type=int, default=1, help="PPO iterations untimed")
    ap.add_argument("--num_envs", type=int, default=8192, help="envs per GPU")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_alt_gemm", action="store_true", help="skip the secondary bf16x3 measurement")
    ap.add_argument("--kernel_reps", type=int, default=200)
    ap.add_argument("--dp_allreduce", choices=["auto", "rccl", "p2p"], default="rccl",
                    help="per-optimizer-step gradient exchange of N > 1 ranks: rccl (default) = torch.distributed.all_reduce; "
                         "p2p = the one-shot peer-to-peer kernel; auto = p2p if its start-up self-test against RCCL passes on "
                         "this node, else RCCL.  (--p2p_variant times the p2p kernel beside an rccl run.)")
    ap.add_argument("--gemm", choices=["f16x2", "bf16x3", "f32"], default=os.environ.get("FLY_GEMM", "f16x2"),
                    help="arithmetic of the MLP GEMMs of `value`: f16x2 (default) = bf16x3 for the rollout's policy and the critic pass, "
                         "and the optimizer-step gradient (82 %% of an iteration) in the two-term fp16 split with three product terms "
                         "and per-class power-of-two scales (csrc/mlp_fused_h2.inc); bf16x3 = fp32 operands split exactly into three "
                         "bf16 terms, six product terms, for every GEMM; f32 = v_mfma_f32_32x32x2_f32.  All three are held to the "
                         "reference's golden vectors at the fp32 tolerances (tests/test_mlp_train_gpu.py, tests/test_ppo_gpu.py) and to "
                         "fp64 (tests/test_fused_h2_gpu.py); the others are measured too and reported as labelled secondary figures")
    ap.add_argument("--p2p_variant", action="store_true",
                    help="with --dp_allreduce rccl: also open, self-test and TIME the one-shot peer-to-peer exchange as a labelled "
                         "variant (`grad_exchange_variants_us.p2p`).  Off by default: its cross-GPU leg has never run on xGMI "
                         "hardware, and nothing optional should be able to take the scaling line down with it")
    ap.add_argument("--no_dqn", action="store_true", help="skip the short labelled DQN block (configs[4]) of the default line")
    ap.add_argument("--fail_rank", type=int, default=-1, help=argparse.SUPPRESS)     # test hook: that rank exits 3 at start
    ap.add_argument("--workload", choices=["ppo", "dqn"], default="ppo",
                    help="ppo (default, BASELINE's metric config) or dqn = BASELINE configs[4] (labelled line of its own)")
    ap.add_argument("--dqn_envs", type=int, default=32768)
    ap.add_argument("--dqn_mini_batch", type=int, default=128, help="sampled replay steps per update (dqn.py:49)")
    ap.add_argument("--dqn_gemm", choices=["f16x2", "bf16x3"], default=None,
                    help="arithmetic of the fused DQN update (default: FLY_DQN_GEMM or f16x2; bf16x3 = round 4's launches, the A/B)")
    ap.add_argument("--dry_run", action="store_true",
                    help="rank plumbing only (spawn, rendezvous, barrier, max-over-ranks timing, the JSON line) with no "
                         "GPU work: what the CPU/gloo test of --gpus N exercises")
    return ap.parse_args()


def spawn_ranks(a):
    """`bench.py --gpus N` without an external launcher: start N rank processes (one per GPU) BEFORE this
    process touches the GPU, hand each the torch.distributed.run environment (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*), and watch ALL of them: the first rank that exits non-zero gets its siblings
    terminated within seconds (they would otherwise sit in a collective until the outer time limit) and its exit
    code returned; the stderr of every rank is kept (rank 0's goes through, the others' tails are printed on
    failure).  Rank 0's stdout carries the one JSON line.  Children are fresh child processes -- never an exec of
    a GPU-initialised parent."""
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, logs = [], []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        log = None if r == 0 else tempfile.TemporaryFile(mode="w+")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, stderr=log))
    rc, first_bad = 0, None
    live = set(range(a.gpus))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and first_bad is None:
                first_bad, rc = r, (abs(code) or 1)
        if first_bad is not None and live:
            for r in live:
                procs[r].terminate()
            deadline = time.time() + 10
            for r in sorted(live):
                try:
                    procs[r].wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            live.clear()
        if live:
            time.sleep(0.2)
    if first_bad is not None:
        sys.stderr.write("bench.py: rank %d exited with code %d; the other ranks were terminated\n" % (first_bad, rc))
        for r, log in enumerate(logs):
            if log is not None:
                log.seek(0)
                tail = log.read()[-2000:]
                if tail.strip():
                    sys.stderr.write("---- stderr of rank %d (tail) ----\n%s\n" % (r, tail))
    return rc


def device_identity(rank, local_rank, cuda=True):
    """What the JSON line needs to show that N DISTINCT GPUs took part: per rank the device index, name, uuid and PCI
    address as the runtime reports them."""
    d = {"rank": rank, "local_rank": local_rank, "pid": os.getpid()}
    if cuda:
        pr = torch.cuda.get_device_properties(local_rank)
        d.update(device_index=local_rank, name=pr.name, uuid=str(getattr(pr, "uuid", "")),
                 pci="%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
                 gcn_arch=getattr(pr, "gcnArchName", ""), cus=pr.multi_processor_count)
    else:
        d.update(device_index=None, name="cpu (dry run)")
    return d


def gather_identities(me, world):
    if world == 1:
        return [me]
    out = [None] * world
    dist.all_gather_object(out, me)
    return out


def make_args(n, **kw):
    import types
    d = dict(sim_device="cuda:0", num_envs=n, headless=True, testing=False, save=False, load=False,
             record=False, save_freq=100, save_path=None, load_path=None, seed=0, rank=0, world_size=1,
             variant="bigGrav", reward="standing")
    d.update(kw)
    return types.SimpleNamespace(**d)


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def _time_launches(fn, reps):
    """Average duration of `fn` (one kernel launch) from HIP events on the launch stream: three batches of `reps`
    back-to-back launches after a warm-up, the MEDIAN batch average (the first batch of a process also pays the
    clock ramp: 114 vs 100 us on the same launch)."""
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e-3 / reps)
    return sorted(out)[1]


# algorithmic work per unit: bytes per env-step / FLOP per sample
MLP_FWD_FLOP = 2 * (73 * 256 + 256 * 128 + 128 * 64 * 2 + 64 * 18 + 64)        # 138 112
MLP_BWD_DX_FLOP = 2 * (64 * 18 + 64 + 128 * 128 + 128 * 256)                   # 100 736
MLP_GRAD_W_FLOP = MLP_FWD_FLOP
PEAK_HBM_GBS, PEAK_F32_MFMA_TFLOPS, PEAK_BF16_MFMA_TFLOPS = 8000.0, 157.3, 2500.0                             # MI355X_MICROARCH.md


PHYSICS_BYTES_PER_ENV = 464        # integrator alone, 4 * (2 * 49 state floats + 18 actions)


class RolloutAllHarness:
    """The loop's rollout launch (`ppo_rollout_all`, T env steps of every 32-env tile in ONE launch: rollout_all_fs_kernel at
    8192 envs) on rollout tensors of its own, for HIP-event timing, for the profiler passes (tools/prof_kernels.py) and -- through
    the stamped diagnostic instantiation -- for the policy / env split of a step."""

    def __init__(self, env, pol, T, var):
        import ctypes as C
        from fly_bproject_amd import _lib
        self.C, self._lib, self.lib = C, _lib, _lib.load()
        self.env, self.pol, self.T, self.var = env, pol, T, var
        n, dev = int(env.args.num_envs), "cuda:0"
        self.n = n
        self.obs_ring = torch.zeros(T + 1, n, 73, device=dev)
        self.eps_all = torch.randn(T, n, 18, device=dev)
        self.act_all = torch.empty(T, n, 18, device=dev)
        self.logp_all = torch.empty(T, n, device=dev)
        self.v_ring = torch.empty(T + 1, n, device=dev)
        self.reward_all = torch.empty(T, n, device=dev)
        self.reset_rows = torch.zeros(T, n, dtype=torch.long, device=dev)
        self.progress_rows = torch.zeros(T, n, dtype=torch.long, device=dev)
        self.rows_applied = torch.zeros(1, dtype=torch.int32, device=dev)
        self._first = True

    def _args(self):
        C = self.C
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        return (self.env._handle, C.byref(self.env._bufs), p(self.pol.P), p(self.pol.PF), p(self.obs_ring), p(self.eps_all),
                p(self.var), C.c_float(0.00001), C.c_float(0.01), p(self.act_all), p(self.logp_all), p(self.v_ring),
                p(self.reward_all), C.c_int(self.T))

    def _chain(self):
        # the next launch starts where this one ended: flags from the last row, the first observation = the last one written
        if self._first:
            self.env._bufs.reset = self.reset_rows[self.T - 1].data_ptr()
            self.env._bufs.progress = self.progress_rows[self.T - 1].data_ptr()
            self._first = False

    def launch(self):
        C = self.C
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        self._lib.check(self.lib.ppo_rollout_all(*self._args(), p(self.rows_applied), self.pol.infer_pb_ptr(), p(self.reset_rows),
                                                 p(self.progress_rows), self._lib.stream_ptr()), "ppo_rollout_all")
        self._chain()

    def phase_split(self):
        """Shader-clock stamps of every workgroup at the top of each step and between its policy and env halves (stamped
        instantiation of rollout_all_fs_kernel): returns (policy share, env share) of a step, averaged over workgroups and
        steps 1 .. T-1 (step 0 reads its rows from HBM).  None when the launch does not take the fused-style kernel."""
        C = self.C
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        wgs = (self.n + 31) // 32
        stamps = torch.zeros(wgs, self.T + 1, 8, dtype=torch.int64, device="cuda:0")
        fn = self.lib.flyhip_debug_rollout_all_stamped
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p] + [C.c_void_p] * 5 + [C.c_float, C.c_float] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 5
        rc = fn(self.env._handle, C.cast(C.byref(self.env._bufs), C.c_void_p), p(self.pol.P), p(self.pol.PF), p(self.obs_ring),
                p(self.eps_all), p(self.var), 0.00001, 0.01, p(self.act_all), p(self.logp_all), p(self.v_ring), p(self.reward_all),
                self.T, self.pol.infer_pb_ptr(), p(self.reset_rows), p(self.progress_rows), p(stamps), self._lib.stream_ptr())
        if rc != 0:
            return None
        self._chain()
        torch.cuda.synchronize()
        raw = stamps.cpu().numpy()
        mask = (1 << 48) - 1
        rt = ((raw[:, self.T, 1] & mask) - (raw[:, 0, 6] & mask)).astype("float64") * 10e-9        # 100 MHz real-time counter
        clock_ghz = float(((raw[:, self.T, 0] - raw[:, 0, 0]).astype("float64") / rt).mean() / 1e9)
        s = raw.astype("float64")
        top, mid = s[:, :, 0], s[:, :-1, 7]
        pol_c = mid[:, 1:] - top[:, 1:-1]
        env_c = top[:, 2:] - mid[:, 1:]
        tot = pol_c.mean() + env_c.mean()
        inner = s[:, 1:-1, :]                # steps 1 .. T-1: [x converted, L1, L2, L3, L4 done] relative to the step's top, then sampling
        marks = [inner[:, :, k] - inner[:, :, k - 1] if k > 1 else inner[:, :, 1] - inner[:, :, 0] for k in range(1, 6)]
        marks.append(inner[:, :, 7] - inner[:, :, 5])
        names = ("x_convert", "layer1", "layer2", "layer3", "layer4_splitk", "outputs_sampling")
        return {"policy_frac": float(pol_c.mean() / tot), "env_frac": float(env_c.mean() / tot),
                "policy_cycles": float(pol_c.mean()), "env_cycles": float(env_c.mean()), "clock_ghz": clock_ghz,
                "policy_sub_cycles": {nm: float(m.mean()) for nm, m in zip(names, marks)}}


def mlp_peak_for(gemm):
    """Roofline of the ALGORITHMIC FLOP of an MLP GEMM: the dense 16-bit MFMA peak over the product terms per multiply-add
    (bf16x3: six; f16x2: three -- fp16 runs at the bf16 rate), or the fp32 matrix peak."""
    if gemm == "f16x2":
        return round(PEAK_BF16_MFMA_TFLOPS / 3.0, 1)
    return PEAK_F32_MFMA_TFLOPS if gemm == "f32" else round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1)


def _round_key(path):
    """Sort key of a profiles/ file by its round tag (r4l < r5a < r10b: the number numerically, then the letter), then by name."""
    import re
    m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(path))
    return (int(m.group(1)), m.group(2), os.path.basename(path)) if m else (-1, "", os.path.basename(path))


def latest_profile(pattern):
    """The newest profiles/<round tag>_<pattern> by round tag (natural order), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", pattern)), key=_round_key)
    return files[-1] if files else None


def fused_step_clock(pol, batch, h2):
    """In-kernel clock of the fused optimizer-step kernel (its STAMP instantiation: s_memtime / s_memrealtime of every workgroup at
    kernel entry and after its slab is written), after the timed launches have warmed the chip: GHz, or None."""
    import ctypes as C
    from fly_bproject_amd import _lib
    try:
        lib = _lib.load()
        x, act, olp, adv, tgt, var = batch
        rows = x.shape[0]
        stamps = torch.zeros(256 * 64 * 32, dtype=torch.int64, device="cuda:0")
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        arr = (C.c_void_p * 8)(stamps.data_ptr(), None, None, None, None, None, None, None)
        if pol._fused_ws is None:
            return None
        # the chip holds its clock down under this load: stamp behind a long run of back-to-back product launches, with no idle
        # gap in between (microarchitecture guide, DVFS item 6)
        for _ in range(400):
            pol.minibatch_grad(x, act, olp, adv, tgt, var, 0.2, fuse_norm=False)
        for _ in range(3):
            if h2:
                _lib.check(lib.mlp_fused_grad_h2(p(pol.P), p(pol.PH), p(pol.PTH), p(pol.h2_scales), p(pol.h2_overflow), 1, p(x), rows, p(act),
                                                 p(olp), p(adv), p(tgt), p(var), C.c_float(1.0 / rows), C.c_float(0.2), p(pol._fused_ws),
                                                 p(pol.G), None, None, None, p(pol.loss_part), arr, _lib.stream_ptr()), "stamp")
            else:
                _lib.check(lib.mlp_fused_grad(p(pol.P), p(pol.PB), p(pol.PTB), p(x), rows, p(act), p(olp), p(adv), p(tgt), p(var),
                                              C.c_float(1.0 / rows), C.c_float(0.2), p(pol._fused_ws), p(pol.G), None, None, None,
                                              p(pol.loss_part), arr, _lib.stream_ptr()), "stamp")
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(256, 64, 32)
        whole = (s[:, 63, 1] - s[:, 62, 0]).astype("float64")
        real_us = ((s[:, 63, 2] & 0xffffffffffff) - (s[:, 62, 1] & 0xffffffffffff)).astype("float64") / 100.0
        ok = real_us > 0
        if not ok.any():
            return None
        # GHz, and what it is the quotient of: shader cycles and real microseconds between kernel entry and "slab written", mean over
        # workgroups, of the STAMPED instantiation (a few per cent slower than the product: the stamps)
        return {"ghz": round(float((whole[ok] / real_us[ok]).mean() / 1e3), 3), "cycles": round(float(whole[ok].mean())),
                "real_us": round(float(real_us[ok].mean()), 2)}
    except Exception:       # noqa: BLE001  (a diagnostic: never cost the line)
        return None


def kernel_rooflines(num_envs, T, reps, gemm="f32"):
    """Per-kernel roofline entries, each measured live with HIP events; the dominant kernel (largest
    share of one PPO iteration) is returned first."""
    import ctypes as C
    from fly_bproject_amd import _lib
    from fly_bproject_amd.fly import Fly
    from fly_bproject_amd.policy import PackedPolicy
    from fly_bproject_amd.ppo import Net
    lib = _lib.load()
    env = Fly(make_args(num_envs))
    a = torch.zeros(num_envs, 18, device="cuda:0").uniform_(-1, 1)
    t_step = _time_launches(lambda: env.step(a), reps)
    rows = (40960 // num_envs) * num_envs
    net = Net(73, 18).to("cuda:0")
    pol = PackedPolicy(net, "cuda:0")
    pol.init_training(rows)
    pol.gemm = gemm
    h2 = pol.h2_live()
    x = torch.randn(rows, 73, device="cuda:0")
    act = torch.rand(rows, 18, device="cuda:0") * 2 - 1
    olp = torch.randn(rows, device="cuda:0") - 20
    adv = torch.randn(rows, device="cuda:0"); tgt = torch.randn(rows, device="cuda:0")
    var = torch.full((18,), 0.2, device="cuda:0")
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    s, d = pol.saves, pol.dz
    if h2:
        pol.calibrate_h2(x, act, olp, adv, tgt, var, 0.2)
    pol.minibatch_grad(x, act, olp, adv, tgt, var, 0.2)
    flags = torch.zeros((rows + 31) // 32, dtype=torch.int32, device="cuda:0")
    err = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    epoch = [0]

    def fwd_bwd(coherent=1):
        epoch[0] += 1
        lib.mlp_forward_backward(p(pol.P), p(pol.PF), p(pol.PT), p(x), rows, p(s["out"]), p(s["h1"]), p(s["h2"]), p(s["h3"]),
                                 p(act), p(olp), p(adv), p(tgt), p(var), 1.0 / rows, 0.2, p(d["dz4"]), p(d["dz3"]),
                                 p(d["dz2"]), p(d["dz1"]), p(pol.loss_part), p(flags), epoch[0], p(err), pol.pb_ptr(), pol.ptb_ptr(),
                                 coherent, _lib.stream_ptr())
    fused = pol.fused_step and pol.gemm == "bf16x3"
    t_fused = None
    fused_clock = None
    if fused:       # the loop's optimizer-step gradient: ONE persistent launch (forward + loss + dX chain + dW) + the slab reduction
        t_fused = _time_launches(lambda: pol.minibatch_grad(x, act, olp, adv, tgt, var, 0.2, fuse_norm=True), reps)
        fused_clock = fused_step_clock(pol, (x, act, olp, adv, tgt, var), h2)
        if h2:
            assert int(pol.h2_overflow.item()) == 0, "the timed fp16x2 launches overflowed"
    t_fb = _time_launches(lambda: fwd_bwd(1 if pol.handoff == "sc1" else 0), reps)
    # A/B of the tile hand-off inside the launch: sc1 write-through + L1-bypassing loads vs plain accesses on one XCD
    t_fb_alt = _time_launches(lambda: fwd_bwd(0 if pol.handoff == "sc1" else 1), reps)
    assert int(err.item()) == 0, "mlp_forward_backward reported a lost tile flag"
    xs = torch.randn(num_envs, 73, device="cuda:0")
    eps = torch.randn(num_envs, 18, device="cuda:0")
    a_o = torch.empty(num_envs, 18, device="cuda:0"); lp_o = torch.empty(num_envs, device="cuda:0")
    v_o = torch.empty(num_envs, device="cuda:0")
    t_pol = _time_launches(lambda: lib.mlp_forward_sample(p(pol.P), p(pol.PF), p(xs), num_envs, p(eps), p(var), 0, 0.0, 0.0, p(a_o),
                                                          p(lp_o), None, p(v_o), pol.infer_pb_ptr(), None, _lib.stream_ptr()), reps)
    # the loop's env step: policy + sampling + env step of every 32-env tile in ONE launch
    t_roll = _time_launches(lambda: lib.ppo_rollout_step(env._handle, C.byref(env._bufs), p(pol.P), p(pol.PF), p(xs), p(eps),
                                                         p(var), 0, 0.0, 0.0, p(a_o), p(lp_o), p(v_o), pol.infer_pb_ptr(),
                                                         None, _lib.stream_ptr()), reps)
    # the loop's rollout: ONE launch for the T steps of every tile (rollout_all_fs_kernel at <= one tile per CU)
    harness = RolloutAllHarness(env, pol, T, var)
    t_all = _time_launches(harness.launch, max(3, reps // 20))
    split = None
    try:
        harness.phase_split()
        split = harness.phase_split()
    except Exception:       # noqa: BLE001  (the split is a diagnostic: never cost the line)
        split = None
    del harness
    env.exit()
    t_gw = _time_launches(lambda: lib.mlp_grad_w(p(x), p(s["h1"]), p(s["h2"]), p(s["h3"]), p(d["dz1"]), p(d["dz2"]),
                                                 p(d["dz3"]), p(d["dz4"]), rows, p(pol.workspace), p(pol.G), None, None, None,
                                                 None, 1 if pol.gemm == "bf16x3" else 0, _lib.stream_ptr()), reps)
    # the optimizer launch is ~5 us: timed back to back it would read the HOST's launch rate (~10 us per ctypes call),
    # so it is captured once into a hipGraph of 20 launches and the replay is timed
    g = torch.cuda.CUDAGraph()
    pol.adam_step(norm_ready=True)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20):
            pol.adam_step(norm_ready=True)
    t_adam = _time_launches(g.replay, max(5, reps // 20)) / 20
    # TD target + GAE over the whole rollout (ppo.py:157-171): 20 B per element (read r, v, v_next; write target, adv)
    Tn = T * num_envs
    rw = torch.randn(T, num_envs, device="cuda:0"); vv = torch.randn(T + 1, num_envs, device="cuda:0")
    dn = torch.ones(num_envs, device="cuda:0"); tg_o = torch.empty(T, num_envs, device="cuda:0"); ad_o = torch.empty_like(tg_o)
    t_gae = _time_launches(lambda: lib.ppo_td_gae(p(rw), p(vv[:T]), p(vv[1:]), p(dn), C.c_float(0.99), C.c_float(0.95), T,
                                                  num_envs, p(tg_o), p(ad_o), 0, _lib.stream_ptr()), reps)

    def hbm(name, dur, bytes_per_launch, per_iter):
        ach = bytes_per_launch / dur / 1e9
        return {"kernel": name, "bound": "hbm", "achieved": round(ach, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(ach / PEAK_HBM_GBS, 5), "traffic": None, "avg_launch_us": round(dur * 1e6, 3),
                "algorithmic_per_launch": bytes_per_launch, "launches_per_iteration": per_iter,
                "iteration_share_ms": round(dur * per_iter * 1e3, 3)}

    # bf16x3 mode: six bf16 MFMA terms per product, so the algorithmic-FLOP roofline of every MLP GEMM is the dense bf16 peak / 6
    mlp_peak = mlp_peak_for(pol.gemm)

    def mfma(name, dur, flop_per_launch, per_iter, peak=None):
        peak = mlp_peak if peak is None else peak
        ach = flop_per_launch / dur / 1e12
        return {"kernel": name, "bound": "mfma", "achieved": round(ach, 3), "peak": peak,
                "unit": "TFLOP/s", "frac": round(ach / peak, 5), "traffic": None,
                "avg_launch_us": round(dur * 1e6, 3), "algorithmic_per_launch": flop_per_launch,
                "launches_per_iteration": per_iter, "iteration_share_ms": round(dur * per_iter * 1e3, 3)}

    fs_kernel = pol.gemm_infer == "bf16x3" and num_envs % 32 == 0      # (persistent over tiles beyond one tile per CU)
    roll_name = "%s (one launch per rollout: T=%d x [policy + sample + env step], %d envs)" % (
        "rollout_all_fs_kernel" if fs_kernel else "rollout_all_kernel", T, num_envs)
    roll_all = mfma(roll_name, t_all, MLP_FWD_FLOP * num_envs * T, 1)
    roll_all["per_env_step_us"] = round(t_all / T * 1e6, 3)
    if split:
        # the north_star's physics kernel AS IT RUNS IN THE LOOP: the env half of a step (action scale, masked reset, 15 substeps,
        # obs / reward / done pack) by in-kernel stamps, priced against the HBM roofline on SURVEY.md §8(d)'s byte counts
        env_us = t_all / T * 1e6 * split["env_frac"]
        pol_us = t_all / T * 1e6 * split["policy_frac"]
        roll_all["phases"] = {
            "source": "s_memtime stamps of the diagnostic instantiation (all workgroups, steps 1..T-1), shares applied to the HIP-event time",
            "policy_us_per_step": round(pol_us, 3), "env_us_per_step": round(env_us, 3),
            "policy_frac": round(split["policy_frac"], 4), "env_frac": round(split["env_frac"], 4),
            "in_kernel_clock_ghz": round(split.get("clock_ghz", 0.0), 3),
            "policy_sub_cycles": {k: round(v) for k, v in split.get("policy_sub_cycles", {}).items()},
            "policy_mfma": {"achieved": round(MLP_FWD_FLOP * num_envs / pol_us / 1e6, 3), "peak": mlp_peak_for(pol.gemm),
                            "unit": "TFLOP/s", "frac": round(MLP_FWD_FLOP * num_envs / pol_us / 1e6 / mlp_peak_for(pol.gemm), 5)},
            "physics_in_loop": {
                "bound": "hbm", "bytes_per_env_step": FUSED_STEP_BYTES_PER_ENV, "achieved": round(FUSED_STEP_BYTES_PER_ENV * num_envs / env_us / 1e3, 2),
                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(FUSED_STEP_BYTES_PER_ENV * num_envs / env_us / 1e3 / PEAK_HBM_GBS, 5),
                "integrator_only_bytes_per_env_step": PHYSICS_BYTES_PER_ENV,
                "integrator_only_frac": round(PHYSICS_BYTES_PER_ENV * num_envs / env_us / 1e3 / PEAK_HBM_GBS, 5),
                "note": "fused step K1-K5 (800 B per env-step) over the env half's time; the state never leaves registers between "
                        "steps, so the bytes are the rows the loop writes/reads per step; the phase is VALU-issue-bound (valu_roofline)"}}
    ks = [
        roll_all,
        # labelled A/B: ONE launch per env step (persistent_rollout=False) = policy forward + sampling + fused env step
        mfma("rollout_step_kernel (A/B: one launch per env step; policy + sample + env step, %d envs)" % num_envs, t_roll,
             MLP_FWD_FLOP * num_envs, 0),
        # north_star's physics kernel on its own (not launched by the default loop any more: share 0)
        hbm("fly_kernel<63> (fly_step)", t_step, FUSED_STEP_BYTES_PER_ENV * num_envs, 0),
        # the update's forward + loss + dX chain of one 40 960-row minibatch is ONE launch
        # bf16x3 mode: six bf16 MFMA terms per product, so the algorithmic-FLOP roofline is the dense bf16 peak / 6
        mfma("mlp_fwd_bwd_kernel", t_fb, (MLP_FWD_FLOP + MLP_BWD_DX_FLOP) * rows, 0 if fused else 75),
        # the forward body alone at num_envs rows: once per iteration for v(last next_obs)
        mfma("mlp_forward_kernel (policy + sample, %d rows)" % num_envs, t_pol, MLP_FWD_FLOP * num_envs, 1),
        mfma("mlp_grad_w_kernel (+reduce)", t_gw, MLP_GRAD_W_FLOP * rows, 0 if fused else 75),
        hbm("mlp_adam_apply_kernel (clip + Adam + fragment refresh)", t_adam, 74272 * 4 * 10, 75),
        hbm("ppo_td_gae_kernel (T=%d x %d envs)" % (T, num_envs), t_gae, 20 * Tn, 1),
    ]
    fused_kernel = ("mlp_fused_step_h2_kernel", "mlp_grad_reduce_h2_kernel") if h2 else ("mlp_fused_step_kernel", "mlp_grad_reduce_kernel")
    if fused:
        # algorithmic FLOP of the whole minibatch gradient (forward + dX chain + dW); share 0 entries above are the three-launch
        # A/B path (FLY_FUSED_STEP=0), timed on the same data.  fp16x2: three fp16 MFMA terms per product -> dense 16-bit peak / 3
        ks.insert(0, mfma("%s (forward + loss + dX + dW, %d rows; + %s)" % (fused_kernel[0], rows, fused_kernel[1]), t_fused,
                          (MLP_FWD_FLOP + MLP_BWD_DX_FLOP + MLP_GRAD_W_FLOP) * rows, 75, peak=mlp_peak_for("f16x2" if h2 else pol.gemm)))
        ks[0]["arithmetic"] = "f16x2" if h2 else pol.gemm
        ks[0]["in_kernel_clock_ghz"] = fused_clock["ghz"] if fused_clock else None
        ks[0]["in_kernel_clock_source"] = fused_clock
    # the env kernel's REAL bound is vector-instruction issue, not HBM: instructions per wave (committed PMC pass,
    # profiles/*_valu.json) x 4 issue cycles at one wave per SIMD, against the measured launch
    try:
        vf = [latest_profile("*_valu.json")]
        if vf[0]:
            vj = json.load(open(vf[-1]))
            ipw = vj["fly_kernel<63>"]["valu_insts_per_wave"]
            clk = vj.get("clock_ghz", 2.1)
            floor_us = ipw * 4 / (clk * 1e3)
            for k in ks:
                if k["kernel"].startswith("fly_kernel<63>"):
                    k["valu_roofline"] = {"bound": "valu-issue", "valu_insts_per_wave": ipw, "issue_cycles_per_inst": 4,
                                          "waves_per_simd": 1, "clock_ghz": clk, "floor_us": round(floor_us, 2),
                                          "frac": round(floor_us / (t_step * 1e6), 4), "source": os.path.basename(vf[-1])}
                ph = k.get("phases")
                if ph:      # the same body as the env half of a rollout step: its instruction count against the half's time,
                    # at the clock the rollout kernel itself runs at (measured by its stamps)
                    clk_in = ph.get("in_kernel_clock_ghz") or clk
                    fl = ipw * 4 / (clk_in * 1e3)
                    ph["physics_in_loop"]["valu_roofline"] = {
                        "bound": "valu-issue", "valu_insts_per_wave": ipw, "issue_cycles_per_inst": 4, "waves_per_simd": 1,
                        "clock_ghz": clk_in, "floor_us": round(fl, 2), "frac": round(fl / ph["env_us_per_step"], 4),
                        "source": os.path.basename(vf[-1])}
    except Exception:
        pass
    # HBM bytes per launch from the committed PMC passes (profiles/*_traffic.json, produced by
    # tools/summarize_pmc.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of the same kernels)
    traffic, traffic_file = {}, None
    try:
        traffic_file = latest_profile("*_traffic.json")
        if traffic_file:
            traffic = json.load(open(traffic_file))
    except Exception:
        traffic = {}
    grids = {"fly_kernel<63> (fly_step)": ("fly_kernel<63>", ((num_envs + 31) // 32) * 256),
             "mlp_fwd_bwd_kernel": ("mlp_fwd_bwd_kernel", 2 * ((rows + 31) // 32) * 256),
             "rollout_step_kernel (A/B: one launch per env step; policy + sample + env step, %d envs)" % num_envs: ("rollout_step_kernel", ((num_envs + 31) // 32) * 256),
             roll_name: ("rollout_all_fs_kernel" if fs_kernel else "rollout_all_kernel", ((num_envs + 31) // 32) * 256),
             "mlp_forward_kernel (policy + sample, %d rows)" % num_envs: ("mlp_forward_kernel", ((num_envs + 31) // 32) * 256),
             "mlp_grad_w_kernel (+reduce)": ("mlp_grad_w_b3_kernel" if pol.gemm == "bf16x3" else "mlp_grad_w_kernel", 256 * 1024)}
    if fused:
        grids[ks[0]["kernel"]] = (fused_kernel[0], 256 * 256)
    for k in ks:
        key = grids.get(k["kernel"])
        if key:
            t = traffic.get("%s@%d" % key)
            if t:
                k["traffic"] = t["hbm_bytes_per_launch"]
                k["traffic_source"] = os.path.basename(traffic_file) + " (committed PMC passes of that build, not live)"
                if key[0] == fused_kernel[0]:       # the entry times the reduction with it: add its bytes
                    r = next((v for kk, v in sorted(traffic.items()) if kk.startswith(fused_kernel[1] + "@")), None)
                    if r:
                        k["traffic"] += r["hbm_bytes_per_launch"]
                        k["traffic_note"] = "fused launch + slab reduction"
    for k in ks:
        if k["kernel"] == "mlp_fwd_bwd_kernel":
            k["handoff"] = pol.handoff
            k["other_handoff_avg_launch_us"] = {("xcd" if pol.handoff == "sc1" else "sc1"): round(t_fb_alt * 1e6, 3)}
    ks.sort(key=lambda k: -k["iteration_share_ms"])
    return ks


def cpu_baseline(num_envs):
    """The oracle ("port") timed on the host cores on a bounded sample of the same workload:
    `S` env steps of num_envs envs (C oracle, OpenMP over envs, + torch-CPU policy forward and
    sampling) and `M` PPO minibatch steps of 40 960 samples + the critic pass over 2*40 960 rows,
    extrapolated to one full iteration (T env steps, 2*T*N critic rows, 75 minibatch steps)."""
    import numpy as np
    from oracle import oracle as O
    from oracle import ppo_oracle as PO
    cores = min(16, os.cpu_count() or 1)      # the host share of one GPU on the box
    torch.set_num_threads(cores)
    O.set_threads(cores)
    T = 16 * (40960 // num_envs)
    cfg = O.default_config(num_envs)
    s = O.EnvState(num_envs)
    net = PO.OracleNet()
    var = torch.full((18,), 0.2)
    rng = np.random.default_rng(0)
    S, M = 480, 120         # ~2 s of env steps + ~12 s of minibatch steps on the box's 16 host cores
    obs = torch.zeros(num_envs, 73)
    O.env_step(cfg, s, np.zeros((num_envs, 18), np.float32))      # warm caches / threads
    t0 = time.perf_counter()
    for _ in range(S):
        with torch.no_grad():
            mu = net.pi(obs).numpy()
        eps = rng.standard_normal((num_envs, 18)).astype(np.float32)
        act, _ = O.sample_logprob(mu, var.numpy(), eps)
        O.env_step(cfg, s, act)
        obs = torch.from_numpy(s.obs.copy())
    t_step = (time.perf_counter() - t0) / S
    mb = 40960
    x = torch.randn(mb, 73); a = torch.rand(mb, 18) * 2 - 1
    olp = torch.randn(mb); tg = torch.randn(mb, 1); adv = torch.randn(mb, 1)
    optim = torch.optim.Adam(net.parameters(), lr=1e-3)
    t0 = time.perf_counter()
    for _ in range(M):
        loss = PO.minibatch_loss(net, x, a, olp, tg, adv, var)
        optim.zero_grad(); loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        optim.step()
    t_mb = (time.perf_counter() - t0) / M
    t0 = time.perf_counter()
    with torch.no_grad():
        net.v(torch.randn(2 * mb, 73))
    t_v = (time.perf_counter() - t0) / (2 * mb)
    t_gae0 = time.perf_counter()
    O.td_gae(np.zeros((T, num_envs), np.float32), np.zeros((T, num_envs), np.float32),
             np.zeros((T, num_envs), np.float32), np.ones(num_envs, np.float32))
    t_gae = time.perf_counter() - t_gae0
    t_iter = T * t_step + 2 * T * num_envs * t_v + t_gae + 75 * t_mb
    return {"value": round(T * num_envs / t_iter, 1), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d env steps of %d envs (C oracle, OpenMP over envs, torch-CPU policy) + %d PPO "
                      "minibatch steps of 40960 samples + critic pass over %d rows, extrapolated to one "
                      "iteration (T=%d, 75 optimizer steps)" % (S, num_envs, M, 2 * mb, T),
            "rollout_env_steps_per_s": round(num_envs / t_step, 1)}


def episode_return_vs_oracle(num_envs, steps=400):
    """BASELINE's second metric, "mean episode return vs ref", as far as it can be had: the reference logs only the mean per-step
    reward (ppo.py:233, :257-260) and its sim cannot run here, so the comparable is the ORACLE (part of the cpu_baseline leg)
    on identical inputs -- the reference reset state (fly.py:446-480), the same initial weights (torch.manual_seed(0),
    trainer.py:24), the same recorded eps, action_var 0.2 decaying 1e-5 per step (ppo.py:152, :236-237), `steps` env steps, no
    update -- against the HIP env + MFMA policy.  Each side runs closed-loop on its OWN observations, so trajectories part
    chaotically at the 1e-6 level and the comparison is STATISTICAL (mean return / length over the episodes that finish)."""
    import numpy as np
    from oracle import oracle as O
    from oracle import ppo_oracle as PO
    from fly_bproject_amd.fly import Fly
    from fly_bproject_amd.policy import PackedPolicy
    from fly_bproject_amd.ppo import Net
    torch.manual_seed(0)
    net = Net(73, 18)
    onet = PO.OracleNet()
    onet.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
    net = net.to("cuda:0")
    pol = PackedPolicy(net, "cuda:0")
    env = Fly(make_args(num_envs))
    cfg = O.default_config(num_envs)
    s = O.EnvState(num_envs)
    rng = np.random.default_rng(0)
    var = np.float32(0.2)
    obs_o = torch.zeros(num_envs, 73)
    obs_h = torch.zeros(num_envs, 73, device="cuda:0")
    ep_ret = np.zeros(num_envs); ep_len = np.zeros(num_envs)
    done_ret, done_len, done_cnt = 0.0, 0.0, 0
    t0 = time.perf_counter()
    for _ in range(steps):
        eps = rng.standard_normal((num_envs, 18)).astype(np.float32)
        sd = np.float32(np.sqrt(var))
        with torch.no_grad():
            mu_o = onet.pi(obs_o).numpy()
            mu_h = pol.forward(obs_h, want_mu=True, want_v=False)[0]
        a_o = np.clip(mu_o + sd * eps, -1.0, 1.0).astype(np.float32)                 # ppo.py:218, :220
        a_h = torch.clamp(mu_h + float(sd) * torch.from_numpy(eps).to("cuda:0"), -1.0, 1.0)
        O.env_step(cfg, s, a_o)
        env.step(a_h)
        obs_o = torch.from_numpy(s.obs.copy())
        obs_h = env.obs_buf
        ep_ret += s.reward; ep_len += 1
        fin = s.reset != 0
        done_ret += float(ep_ret[fin].sum()); done_len += float(ep_len[fin].sum()); done_cnt += int(fin.sum())
        ep_ret[fin] = 0; ep_len[fin] = 0
        var = np.float32(max(np.float32(0.01), np.float32(var - np.float32(0.00001))))
    mr, ml, cnt = env.episode_stats()
    env.exit()
    o_ret = done_ret / max(done_cnt, 1)
    o_len = done_len / max(done_cnt, 1)
    return {"hip": round(mr, 4), "oracle": round(o_ret, 4), "rel_diff": round(abs(mr - o_ret) / (abs(o_ret) + 1e-9), 5),
            "episodes": {"hip": cnt, "oracle": done_cnt}, "mean_length": {"hip": round(ml, 2), "oracle": round(o_len, 2)},
            "env_steps": steps, "num_envs": num_envs, "seconds": round(time.perf_counter() - t0, 1),
            "note": "statistical: both sides start from the reference reset state with the same initial weights, the same recorded "
                    "eps and the variance schedule of ppo.py:152/:236-237, no update; each runs closed-loop on its own observations "
                    "(fp32 trajectories part chaotically), so means over finished episodes are compared, not trajectories.  The "
                    "reference itself logs only the mean per-step reward (ppo.py:233)"}


def dry_run(a):
    """The rank plumbing of main() with the GPU work replaced by a sleep: rendezvous (gloo), fence,
    K timed "steps", max over ranks, one JSON line on rank 0."""
    from fly_bproject_amd.dist import init_from_env
    os.environ.setdefault("FLY_DIST_BACKEND", "gloo")
    if a.fail_rank == int(os.environ.get("RANK", "0")):
        sys.exit(3)                                    # test hook: this rank dies before the rendezvous
    rank, local_rank, world = init_from_env("cpu")
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    devices = gather_identities(device_identity(rank, local_rank, cuda=False), world)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        time.sleep(0.01 * (1 + rank))                # ranks differ: the line must carry the MAX
    if world > 1:
        dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "value": 0.0, "unit": "env-steps/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(float(tt[0]) / a.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "dry_run": True,
                          "config": {"workload": "dry_run", "parallelism": "dp%d" % world, "devices": devices,
                                     "world_size_seen": dist.get_world_size() if world > 1 else 1,
                                     "backend": dist.get_backend() if world > 1 else None}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


DQN_FWD_FLOP = 2 * (73 * 256 + 256 * 256 + 256 * 18)            # 177 664 per sample
DQN_BWD_DX_FLOP = 2 * (18 * 256 + 256 * 256)                     # 140 288


def dqn_measure(n, mb, warmup, steps, kernel_reps, gemm=None):
    """BASELINE configs[4]: DQN, 32768 envs, HBM-resident replay ring (stated capacity), per-env eps-greedy and
    Huber-TD on the HIP kernels.  A "step" = one DQN.run(): act (one launch), env step, record into the ring,
    one update on `dqn_mini_batch` sampled steps x num_envs rows (dqn.py:102-126).  One rank (the reference DQN
    has no data-parallel form and BASELINE names none).  Returns the line as a dict."""
    import ctypes as C
    from fly_bproject_amd import _lib
    from fly_bproject_amd.dqn import DQN
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    cap = max(4 * mb, 64)
    with quiet():
        agent = DQN(make_args(n, dqn_mini_batch_size=mb, replay_steps=cap, dqn_gemm=gemm))
        for _ in range(mb + warmup):                # fill the ring to the sample size (no updates yet), then `warmup` steps with updates
            agent.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with quiet():
        for _ in range(steps):
            agent.run()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert int(agent.packed.step.item()) == warmup + steps, "an update was skipped"
    h2_refused = int(agent.h2_overflows)
    recon_form = bool(getattr(agent, "dw2_recon", False))
    finite = all(torch.isfinite(p).all().item() for p in agent.q.parameters())
    lib = _lib.load()
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    pk, rp = agent.packed, agent.replay
    st = _lib.stream_ptr()
    t_act = _time_launches(lambda: lib.dqn_act(p(pk.P), p(pk.PF), p(rp.obs[0]), n, p(agent._coin), p(agent._rand),
                                               C.c_float(0.1), p(agent._dz3), None, st), kernel_reps)

    def mfma(name, dur, flop, per_step, peak=PEAK_F32_MFMA_TFLOPS):
        ach = flop / dur / 1e12
        return {"kernel": name, "bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 5), "traffic": None, "avg_launch_us": round(dur * 1e6, 3),
                "algorithmic_per_launch": flop, "launches_per_step": per_step, "step_share_ms": round(dur * per_step * 1e3, 3)}
    fused = bool(agent.fused_update) and n % 32 == 0
    if fused:
        # the update as it runs: dqn_fused_update = dqn_chain_kernel + dqn_dw2_kernel + the slab reduction, once per update over all
        # `mb` sampled steps.  Each launch timed alone (flyhip_debug_set_dqn_fused_phases), on the update's own sampled chunks.
        lib.flyhip_debug_set_dqn_fused_phases.argtypes = [C.c_int]
        lib.flyhip_debug_set_dqn_fused_phases.restype = None
        chunks = rp.sample(mb)
        inv_B = 1.0 / float(n * mb)
        t = {}
        agent._update_fused(chunks, inv_B)                                   # sizes the buffers
        dev, aligned = agent._fused_table(chunks)
        h2 = agent.update_gemm == "f16x2"
        agent.h2_freeze = True                                               # (timing launches: leave the scales where the run left them)
        launch = (lambda: agent._fused_launch_h2(dev, mb, n, aligned, inv_B, 1)) if h2 else (lambda: agent._fused_launch(dev, mb, n, aligned, inv_B))
        t_extra = 0.0
        if h2:      # the two small kernels every fp16x2 update adds (weight planes + weight scales in front, next scales behind)
            lib.flyhip_debug_set_dqn_fused_phases(0)
            t_extra = _time_launches(launch, 20)
        for name, mask in (("chain", 1), ("dw2", 2), ("reduce", 4)):
            lib.flyhip_debug_set_dqn_fused_phases(mask)
            t[name] = _time_launches(launch, 3 if mask < 4 else 20) - t_extra
        lib.flyhip_debug_set_dqn_fused_phases(7)
        agent.packed.h2_overflow.zero_()
        peak = mlp_peak_for("f16x2" if h2 else "bf16x3")
        kn = ("dqn_chain_h2_kernel", "dqn_dw2r_h2_kernel" if getattr(agent, "dw2_recon", False) else "dqn_dw2_h2_kernel") if h2 \
            else ("dqn_chain_kernel", "dqn_dw2_kernel")
        rows = n * mb
        dw13 = 2 * (73 * 256 + 256 * 18)                                   # dW1 + dW3 per row
        ks = [mfma(kn[0] + " (per 32-row tile: target fwd, online fwd, Huber-TD, dX chain, dW1 / dW3 / db in registers; "
                   "%d rows = one update%s)" % (rows, "; with dqn_h2_scales_kernel behind it" if h2 else ""),
                   t["chain"], (2 * DQN_FWD_FLOP + DQN_BWD_DX_FLOP + dw13) * rows, 1, peak),
              mfma(kn[1] + (" (dW2 over the saved H1 plane image, dZ2 rebuilt from a 2.3 KB record per tile)" if kn[1].startswith("dqn_dw2r")
                            else " (dW2 over the saved H1 | dZ2 plane images)"), t["dw2"], 2 * 256 * 256 * rows, 1, peak),
              {"kernel": "dqn_grad_reduce_kernel (fixed-order sum of the per-CU slabs)", "avg_launch_us": round(t["reduce"] * 1e6, 3),
               "launches_per_step": 1, "step_share_ms": round(t["reduce"] * 1e3, 3)},
              mfma("dqn_act_kernel (forward + argmax + eps-greedy)", t_act, DQN_FWD_FLOP * n, 1)]
        dtype = "f16x2" if h2 else "bf16x3"
        if h2:
            ks.insert(3, {"kernel": "dqn_h2_planes_kernel (per update: two-term weight planes and weight scales of both networks)",
                          "avg_launch_us": round(t_extra * 1e6, 3),
                          "launches_per_step": 1, "step_share_ms": round(t_extra * 1e3, 3)})
        # HBM bytes from the committed PMC passes (profiles/*_traffic.json: per launch of PROF_DQN_MB sampled steps -> scaled to `mb`)
        try:
            files = [f for f in [latest_profile("*_traffic.json")] if f]
            tr = json.load(open(files[-1])) if files else {}
            for k, nm in ((ks[0], kn[0]), (ks[1], kn[1])):
                t = next((v for kk, v in sorted(tr.items()) if kk.startswith(nm + "@") and "hbm_bytes_per_sampled_step" in v), None)
                if t and n == 32768:
                    k["traffic"] = t["hbm_bytes_per_sampled_step"] * mb
                    k["traffic_per_sampled_step"] = t["hbm_bytes_per_sampled_step"]
                    k["traffic_source"] = os.path.basename(files[-1])
        except Exception:
            pass
    else:
        t_td = _time_launches(lambda: lib.dqn_td_step(p(pk.P), p(pk.PF), p(pk.PT), p(pk.P_tgt), p(pk.PF_tgt), p(rp.obs[0]),
                                                      p(rp.next_obs[0]), p(rp.action[0]), p(rp.reward[0]), p(rp.done[0]), n,
                                                      C.c_float(0.99), C.c_float(1.0 / (n * mb)), p(agent._h1), p(agent._h2),
                                                      p(agent._dz3), p(agent._dz2), p(agent._dz1), p(agent._loss_part[0]), st),
                              kernel_reps)
        t_gw = _time_launches(lambda: lib.dqn_grad_w(p(rp.obs[0]), p(agent._h1), p(agent._h2), p(agent._dz1), p(agent._dz2),
                                                     p(agent._dz3), n, p(agent._gw_ws), p(pk.G), 1, st), kernel_reps)  # a middle step of a batch: no reduction
        ks = [mfma("dqn_td_kernel (target fwd + online fwd + Huber-TD + dX chain, %d rows)" % n, t_td,
                   (2 * DQN_FWD_FLOP + DQN_BWD_DX_FLOP) * n, mb),
              mfma("dqn_grad_w_kernel (partials accumulate; one reduction per update)", t_gw, DQN_FWD_FLOP * n, mb),
              mfma("dqn_act_kernel (forward + argmax + eps-greedy)", t_act, DQN_FWD_FLOP * n, 1)]
        dtype = "f32"
    capacity, rbytes = rp.capacity, rp.bytes
    agent.exit()
    del agent
    torch.cuda.empty_cache()
    return {
        "metric": "env-steps/sec (DQN act + env step + update), %d envs" % n, "value": round(n * steps / elapsed, 1),
        "unit": "env-steps/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "update_path": (("dqn_fused_update_h2 (fp16x2 chain + dW2 over H1's two-plane image, dZ2 %s; %d of the run's updates refused and formed "
                         "again in bf16x3)" % ("rebuilt from a 2.3 KB record per tile" if recon_form else "from its image", h2_refused))
                        if dtype == "f16x2" else "dqn_fused_update (bf16x3 chain + dW2 over plane images)") if fused
                       else "dqn_td_step + dqn_grad_w per sampled step (fp32 MFMA)",
        "config": {"workload": "fly_dqn_%denvs_batch%dsteps" % (n, mb), "num_envs_per_gpu": n, "sampled_steps_per_update": mb,
                   "rows_per_update": n * mb, "replay_capacity_steps": capacity, "replay_bytes": rbytes,
                   "updates_per_env_step": 1, "parallelism": "dp1"},
        "params_finite": finite, "roofline": ks[0], "kernels": ks[1:]}


def dqn_bench(a):
    if a.gpus != 1:
        sys.exit("bench.py --workload dqn runs on one GPU")
    print(json.dumps(dqn_measure(a.dqn_envs, a.dqn_mini_batch, a.warmup, a.steps, a.kernel_reps, a.dqn_gemm)), flush=True)


GEMM_LABEL = {
    "f16x2": "the optimizer-step gradient (forward, loss, dX chain, dW of a minibatch: mlp_fused_grad_h2) with every fp32 operand as two "
             "fp16 terms of its value times a per-tensor-class power of two, three product terms on v_mfma_f32_32x32x16_f16, fp32 "
             "accumulate (error vs fp64 inside the fp32 MFMA chain's: tests/test_fused_h2_gpu.py, tools/bf16x3_gemm.hip; a value that "
             "does not fit fp16 -> the step is refused on the device and redone in bf16x3); the rollout's policy and the critic pass "
             "in bf16x3 (below)",
    "f32": "fp32 operands on v_mfma_f32_32x32x2_f32, fp32 accumulate",
    "bf16x3": "fp32 operands split exactly into three bf16 terms each, six product terms on v_mfma_f32_32x32x16_bf16, fp32 "
              "accumulate (error vs fp64 below the fp32 MFMA chain's: tools/bf16x3_gemm.hip) for EVERY MLP GEMM: rollout policy, "
              "critic pass, the update's forward, dX chain and dW",
}


def time_exchange(agent, world, reps=50, p2p_variant=False):
    """The per-optimizer-step gradient exchange ALONE (297 KB all-reduce of the packed gradient), timed with HIP events on
    the stream it is issued on; every rank runs it, the caller reports rank 0's figure.  Returns {variant: us per call}."""
    out = {}
    if world == 1:
        return out
    G = agent.policy.G
    keep = G.clone()

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) * 1e3 / reps, 2)
    out["rccl"] = timed(lambda: dist.all_reduce(G, op=dist.ReduceOp.SUM))
    G.copy_(keep)
    p2p, own = agent._p2p, False
    if p2p is None and p2p_variant:       # the labelled variant: open + self-test it now (collective; votes on every stage)
        try:
            want = agent.dp_allreduce
            agent.dp_allreduce = "auto"
            p2p = agent._open_p2p(G.numel())
            agent.dp_allreduce = want
            own = p2p is not None
        except Exception as e:      # noqa: BLE001
            agent.p2p_selftest = "failed: %r" % (e,)
            p2p = None
    if p2p is not None:
        out["p2p"] = timed(lambda: p2p.allreduce_(G))
        ok = p2p.check()
        flag = torch.tensor([1 if ok else 0], device=G.device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            out["p2p"] = None
        if own:
            p2p.close()
    G.copy_(keep)
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))                    # before anything touches the GPU
    if a.dry_run:
        return dry_run(a)
    if a.workload == "dqn":
        return dqn_bench(a)
    if a.fail_rank == int(os.environ.get("RANK", "0")):
        sys.exit(3)
    from fly_bproject_amd.dist import broadcast_policy, init_from_env
    from fly_bproject_amd.ppo import PPO

    rank, local_rank, world = init_from_env("cuda")
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (the launcher and the flag must agree)" % (a.gpus, world))
    if os.environ.get("FLY_SINGLE_GPU"):         # rehearsal of the N>1 path on a one-GPU box (gloo transport)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    devices = gather_identities(device_identity(rank, local_rank), world)
    torch.manual_seed(0)
    args = make_args(a.num_envs, sim_device=dev, rank=rank, world_size=world, dp_allreduce=a.dp_allreduce)
    with quiet():
        agent = PPO(args)
        agent.policy.gemm = a.gemm
    broadcast_policy(agent)
    with quiet():
        agent.prepare()                              # peer windows + self-test (if asked for) BEFORE any warm-up or timing
    T = agent.rollout_size

    def iteration():
        with quiet():
            for _ in range(T):
                agent.run()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        iteration()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        iteration()
    fence()
    elapsed = time.perf_counter() - t0
    steps_done = agent.optim_step
    # rollout-only rate (no update) for the breakdown
    agent.args.testing = True
    iteration()
    fence()
    ROLL_REPS = 5               # back to back: one fence's sync latency (~0.1 ms) would be 5 % of a single 1.8 ms rollout
    r0 = time.perf_counter()
    for _ in range(ROLL_REPS):
        iteration()
    fence()
    rollout_s = (time.perf_counter() - r0) / ROLL_REPS
    # labelled A/B: the same rollout with one launch per env step (`persistent_rollout=False`; the default is ONE launch per
    # rollout, ppo_rollout_all, with per-step reset / progress rows)
    persist_s = None
    was_persistent = agent.persistent_rollout
    try:
        agent.persistent_rollout = not was_persistent
        iteration()
        fence()
        p0 = time.perf_counter()
        for _ in range(ROLL_REPS):
            iteration()
        fence()
        persist_s = (time.perf_counter() - p0) / ROLL_REPS
    finally:
        agent.persistent_rollout = was_persistent
    agent.args.testing = False
    if world > 1:
        tt = torch.tensor([elapsed, rollout_s], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, rollout_s = float(tt[0]), float(tt[1])
    assert steps_done == 75 * (a.warmup + a.steps), steps_done
    exchange_us = time_exchange(agent, world, p2p_variant=a.p2p_variant)
    # Secondary, clearly labelled measurement: the same iteration with the OTHER arithmetic of the MLP GEMMs.
    alt = None
    main_gemm = "f16x2" if agent.policy.h2_live() else agent.policy.gemm
    h2_overflows = int(agent.policy.h2_overflows)
    others = {"f16x2": ["f32", "bf16x3"], "bf16x3": ["f32"], "f32": ["bf16x3"]}[main_gemm]
    alts = []
    if not a.no_alt_gemm:
        for other in others:
            try:        # the secondary figures must never cost the primary one
                agent.policy.gemm = other
                iteration()
                fence()
                a0 = time.perf_counter()
                for _ in range(2):
                    iteration()
                fence()
                alt_s = time.perf_counter() - a0
                if world > 1:
                    tt = torch.tensor([alt_s], device=dev, dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    alt_s = float(tt[0])
                alts.append({"gemm": other, "arithmetic": GEMM_LABEL[other],
                             "value": round(world * a.num_envs * T * 2 / alt_s, 1), "unit": "env-steps/s", "steps": 2,
                             "ms_per_step": round(alt_s / 2 * 1e3, 3), "update_path": agent.policy.update_path()})
            except Exception as e:      # noqa: BLE001
                alts.append({"gemm": other, "error": repr(e)[:200]})
            finally:
                agent.policy.gemm = main_gemm
        alt = alts[0] if alts else None
    finite = all(torch.isfinite(p).all().item() for p in agent.net.parameters())
    backend = dist.get_backend() if world > 1 else None
    agent_exchange = {"p2p": "one-shot peer-to-peer kernel over hipIpc windows (dp_allreduce_p2p)",
                      "rccl": "torch.distributed.all_reduce (%s)" % (os.environ.get("FLY_DIST_BACKEND") or "nccl = RCCL")}.get(
        agent.dp_allreduce, agent.dp_allreduce)
    ep_ret, ep_len, ep_cnt = agent.env.episode_stats()
    refused = int(getattr(agent.policy, "fused_launch_failures", 0))
    update_path = agent.policy.update_path()
    p2p_selftest = getattr(agent, "p2p_selftest", None)
    used_exchange = agent.dp_allreduce
    with quiet():
        agent.exit()                                 # (flushes the score lines still queued: they must not reach the JSON's stdout)
    del agent
    torch.cuda.empty_cache()

    if rank == 0:
        env_steps = world * a.num_envs * T * a.steps
        line = {
            "metric": "env-steps/sec (PPO rollout + update), %d envs per GPU" % a.num_envs,
            "value": round(env_steps / elapsed, 1), "unit": "env-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "f32 (bf16x3 operand split, fp32 accumulate)",
                      "f16x2": "f32 (optimizer step: fp16x2 operand split; rollout policy + critic: bf16x3 operand split; fp32 accumulate)"}[main_gemm],
            "data": "synthetic",
            "config": {"workload": "fly_ppo_iteration_%denvs_T%d" % (a.num_envs, T),
                       "num_envs_per_gpu": a.num_envs, "rollout_size": T, "optimizer_steps_per_iteration": 75,
                       "minibatch_samples": (40960 // a.num_envs) * a.num_envs, "variant": "bigGrav",
                       "parallelism": "dp%d" % world, "gemm": main_gemm, "arithmetic": GEMM_LABEL[main_gemm],
                       "update_path": update_path,
                       "devices": devices, "world_size_seen": dist.get_world_size() if world > 1 else 1, "backend": backend,
                       "grad_exchange": "none (1 rank)" if world == 1 else
                       ("%s, one call per optimizer step (75 per iteration), 297 KB" % agent_exchange)},
            "rollout_only_env_steps_per_s": round(world * a.num_envs * T / rollout_s, 1),
            "rollout_launches": "one per rollout (ppo_rollout_all)" if was_persistent else "one per env step (ppo_rollout_step)",
            ("rollout_only_one_launch_per_step_env_steps_per_s" if was_persistent else "rollout_only_one_launch_per_rollout_env_steps_per_s"):
                None if not persist_s else round(world * a.num_envs * T / persist_s, 1),
            "params_finite": finite,
            "refused_steps": refused,                  # fused forward+backward launches whose optimizer steps were refused and redone
            "h2_overflow_updates": h2_overflows,       # updates in which an fp16x2 step did not fit fp16 and the rest was redone in bf16x3
            "mean_episode_return": None if ep_cnt == 0 else round(ep_ret, 4),
            "mean_episode_length": None if ep_cnt == 0 else round(ep_len, 2), "episodes_finished": ep_cnt,
        }
        if world > 1:
            line["grad_exchange_us_per_step"] = exchange_us.get(used_exchange)   # the exchange `value` ran on
            line["grad_exchange_variants_us"] = exchange_us         # rank 0, HIP events, the exchange alone
            line["p2p_selftest"] = p2p_selftest
        if alt:
            line["alt_gemm"] = alt                   # strict fp32 (v_mfma_f32_32x32x2_f32) when `value` is f16x2 / bf16x3
        if len(alts) > 1:
            line["alt_gemm_2"] = alts[1]             # f16x2 runs: bf16x3 for every GEMM (round 4's default), same box
        ks = kernel_rooflines(a.num_envs, T, a.kernel_reps, main_gemm)
        line["roofline"] = ks[0]                     # the dominant kernel of one iteration
        line["kernels"] = ks[1:]
        if world == 1 and not a.no_dqn:
            try:        # BASELINE configs[4], short and labelled: 32768 envs, stated ring, 10 timed env steps (after 2) with one update each
                d = dqn_measure(a.dqn_envs, a.dqn_mini_batch, 2, 10, max(5, a.kernel_reps // 4), a.dqn_gemm)
                line["dqn"] = {k: d[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "update_path",
                                                 "config", "params_finite", "roofline")}
            except Exception as e:      # noqa: BLE001
                line["dqn"] = {"error": repr(e)[:200]}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.num_envs)
            try:        # the second half of BASELINE's metric, beside the oracle (same leg: the only place bench.py touches oracle/)
                line["episode_return_vs_oracle"] = episode_return_vs_oracle(a.num_envs)
            except Exception as e:      # noqa: BLE001
                line["episode_return_vs_oracle"] = {"error": repr(e)[:200]}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
