"""`DQN`: the reference's DQN variant (UselessFiles/dqn.py, UselessFiles/replay.py; BASELINE
configs[4]) on the MI355X-native Fly environment.

Same structure and hyper-parameters as the reference (`Net` 3-layer LeakyReLU Q-network, Adam
3e-4, discount 0.99, soft target update tau 0.995, eps = max(0.01, 0.8 - 0.01*step/20), a batch of
128 stored steps x num_envs rows per update).  Everything on its hot path is hand-written HIP behind the C
ABI (csrc/dqn_mfma.hip, fp32 MFMA):
  * `act`      ONE launch per env step: Q-network forward + per-env first-argmax + eps-greedy mix
               (`dqn_act`; dqn.py:89-100, a per-env Python loop upstream);
  * `update`   per sampled replay step ONE launch for q_target(next_obs).max, q(obs), TD target, Huber
               loss, its gradient and the backward chain (`dqn_td_step`), one for dW (`dqn_grad_w`,
               accumulating over the sampled steps), and ONE launch for Adam + the soft target update
               (`dqn_adam_soft_update`; dqn.py:64-85);
  * the replay is an HBM-resident ring of whole steps (`[capacity, N, .]` tensors) instead of a Python
    deque of tuples, and a sample is a list of ring SLOTS: the kernels read the rows where they lie,
    nothing is gathered, concatenated or shuffled (the shuffle of replay.py:22 cannot change a mean).
`Net` stays a torch module whose parameters are views into the packed buffers (checkpoints, tests).

The upstream file is stale against the current `Fly` and cannot run as written; the two repairs
are explicit:
  D1  `num_obs` is the environment's 73, not the stale default 84 (dqn.py:17);
  D2  `act()` yields ONE scalar per env (dqn.py:89-100) while `Fly.step` takes 18 joint targets:
      the scalar is broadcast to all 18 DoFs.
"""
import ctypes as C
import random

import torch
import torch.nn as nn

from . import _lib
from .fly import Fly
from .params import NUM_DOF

IN, IN_PAD, H, OUT, NACT = 73, 80, 256, 32, 18
OFF_W1 = 0
OFF_B1 = OFF_W1 + H * IN_PAD
OFF_W2 = OFF_B1 + H
OFF_B2 = OFF_W2 + H * H
OFF_W3 = OFF_B2 + H
OFF_B3 = OFF_W3 + OUT * H
PACKED = OFF_B3 + OUT                       # 94752 (csrc/dqn_layout.h)
OFF_F1, OFF_F2, OFF_F3 = 0, H * IN_PAD, H * IN_PAD + H * H
FRAG = OFF_F3 + OUT * H                     # 94208
OFF_T3, OFF_T2 = 0, H * OUT
FRAG_T = OFF_T2 + H * H                     # 73728


def build_index_maps():
    """int32 [PACKED] maps master index -> position in the forward / transposed fragment copies (-1: none)."""
    import numpy as np
    from .policy import _frag_index
    idx_f = np.full(PACKED, -1, np.int32)
    idx_t = np.full(PACKED, -1, np.int32)
    for off_w, N, K, off_f in ((OFF_W1, H, IN_PAD, OFF_F1), (OFF_W2, H, H, OFF_F2)):
        n, k = np.meshgrid(np.arange(N), np.arange(K), indexing="ij")
        idx_f[off_w + n * K + k] = off_f + _frag_index(n, k, K)
    n, k = np.meshgrid(np.arange(OUT), np.arange(H), indexing="ij")
    w, h, kq, q = k // 64, (k % 64) // 32, (k % 32) // 4, k % 4           # layer 3 forward: split-K over the four waves
    idx_f[OFF_W3 + n * H + k] = OFF_F3 + ((w * 8 + kq) * 64 + (h * 32 + n)) * 4 + q
    idx_t[OFF_W3 + n * H + k] = OFF_T3 + _frag_index(k, n, OUT)           # W3^T: 256 outputs, 32 reduced
    n, k = np.meshgrid(np.arange(H), np.arange(H), indexing="ij")
    idx_t[OFF_W2 + n * H + k] = OFF_T2 + _frag_index(k, n, H)             # W2^T
    for idx, size in ((idx_f, FRAG), (idx_t, FRAG_T)):
        used = idx[idx >= 0]
        assert len(np.unique(used)) == len(used) == size and used.max() == size - 1
    return idx_f, idx_t


QB_HALVES = 3 * (H * IN_PAD + H * H + OUT * H)      # 282624 (csrc/dqn_layout.h)
QH_HALVES = 2 * (H * IN_PAD + H * H + OUT * H)      # 188416: two fp16 terms per weight (csrc/dqn_fused_h2.inc)
QTH_HALVES = 2 * (H * OUT + H * H)                  # 147456
QTB_HALVES = 3 * (H * OUT + H * H)                  # 221184
OFF_QB = (0, 3 * H * IN_PAD, 3 * (H * IN_PAD + H * H))
OFF_QTB3, OFF_QTB2 = 0, 3 * H * OUT


def build_plane_maps():
    """int32 [PACKED] maps master index -> term-0 position in the bf16x3 plane buffers QB / QTB (-1: no copy); terms 1 and 2
    follow 512 and 1024 16-bit words later (csrc/dqn_layout.h, the operand order of csrc/mlp_layout.h)."""
    import numpy as np
    from .policy import _plane_index
    idx_fb = np.full(PACKED, -1, np.int32)
    idx_tb = np.full(PACKED, -1, np.int32)
    for li, (off_w, N, K) in enumerate(((OFF_W1, H, IN_PAD), (OFF_W2, H, H), (OFF_W3, OUT, H))):
        n, k = np.meshgrid(np.arange(N), np.arange(K), indexing="ij")
        src = off_w + n * K + k
        idx_fb[src] = OFF_QB[li] + _plane_index(n, k, K)
        if li == 1:
            idx_tb[src] = OFF_QTB2 + _plane_index(k, n, N)          # W2^T: 256 outputs, 256 reduced
        elif li == 2:
            idx_tb[src] = OFF_QTB3 + _plane_index(k, n, N)          # W3^T: 256 outputs, 32 reduced
    for idx, size in ((idx_fb, QB_HALVES), (idx_tb, QTB_HALVES)):
        used = idx[idx >= 0].astype(np.int64)
        allpos = np.concatenate([used, used + 512, used + 1024])
        assert len(np.unique(allpos)) == len(allpos) == size and allpos.max() == size - 1
    return idx_fb, idx_tb


class Net(nn.Module):
    """dqn.py:17-29."""

    def __init__(self, num_obs=73, num_act=18):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(num_obs, 256), nn.LeakyReLU(), nn.Linear(256, 256), nn.LeakyReLU(),
                                 nn.Linear(256, num_act))

    def forward(self, x):
        return self.net(x)


def soft_update(net, net_target, tau):
    """dqn.py:33-36."""
    with torch.no_grad():
        for param_target, param in zip(net_target.parameters(), net.parameters()):
            param_target.data.copy_(param_target.data * tau + param.data * (1.0 - tau))


class ReplayBuffer:
    """replay.py:10-31 as an HBM ring of steps.  `buffer_limit` counts STEPS as upstream, where it is
    1e6 -- impossible with whole vectorised steps (1e6 x 32768 envs x 600 B = 19.7 TB), so the default
    capacity is a STATED number of steps (512: 4 x the 128-step sample; 10.1 GB at 32768 envs, 3.5 % of
    the 288 GB), capped by `budget_bytes`."""

    DEFAULT_STEPS = 512

    def __init__(self, num_envs, num_obs, device, buffer_limit=None, budget_bytes=32 << 30, seed=0):
        per_step = num_envs * (2 * num_obs + 3) * 4
        want = self.DEFAULT_STEPS if buffer_limit is None else int(buffer_limit)
        self.capacity = max(2, min(want, budget_bytes // per_step))
        self.bytes = self.capacity * per_step
        self.num_envs = num_envs
        c, n = self.capacity, num_envs
        self.obs = torch.empty((c, n, num_obs), device=device)
        self.next_obs = torch.empty((c, n, num_obs), device=device)
        self.action = torch.empty((c, n), device=device)
        self.reward = torch.empty((c, n), device=device)
        self.done = torch.empty((c, n), device=device)
        self.head = 0
        self.count = 0
        self._rng = random.Random(seed)                     # replay.py:19 uses random.sample: a HOST draw, no device sync

    def slot(self):
        """The ring slot the next step goes to: `(obs, action, reward, next_obs, done)` row views the
        caller (or a kernel) writes in place; `commit()` then publishes it."""
        h = self.head
        return self.obs[h], self.action[h], self.reward[h], self.next_obs[h], self.done[h]

    def commit(self):
        self.head = (self.head + 1) % self.capacity
        self.count = min(self.count + 1, self.capacity)

    def push(self, obs, action, reward, next_obs, done):
        o, a, r, no, d = self.slot()
        o.copy_(obs); no.copy_(next_obs); a.copy_(action); r.copy_(reward); d.copy_(done)
        self.commit()

    def sample_slots(self, mini_batch_size):
        """`mini_batch_size` distinct stored steps (replay.py:19)."""
        return self._rng.sample(range(self.count), mini_batch_size)

    def sample(self, mini_batch_size):
        """A list of per-step chunks `(obs [N,73], act [N], reward [N], next_obs [N,73], done [N])`: views of the
        ring, nothing is copied.  (replay.py:22-28 concatenates and shuffles the rows; neither changes the
        batch mean the update computes.)"""
        return [(self.obs[s], self.action[s], self.reward[s], self.next_obs[s], self.done[s])
                for s in self.sample_slots(mini_batch_size)]

    def size(self):
        return self.count


class QNetPacked:
    """The packed Q-network pair (online + target) of csrc/dqn_layout.h with the fragment-ordered copies the
    kernels stream; `net` / `net_target` parameters are re-pointed at strided views of the packed buffers."""

    def __init__(self, net, net_target, device):
        import numpy as np
        self.device = torch.device(device)
        z = lambda k: torch.zeros(k, dtype=torch.float32, device=self.device)   # noqa: E731
        self.P, self.PF, self.PT = z(PACKED), z(FRAG), z(FRAG_T)
        self.P_tgt, self.PF_tgt = z(PACKED), z(FRAG)
        idx_f, idx_t = build_index_maps()
        self.idx_f = torch.from_numpy(idx_f).to(self.device)
        self.idx_t = torch.from_numpy(idx_t).to(self.device)
        self._src_f = torch.nonzero(self.idx_f >= 0).squeeze(-1); self._dst_f = self.idx_f[self._src_f].long()
        self._src_t = torch.nonzero(self.idx_t >= 0).squeeze(-1); self._dst_t = self.idx_t[self._src_t].long()
        mask = np.zeros(PACKED, np.float32)
        for buf, module in ((self.P, net), (self.P_tgt, net_target)):
            views = {"net.0.weight": buf[OFF_W1:OFF_B1].view(H, IN_PAD)[:, :IN], "net.0.bias": buf[OFF_B1:OFF_W2],
                     "net.2.weight": buf[OFF_W2:OFF_B2].view(H, H), "net.2.bias": buf[OFF_B2:OFF_W3],
                     "net.4.weight": buf[OFF_W3:OFF_B3].view(OUT, H)[:NACT], "net.4.bias": buf[OFF_B3:OFF_B3 + NACT]}
            params = dict(module.named_parameters())
            assert set(params) == set(views), "Net does not have the reference's parameter set"
            with torch.no_grad():
                for k, view in views.items():
                    view.copy_(params[k].data.to(self.device))
                    params[k].data = view
            if buf is self.P:
                for view in views.values():
                    idx = torch.arange(PACKED).as_strided(view.shape, view.stride(), view.storage_offset())
                    mask[idx.reshape(-1).numpy()] = 1.0
        self.grad_mask = torch.from_numpy(mask).to(self.device)
        assert int(mask.sum()) == IN * H + H + H * H + H + NACT * H + NACT
        self.G, self.exp_avg, self.exp_avg_sq = z(PACKED), z(PACKED), z(PACKED)
        self.step = torch.zeros(1, dtype=torch.int32, device=self.device)
        # three-term bf16 planes of the fused update (csrc/dqn_fused.inc): online forward / transposed, target forward
        zi = lambda k: torch.zeros(k, dtype=torch.int16, device=self.device)   # noqa: E731
        self.QB, self.QTB, self.QB_tgt = zi(QB_HALVES), zi(QTB_HALVES), zi(QB_HALVES)
        idx_fb, idx_tb = build_plane_maps()
        self.idx_fb = torch.from_numpy(idx_fb).to(self.device)
        self.idx_tb = torch.from_numpy(idx_tb).to(self.device)
        self._src_fb = torch.nonzero(self.idx_fb >= 0).squeeze(-1); self._dst_fb = self.idx_fb[self._src_fb].long()
        self._src_tb = torch.nonzero(self.idx_tb >= 0).squeeze(-1); self._dst_tb = self.idx_tb[self._src_tb].long()
        # the fp16x2 update (csrc/dqn_fused_h2.inc): two-term weight planes (rebuilt by every `dqn_fused_update_h2` call from the masters),
        # the scale table (s | 1 / s | last maxima: 48 floats) and the overflow word
        self.QH, self.QTH, self.QH_tgt = zi(QH_HALVES), zi(QTH_HALVES), zi(QH_HALVES)
        self.h2_scales = z(48)
        self.h2_scales[:32] = 1.0
        self.h2_overflow = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.refresh()

    def refresh(self):
        """Rebuild the fragment copies and term planes from the packed masters (after a load / out-of-band change)."""
        from .policy import split_bf16x3
        with torch.no_grad():
            self.PF[self._dst_f] = self.P[self._src_f]
            self.PT[self._dst_t] = self.P[self._src_t]
            self.PF_tgt[self._dst_f] = self.P_tgt[self._src_f]
            for dst_buf, master, src, dst in ((self.QB, self.P, self._src_fb, self._dst_fb), (self.QTB, self.P, self._src_tb, self._dst_tb),
                                              (self.QB_tgt, self.P_tgt, self._src_fb, self._dst_fb)):
                for term, plane in enumerate(split_bf16x3(master[src])):
                    dst_buf[dst + 512 * term] = plane

    def plane_args(self):
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        return (p(self.QB), p(self.QTB), p(self.QB_tgt), p(self.idx_fb), p(self.idx_tb))


class DQN:
    def __init__(self, args, env=None):
        self.args = args
        self.env = env if env is not None else Fly(args)                  # dqn.py:44
        dev = self.device = self.env.device
        n = int(args.num_envs)
        self.act_space = 18                                               # dqn.py:47
        self.discount = 0.99
        self.mini_batch_size = int(getattr(args, "dqn_mini_batch_size", 128))
        self.batch_size = n * self.mini_batch_size
        self.tau = 0.995
        self.num_eval_freq = 100
        self.lr = 3e-4
        self.run_step = 1
        self.score = 0
        cap = getattr(args, "replay_steps", None)
        if cap is None:
            cap = max(4 * self.mini_batch_size, 64)                       # stated default: 4 samples' worth of steps
        self.replay = ReplayBuffer(n, self.env.num_obs, dev, buffer_limit=cap,
                                   budget_bytes=int(getattr(args, "replay_bytes", 32 << 30)),
                                   seed=int(getattr(args, "seed", 0)))
        print("replay capacity: %d steps x %d envs = %.2f GB of HBM" % (self.replay.capacity, n, self.replay.bytes / 1e9))
        self.q = Net(self.env.num_obs, self.act_space).to(dev)            # D1
        self.q_target = Net(self.env.num_obs, self.act_space).to(dev)
        soft_update(self.q, self.q_target, tau=0.0)                       # dqn.py:60
        self.q_target.eval()
        self._lib = _lib.load()
        self.packed = QNetPacked(self.q, self.q_target, dev)
        self._alloc_workspace(n)
        # the update's gradient through the fused bf16x3 launches (csrc/dqn_fused.inc) whenever the row blocks are whole 32-row tiles
        # of one size; FLY_DQN_FUSED=0 / args.dqn_fused=False keeps the per-step fp32-MFMA launches (the A/B)
        import os
        want = getattr(args, "dqn_fused", None)
        self.fused_update = (os.environ.get("FLY_DQN_FUSED", "1") != "0") if want is None else bool(want)
        # the fused update's arithmetic: "f16x2" (two fp16 terms per operand, three MFMA products per k block, per-class power-of-two
        # scales: csrc/dqn_fused_h2.inc; an update whose values did not fit is formed again in bf16x3) or "bf16x3" (csrc/dqn_fused.inc)
        gemm = getattr(args, "dqn_gemm", None) or os.environ.get("FLY_DQN_GEMM", "f16x2")
        if gemm not in ("f16x2", "bf16x3"):
            raise ValueError("dqn_gemm / FLY_DQN_GEMM must be f16x2 or bf16x3, not %r" % (gemm,))
        self.update_gemm = gemm
        # dZ2's plane image stays home (flags bit 2 of dqn_fused_update_h2: the dW2 kernel rebuilds dZ2 from a 2.3 KB record per tile);
        # FLY_DQN_DW2_RECON=0 keeps the image (the first form, the A/B)
        self.dw2_recon = os.environ.get("FLY_DQN_DW2_RECON", "1") != "0"
        self.h2_calibrated = False          # the lagged scales have seen an update's maxima
        self.h2_freeze = False              # tests: leave the lagged scales alone (run-to-run comparisons)
        self.h2_overflows = 0               # updates whose fp16x2 gradient was refused and formed again in bf16x3
        self._updates_issued = 0            # optimizer launches issued (the device counter packed.step lags by the refused ones)
        self._h2_use_b3 = False             # inside _h2_poll's redo: form the gradient with the bf16x3 launches
        self._h2_guard = False
        self._gen = torch.Generator(device=dev)
        self._gen.manual_seed(int(getattr(args, "seed", 0)))
        self._coin = torch.empty(n, device=dev)
        self._rand = torch.empty(n, device=dev)
        self._score_acc = torch.zeros((), device=dev)
        self.last_loss = None

    def _alloc_workspace(self, rows):
        """Saved activations / gradients of ONE sampled step (the update walks the sampled steps one by one)."""
        dev = self.device
        r = (int(rows) + 31) // 32 * 32
        e = lambda c: torch.empty(r, c, device=dev)   # noqa: E731
        self._ws_rows = r
        self._h1, self._h2, self._dz1, self._dz2, self._dz3 = e(H), e(H), e(H), e(H), e(OUT)
        self._gw_ws = torch.empty(int(self._lib.dqn_grad_workspace_floats()), device=dev)
        # per-tile Huber sums of every sampled step of one update (summed ONCE at its end, not per step)
        self._loss_part = torch.zeros(max(1, int(getattr(self, "mini_batch_size", 1))), r // 32, device=dev)

    def _h2_poll(self, block=False):
        """The lazy half of the fp16x2 update's refusal path (training loop): look at overflow words copied to the host after earlier
        updates.  A set word means: that update and every later one were refused by the optimizer launch (the word is sticky), nothing
        moved.  Clear it, count the refused updates on the device step counter, form as many updates in bf16x3 on fresh samples of the
        replay ring (the refused ones' samples are as good as any: replay.py:19 draws uniformly), calibrate again at the next update."""
        pend = getattr(self, "_h2_pending", None)
        if not pend:
            return
        while pend and (block or pend[0][0].query()):
            ev, host = pend.pop(0)
            ev.synchronize()
            if int(host[0]) == 0:
                continue
            pend.clear()
            pk = self.packed
            pk.h2_overflow.zero_()
            refused = self._updates_issued - int(pk.step.item())
            self.h2_overflows += max(refused, 1)
            self.h2_calibrated = False
            self._h2_use_b3 = True
            try:
                for _ in range(refused):
                    self._updates_issued -= 1
                    self.update()
            finally:
                self._h2_use_b3 = False
            return

    def _h2_watch(self):
        """Copy the overflow word to the host behind this update's launches (pinned, asynchronous) for a later _h2_poll."""
        if not hasattr(self, "_h2_pending"):
            self._h2_pending = []
        host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        host.copy_(self.packed.h2_overflow, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._h2_pending.append((ev, host))

    def _update_fused(self, chunks, inv_B, lazy=False):
        """The gradient of one update through `dqn_fused_update` (csrc/dqn_fused.inc): ALL sampled steps in two persistent launches
        (chain per tile with dW1 / dW3 in registers; dW2 over the saved H1 / dZ2 plane images) + one slab reduction.  bf16x3."""
        pk, lib = self.packed, self._lib
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        S, n = len(chunks), int(chunks[0][0].shape[0])
        tiles = S * n // 32
        if getattr(self, "_fu_rows", 0) < S * n:
            self._fu_rows = S * n
            # (sized for the bf16x3 launches -- the larger images -- and the fp16x2 workspace -- the larger workspace: either can run)
            self._fu_images = torch.empty(int(lib.dqn_fused_image_halves(C.c_int64(S * n))), dtype=torch.int16, device=self.device)
            self._fu_ws = torch.empty(max(int(lib.dqn_fused_workspace_floats()), int(lib.dqn_fused_h2_workspace_floats())), device=self.device)
            self._fu_loss = torch.zeros(tiles, device=self.device)
        if getattr(self, "_fu_S", 0) < S:
            # the chunk tables have a capacity of their own: an update of MORE chunks over no more rows (8 x 4096 after 4 x 8192)
            # must not walk DqnChunk records past the end of a table sized for the earlier split
            if getattr(self, "_fu_tab", None):
                torch.cuda.synchronize(self.device)           # copies from the old pinned tables may still be in flight
            self._fu_S = S
            self._fu_tab = [(torch.empty((S, 5), dtype=torch.int64, pin_memory=True), torch.empty((S, 5), dtype=torch.int64, device=self.device),
                             [None]) for _ in range(4)]
            self._fu_i = 0
        dev, aligned = self._fused_table(chunks)
        if self.update_gemm == "f16x2" and not self._h2_use_b3:
            if not self.h2_calibrated:
                # two passes over this update's rows that only take the class maxima (the second with the first's scales)
                for _ in range(2):
                    self._fused_launch_h2(dev, S, n, aligned, inv_B, 2)
                pk.h2_overflow.zero_()
                self.h2_calibrated = True
            loss_part = self._fused_launch_h2(dev, S, n, aligned, inv_B, 1 if self.h2_freeze else 0)
            if lazy:
                # the training loop: no host synchronisation here (a blocking read per env step drains the launch queue: measured
                # ~1 ms of a 13 ms step).  The optimizer launch takes the overflow word as `grad_invalid` and refuses on the device;
                # the word travels to the host asynchronously and is looked at when a later update begins (_h2_poll).
                self._h2_guard = True
                return loss_part
            if int(pk.h2_overflow.item()) == 0:          # (explicit batches -- tests, tools: a blocking read, the answer at once)
                return loss_part
            # some value did not fit fp16 under the scales the previous update left: nothing has been applied yet -- the bf16x3
            # launches form the same gradient with no scales at all, and the next update calibrates again
            pk.h2_overflow.zero_()
            self.h2_overflows += 1
            self.h2_calibrated = False
        return self._fused_launch(dev, S, n, aligned, inv_B)

    def _fused_table(self, chunks):
        """The device table of chunk pointers `dqn_fused_update` walks (DqnChunk[S]), through a rotating pinned staging buffer."""
        S = len(chunks)
        host, dev, ev = self._fu_tab[self._fu_i % len(self._fu_tab)]
        self._fu_i += 1
        if ev[0] is not None:
            ev[0].synchronize()                               # the copy that last used this pinned buffer (4 updates ago) is long done
        aligned = 1
        hn = host.numpy()                                     # (a view of the pinned block: row writes without tensor indexing)
        for i, (obs, act, reward, next_obs, done_mask) in enumerate(chunks):
            po, pn = obs.data_ptr(), next_obs.data_ptr()
            hn[i] = (po, pn, act.data_ptr(), reward.data_ptr(), done_mask.data_ptr())
            if (po | pn) & 15:
                aligned = 0
        dev[:S].copy_(host[:S], non_blocking=True)
        ev[0] = torch.cuda.Event()
        ev[0].record()
        return dev, aligned

    def _fused_launch(self, dev, S, n, aligned, inv_B):
        pk, lib = self.packed, self._lib
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        loss_part = self._fu_loss[:S * n // 32]
        _lib.check(lib.dqn_fused_update(p(pk.P), p(pk.QB), p(pk.QTB), p(pk.P_tgt), p(pk.QB_tgt), p(dev), C.c_int(S), C.c_int64(n),
                                        C.c_float(self.discount), C.c_float(inv_B), p(self._fu_images), p(self._fu_ws), p(pk.G),
                                        p(loss_part), C.c_int(aligned), _lib.stream_ptr()), "dqn_fused_update")
        return loss_part

    def _fused_launch_h2(self, dev, S, n, aligned, inv_B, flags):
        pk, lib = self.packed, self._lib
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        loss_part = self._fu_loss[:S * n // 32]
        _lib.check(lib.dqn_fused_update_h2(p(pk.P), p(pk.QH), p(pk.QTH), p(pk.P_tgt), p(pk.QH_tgt), p(pk.idx_fb), p(pk.idx_tb),
                                           p(pk.h2_scales), p(pk.h2_overflow), p(dev), C.c_int(S), C.c_int64(n), C.c_float(self.discount),
                                           C.c_float(inv_B), p(self._fu_images), p(self._fu_ws), p(pk.G), p(loss_part), C.c_int(aligned),
                                           C.c_int(flags | (4 if getattr(self, "dw2_recon", False) else 0)), _lib.stream_ptr()), "dqn_fused_update_h2")
        return loss_part

    def update(self, chunks=None):
        """dqn.py:64-85.  `chunks` (tests) = a list of `(obs, act, reward, next_obs, done_mask)` row blocks that
        together form the batch; default: `mini_batch_size` sampled steps of the replay ring."""
        lazy = chunks is None                   # the training loop's own update: refusals are detected later (_h2_poll)
        if lazy:
            self._h2_poll()
            chunks = self.replay.sample(self.mini_batch_size)
        self._updates_issued += 1
        self._h2_guard = False
        pk, lib = self.packed, self._lib
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        st = _lib.stream_ptr()
        B = sum(int(c[0].shape[0]) for c in chunks)
        inv_B = 1.0 / float(B)
        n0 = int(chunks[0][0].shape[0])
        if self.fused_update and n0 % 32 == 0 and all(int(c[0].shape[0]) == n0 for c in chunks):
            for c in chunks:
                for t in c:
                    assert t.is_contiguous() and t.dtype == torch.float32
            loss_part = self._update_fused(chunks, inv_B, lazy)
            guard = p(pk.h2_overflow) if self._h2_guard else None
            _lib.check(lib.dqn_adam_soft_update(p(pk.P), p(pk.PF), p(pk.PT), p(pk.P_tgt), p(pk.PF_tgt), p(pk.idx_f), p(pk.idx_t),
                                                p(pk.G), p(pk.grad_mask), p(pk.exp_avg), p(pk.exp_avg_sq), p(pk.step),
                                                C.c_float(self.lr), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8),
                                                C.c_float(self.tau), *pk.plane_args(), guard, st), "dqn_adam_soft_update")
            if self._h2_guard:
                self._h2_watch()
            return loss_part.sum() * inv_B
        if len(chunks) > self._loss_part.shape[0] or max(int(c[0].shape[0]) for c in chunks) > self._ws_rows:
            self.mini_batch_size = max(self.mini_batch_size, len(chunks))
            self._alloc_workspace(max(int(c[0].shape[0]) for c in chunks))
        self._loss_part.zero_()
        for i, (obs, act, reward, next_obs, done_mask) in enumerate(chunks):
            n = int(obs.shape[0])
            for t in (obs, act, reward, next_obs, done_mask):
                assert t.is_contiguous() and t.dtype == torch.float32
            _lib.check(lib.dqn_td_step(p(pk.P), p(pk.PF), p(pk.PT), p(pk.P_tgt), p(pk.PF_tgt), p(obs), p(next_obs), p(act),
                                       p(reward), p(done_mask), C.c_int64(n), C.c_float(self.discount), C.c_float(inv_B),
                                       p(self._h1), p(self._h2), p(self._dz3), p(self._dz2), p(self._dz1), p(self._loss_part[i]),
                                       st), "dqn_td_step")
            _lib.check(lib.dqn_grad_w(p(obs), p(self._h1), p(self._h2), p(self._dz1), p(self._dz2), p(self._dz3), C.c_int64(n),
                                      p(self._gw_ws), p(pk.G), C.c_int((1 if i else 0) | (2 if i == len(chunks) - 1 else 0)), st),
                       "dqn_grad_w")
        _lib.check(lib.dqn_adam_soft_update(p(pk.P), p(pk.PF), p(pk.PT), p(pk.P_tgt), p(pk.PF_tgt), p(pk.idx_f), p(pk.idx_t),
                                            p(pk.G), p(pk.grad_mask), p(pk.exp_avg), p(pk.exp_avg_sq), p(pk.step),
                                            C.c_float(self.lr), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8),
                                            C.c_float(self.tau), *pk.plane_args(), None, st), "dqn_adam_soft_update")
        return self._loss_part.sum() * inv_B                              # F.smooth_l1_loss: mean over the batch

    def q_parameters(self):
        return list(self.q.parameters())

    def q_values(self, obs):
        """Q table f32 [n,18] of `obs` through the MFMA forward (dqn.py:28)."""
        x = obs.reshape(-1, IN)
        x = x if x.is_contiguous() else x.contiguous()
        out = torch.empty((x.shape[0], NACT), device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        _lib.check(self._lib.dqn_forward(p(self.packed.P), p(self.packed.PF), p(x), C.c_int64(x.shape[0]), p(out),
                                         _lib.stream_ptr()), "dqn_forward")
        return out

    def act(self, obs, epsilon=0.0):
        """dqn.py:89-100: one scalar in [-1,1] per env, in one launch."""
        n = obs.shape[0]
        self._coin.uniform_(generator=self._gen)                         # the two draws of dqn.py:90-92, in order
        self._rand.uniform_(generator=self._gen)
        x = obs if obs.is_contiguous() else obs.contiguous()
        out = torch.empty(n, device=obs.device)
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        _lib.check(self._lib.dqn_act(p(self.packed.P), p(self.packed.PF), p(x), C.c_int64(n), p(self._coin), p(self._rand),
                                     C.c_float(epsilon), p(out), None, _lib.stream_ptr()), "dqn_act")
        return out

    def run(self):
        """dqn.py:102-126."""
        epsilon = max(0.01, 0.8 - 0.01 * (self.run_step / 20))
        obs_row, act_row, rew_row, next_row, done_row = self.replay.slot()       # the step is recorded where it will live
        obs_row.copy_(self.env.obs_buf)                                          # dqn.py:106
        act_row.copy_(self.act(obs_row, epsilon))
        self.env.step(act_row.unsqueeze(-1).expand(-1, NUM_DOF).contiguous())    # D2
        next_row.copy_(self.env.obs_buf); rew_row.copy_(self.env.reward_buf)     # dqn.py:109
        torch.sub(1.0, self.env.reset_buf, out=done_row)                         # 1 - done (dqn.py:113), BEFORE the reset clears it
        self.env.reset_async()                                                   # dqn.py:110
        self.replay.commit()
        if self.replay.size() > self.mini_batch_size:
            loss = self.update()
            self.last_loss = loss
            self._score_acc += rew_row.mean() / self.num_eval_freq
            if self.run_step % self.num_eval_freq == 0:
                self.score = float(self._score_acc.item()); self._score_acc.zero_()
                print('Steps: {:04d} | Reward {:.04f} | TD Loss {:.04f} Epsilon {:.04f} Buffer {:03d}'
                      .format(self.run_step, self.score, float(loss.item()), epsilon, self.replay.size()))
                self.score = 0
        self.run_step += 1

    def exit(self):
        self._h2_poll(block=True)               # settle what the last updates left to look at
        self.env.exit()
