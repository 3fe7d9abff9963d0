"""Packed parameter storage of `Net` and the MFMA forward (`mlp_forward`, include/flyhip.h).

`PackedPolicy` owns ONE flat fp32 buffer in the layout of csrc/mlp_layout.h and re-points the
`.data` of every `Net` parameter at a strided view into it, so torch (checkpoint I/O, autograd,
optimizers) and the HIP kernels see the same memory.  Padding and the structural zeros of the
stacked last layer never receive gradient (`grad_mask`).
"""
import ctypes as C
import os

import torch

from . import _lib

IN, IN_PAD, H1, H2, H3, OUT, NACT = 73, 80, 256, 128, 128, 32, 18
OFF_W1 = 0
OFF_B1 = OFF_W1 + H1 * IN_PAD
OFF_W2 = OFF_B1 + H1
OFF_B2 = OFF_W2 + H2 * H1
OFF_W3 = OFF_B2 + H2
OFF_B3 = OFF_W3 + H3 * H2
OFF_W4 = OFF_B3 + H3
OFF_B4 = OFF_W4 + OUT * H3
PACKED = OFF_B4 + OUT                      # 74272
ERR_SLOT = OFF_W1 + 76                     # a masked padding element of W1: the "this gradient is invalid" mark (csrc/mlp_grad_w.inc)
OFF_F1 = 0
OFF_F2 = OFF_F1 + H1 * IN_PAD
OFF_F3 = OFF_F2 + H2 * H1
OFF_F4 = OFF_F3 + H3 * H2
FRAG = OFF_F4 + OUT * H3                   # 73728
OFF_TF2 = 0
OFF_TF3 = OFF_TF2 + H1 * H2
OFF_TF4 = OFF_TF3 + H2 * H3
FRAG_T = OFF_TF4 + H3 * OUT                # 53248


def _frag_index(n, k, K):
    """Fragment-order offset of element (n, k) of an operand with K reduced (csrc/mlp_layout.h)."""
    h, kk = k // (K // 2), k % (K // 2)
    return (((n // 32) * (K // 8) + kk // 4) * 64 + (h * 32 + n % 32)) * 4 + kk % 4


def build_index_maps():
    """int32 [PACKED] maps master index -> position in PF / PTF (-1: no copy)."""
    import numpy as np
    idx_f = np.full(PACKED, -1, np.int32)
    idx_t = np.full(PACKED, -1, np.int32)
    layers = [(OFF_W1, H1, IN_PAD, OFF_F1, None), (OFF_W2, H2, H1, OFF_F2, OFF_TF2),
              (OFF_W3, H3, H2, OFF_F3, OFF_TF3), (OFF_W4, OUT, H3, OFF_F4, OFF_TF4)]
    for li, (off_w, N, K, off_f, off_t) in enumerate(layers):
        n, k = np.meshgrid(np.arange(N), np.arange(K), indexing="ij")
        src = off_w + n * K + k
        if li == 1:     # layer 2 forward: two K=128 operands (k < 128, k >= 128), one per half of H1 in LDS
            idx_f[src] = off_f + (k // (K // 2)) * (N * (K // 2)) + _frag_index(n, k % (K // 2), K // 2)
        elif li < 3:
            idx_f[src] = off_f + _frag_index(n, k, K)
        else:       # layer 4 forward: split-K over the four waves
            w, h, kq, q = k // 32, (k % 32) // 16, (k % 16) // 4, k % 4
            idx_f[src] = off_f + ((w * 4 + kq) * 64 + (h * 32 + n)) * 4 + q
        if off_t is not None:   # W^T: K outputs, N reduced
            idx_t[src] = off_t + _frag_index(k, n, N)
    for name, idx, size in (("PF", idx_f, FRAG), ("PTF", idx_t, FRAG_T)):
        used = idx[idx >= 0]
        assert len(np.unique(used)) == len(used) and used.max() < size, name
    assert (idx_f >= 0).sum() == FRAG and (idx_t >= 0).sum() == FRAG_T
    return idx_f, idx_t


def _plane_index(n, k, K):
    """Position (16-bit words) of term 0 of element (n, k) of a bf16x3 operand with K reduced
    (csrc/mlp_layout.h); terms 1 and 2 follow 512 and 1024 words later."""
    return (((n // 32) * (K // 16) + k // 16) * 3) * 512 + (((k % 16) // 8) * 32 + n % 32) * 8 + k % 8


PB_HALVES, PTB_HALVES = 221184, 159744
OFF_PB = (0, 61440, 159744, 208896)
OFF_PTB = (None, 0, 98304, 147456)


def build_plane_maps():
    """int32 [PACKED] maps master index -> term-0 position in PB / PTB (-1: no copy)."""
    import numpy as np
    idx_fb = np.full(PACKED, -1, np.int32)
    idx_tb = np.full(PACKED, -1, np.int32)
    layers = [(OFF_W1, H1, IN_PAD), (OFF_W2, H2, H1), (OFF_W3, H3, H2), (OFF_W4, OUT, H3)]
    for li, (off_w, N, K) in enumerate(layers):
        n, k = np.meshgrid(np.arange(N), np.arange(K), indexing="ij")
        src = off_w + n * K + k
        if li == 1:     # two K = 128 operands
            idx_fb[src] = OFF_PB[li] + (k // 128) * (3 * N * 128) + _plane_index(n, k % 128, 128)
        else:
            idx_fb[src] = OFF_PB[li] + _plane_index(n, k, K)
        if OFF_PTB[li] is not None:     # W^T: K outputs, N reduced
            idx_tb[src] = OFF_PTB[li] + _plane_index(k, n, N)
    for idx, size in ((idx_fb, PB_HALVES), (idx_tb, PTB_HALVES)):
        used = idx[idx >= 0].astype(np.int64)
        allpos = np.concatenate([used, used + 512, used + 1024])
        assert len(np.unique(allpos)) == len(allpos) and allpos.max() < size
    assert 3 * (idx_fb >= 0).sum() == PB_HALVES and 3 * (idx_tb >= 0).sum() == PTB_HALVES
    return idx_fb, idx_tb


# fp16x2 planes of the fused optimizer step (csrc/mlp_fused_h2.inc): PB's / PTB's layout with two terms per k block
PH_HALVES, PTH_HALVES = PB_HALVES * 2 // 3, PTB_HALVES * 2 // 3
H2_SCALE_FLOATS, H2_INV, H2_W0 = 48, 16, 8
H2_CLASSES = ("x", "h1", "h2", "h3", "dz4", "dz3", "dz2", "dz1")


def split_f16x2(w, scale):
    """fp32 tensor -> two int16 tensors holding the fp16 terms h0 + h1 ~= w * scale (|error| <= 2^-23 |w scale|)."""
    y = w * scale
    h0 = y.to(torch.float16)
    h1 = (y - h0.float()).to(torch.float16)
    return [t.view(torch.int16) for t in (h0, h1)]


def h2_weight_scale(wmax):
    """The power of two mlp_h2_rescale gives a layer whose largest |w| is `wmax`: max(wmax, 2^-4) lands in [2^11, 2^12)."""
    import math
    m = min(max(float(wmax), 2.0 ** -4), 2.0 ** 60)
    return 2.0 ** (11 - math.floor(math.log2(m)))


def untile(t, n, N):
    """[rows][N] view of a saved activation / dZ buffer, which the kernels keep in tile-fragment order
    (csrc/mlp_mfma.hip, frag_off): per 32-row tile and 32-column tile a block of 1024 floats
    [g][lane][4] with lane = 32 * ((col % 8) // 4) + row % 32, g = (col % 32) // 8."""
    rows = torch.arange(n, device=t.device).view(-1, 1)
    cols = torch.arange(N, device=t.device).view(1, -1)
    c = cols % 32
    off = ((rows // 32) * (N // 32) + cols // 32) * 1024 + ((c // 8) * 64 + ((c // 4) % 2) * 32 + rows % 32) * 4 + c % 4
    return t.reshape(-1)[off]


def split_bf16x3(w):
    """fp32 tensor -> three int16 tensors holding the bf16 terms w0 + w1 + w2 == w (exact)."""
    w0 = w.to(torch.bfloat16)
    r1 = w - w0.float()
    w1 = r1.to(torch.bfloat16)
    w2 = (r1 - w1.float()).to(torch.bfloat16)
    return [t.view(torch.int16) for t in (w0, w1, w2)]


class PackedPolicy:
    def __init__(self, net, device):
        self.device = torch.device(device)
        self._lib = _lib.load()
        self.P = torch.zeros(PACKED, dtype=torch.float32, device=self.device)
        self.PF = torch.zeros(FRAG, dtype=torch.float32, device=self.device)      # forward operands, fragment order
        self.PT = torch.zeros(FRAG_T, dtype=torch.float32, device=self.device)    # W^T operands, fragment order
        # GEMM arithmetic of the MFMA kernels: "bf16x3" (default) = three-term bf16 split of both fp32 operands on
        # v_mfma_f32_32x32x16_bf16, fp32 accumulate (held to the reference's golden vectors at the fp32 tolerances and to fp64:
        # tests/test_ppo_gpu.py, tests/test_mlp_train_gpu.py, csrc/mlp_layout.h); "f32" = v_mfma_f32_32x32x2_f32 (FLY_GEMM=f32)
        env_gemm = os.environ.get("FLY_GEMM")               # names EVERY GEMM when given ("f16x2": bf16x3 + the fp16x2 optimizer step)
        self._gemm = "bf16x3" if env_gemm in (None, "f16x2") else env_gemm
        self.gemm_infer = os.environ.get("FLY_GEMM_INFER", self._gemm)
        assert self._gemm in ("f32", "bf16x3") and self.gemm_infer in ("f32", "bf16x3")
        self.PB = torch.zeros(PB_HALVES, dtype=torch.int16, device=self.device)
        self.PTB = torch.zeros(PTB_HALVES, dtype=torch.int16, device=self.device)
        # Arithmetic of the fused optimizer-step gradient (only): "bf16x3" (mlp_fused_grad) or "f16x2" (mlp_fused_grad_h2: two fp16 terms
        # per operand, three products per k block, per-class power-of-two scales; csrc/mlp_fused_h2.inc).  The rollout's policy, the
        # critic pass and every fallback stay on `gemm`.  FLY_STEP_GEMM / --step_gemm / policy.step_gemm = ...
        # DEFAULT (round 5): f16x2, with bf16x3 as the A/B and as the fallback of a refused step; `policy.gemm = "bf16x3"` (or
        # FLY_GEMM=bf16x3) names every GEMM, the step included.
        self._step_gemm = os.environ.get("FLY_STEP_GEMM", "f16x2" if env_gemm in (None, "f16x2") else "bf16x3")
        assert self._step_gemm in ("bf16x3", "f16x2")
        self.PH = torch.zeros(PH_HALVES, dtype=torch.int16, device=self.device)
        self.PTH = torch.zeros(PTH_HALVES, dtype=torch.int16, device=self.device)
        self.h2_scales = torch.zeros(H2_SCALE_FLOATS, dtype=torch.float32, device=self.device)
        self.h2_overflow = torch.zeros(1, dtype=torch.int32, device=self.device)     # sticky: a launch's values did not fit fp16
        self.h2_freeze = False              # tests: the reduction leaves the scale table alone
        self.h2_calibrated = False
        self._h2_steps_since_rescale = 0
        self.h2_suspended = False           # True while refused steps are redone on the bf16x3 kernel (the planes stay maintained)
        self.h2_overflows = 0               # updates in which the fp16x2 step was refused and redone on bf16x3
        self._h2_reset_scales()
        idx_fb, idx_tb = build_plane_maps()
        self.idx_fb = torch.from_numpy(idx_fb).to(self.device)
        self.idx_tb = torch.from_numpy(idx_tb).to(self.device)
        self._src_fb = torch.nonzero(self.idx_fb >= 0).squeeze(-1)
        self._dst_fb = self.idx_fb[self._src_fb].long()
        self._src_tb = torch.nonzero(self.idx_tb >= 0).squeeze(-1)
        self._dst_tb = self.idx_tb[self._src_tb].long()
        idx_f, idx_t = build_index_maps()
        self.idx_f = torch.from_numpy(idx_f).to(self.device)
        self.idx_t = torch.from_numpy(idx_t).to(self.device)
        self._src_f = torch.nonzero(self.idx_f >= 0).squeeze(-1)
        self._dst_f = self.idx_f[self._src_f].long()
        self._src_t = torch.nonzero(self.idx_t >= 0).squeeze(-1)
        self._dst_t = self.idx_t[self._src_t].long()
        P = self.P
        self.W1 = P[OFF_W1:OFF_B1].view(H1, IN_PAD)
        self.b1 = P[OFF_B1:OFF_W2]
        self.W2 = P[OFF_W2:OFF_B2].view(H2, H1)
        self.b2 = P[OFF_B2:OFF_W3]
        self.W3 = P[OFF_W3:OFF_B3].view(H3, H2)
        self.b3 = P[OFF_B3:OFF_W4]
        self.W4 = P[OFF_W4:OFF_B4].view(OUT, H3)
        self.b4 = P[OFF_B4:PACKED]
        self.views = {
            "shared_net.0.weight": self.W1[:, :IN], "shared_net.0.bias": self.b1,
            "shared_net.2.weight": self.W2, "shared_net.2.bias": self.b2,
            "to_mean.0.weight": self.W3[:64], "to_mean.0.bias": self.b3[:64],
            "to_value.0.weight": self.W3[64:], "to_value.0.bias": self.b3[64:],
            "to_mean.2.weight": self.W4[:NACT, :64], "to_mean.2.bias": self.b4[:NACT],
            "to_value.2.weight": self.W4[NACT:NACT + 1, 64:], "to_value.2.bias": self.b4[NACT:NACT + 1],
        }
        params = dict(net.named_parameters())
        assert set(params) == set(self.views), "Net does not have the reference's parameter set"
        with torch.no_grad():
            for k, view in self.views.items():
                view.copy_(params[k].data.to(self.device))
                params[k].data = view                       # one copy of truth
        mask = torch.zeros(PACKED, dtype=torch.float32, device=self.device)
        for k, view in self.views.items():
            off = view.storage_offset()
            idx = torch.arange(PACKED, device=self.device).as_strided(view.shape, view.stride(), off)
            mask[idx.reshape(-1)] = 1.0
        self.grad_mask = mask
        assert int(mask.sum().item()) == 69587            # every reference parameter exactly once
        self.refresh()

    def refresh(self):
        """Rebuild the fragment-ordered copies from the master weights (after a load or any
        out-of-band change of the parameters; mlp_adam_step keeps them in step by itself)."""
        with torch.no_grad():
            self.PF[self._dst_f] = self.P[self._src_f]
            self.PT[self._dst_t] = self.P[self._src_t]
            self._refresh_planes()
        self.version += 1

    refresh_transposes = refresh

    def _refresh_planes(self):
        with torch.no_grad():
            for dst_buf, src, dst in ((self.PB, self._src_fb, self._dst_fb), (self.PTB, self._src_tb, self._dst_tb)):
                for term, plane in enumerate(split_bf16x3(self.P[src])):
                    dst_buf[dst + 512 * term] = plane

        if self.h2_live():
            self._refresh_planes_h2()

    def h2_live(self):
        return self._step_gemm == "f16x2" and self._gemm == "bf16x3"

    def _h2_reset_scales(self):
        """Starting scales of the activation / gradient classes (calibrate_h2 replaces them by measured ones)."""
        s = torch.ones(H2_SCALE_FLOATS)
        s[:4] = 2.0 ** 4                    # x, h1 .. h3: O(1 .. 100)
        s[4:8] = 2.0 ** 28                  # dz4 .. dz1: O(1e-6)
        s[H2_INV:H2_INV + 16] = 1.0 / s[:16]
        s[32:] = 0.0
        self.h2_scales.copy_(s)
        self.h2_calibrated = False

    def _refresh_planes_h2(self):
        """fp16x2 weight planes and their per-layer scales from the master weights (`mlp_h2_rescale`: one small launch, no host sync).
        The scales then stay fixed while mlp_adam_step splits the updated weights under them; an Adam step moves a weight by at most
        3.2 lr, and overflowing needs a weight to travel 15/16 at least, so a call every `0.9 / (3.2 lr)` steps keeps fp16 safe."""
        if self.device.type != "cuda":
            return                                  # (layout-only uses of the class on the CPU: tests/test_dist_cpu.py)
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        _lib.check(self._lib.mlp_h2_rescale(p(self.P), p(self.idx_fb), p(self.idx_tb), p(self.PH), p(self.PTH), p(self.h2_scales),
                                            _lib.stream_ptr()), "mlp_h2_rescale")
        self._h2_steps_since_rescale = 0

    def _planes_live(self):
        return self._gemm == "bf16x3" or self.gemm_infer == "bf16x3"

    @property
    def step_gemm(self):
        return self._step_gemm if self._gemm == "bf16x3" else self._gemm

    @step_gemm.setter
    def step_gemm(self, mode):
        assert mode in ("bf16x3", "f16x2")
        was = self.h2_live()
        self._step_gemm = mode
        if self.h2_live() and not was:
            self._refresh_planes_h2()

    @property
    def gemm(self):
        return self._gemm

    @gemm.setter
    def gemm(self, mode):
        """Arithmetic of EVERY MLP GEMM of this policy: the update's forward, dX chain and dW, the rollout's
        policy launch and the critic pass.  Switching to bf16x3 rebuilds the term planes: the Adam kernel only
        maintains them while a bf16x3 mode is active (six extra scattered stores per weight otherwise wasted)."""
        if mode == "f16x2":                 # bf16x3 everywhere, the fused optimizer step in fp16x2 (the default configuration)
            self.gemm = "bf16x3"
            self.step_gemm = "f16x2"
            return
        assert mode in ("f32", "bf16x3")
        was_live, was_h2 = self._planes_live(), self.h2_live()
        self._gemm = mode
        self.gemm_infer = mode
        self._step_gemm = "bf16x3"          # an explicit arithmetic names every GEMM, the optimizer step included
        if self._planes_live() and not was_live:
            self._refresh_planes()
        elif self.h2_live() and not was_h2:
            self._refresh_planes_h2()
    version = 0         # bumped whenever the weights change: consumers of cached network outputs compare it

    def _plane_args(self):
        if not self._planes_live():
            return (None, None, None, None)
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        return (p(self.PB), p(self.PTB), p(self.idx_fb), p(self.idx_tb))

    def _h2_args(self):
        if not self.h2_live():
            return (None, None, None, C.c_int(0))
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        # a RESCALE step (the kernel re-derives the weight scales from the weights before it splits under them) often enough that a
        # weight cannot have drifted out of the scale's 16x headroom: an Adam step moves it by at most 3.2 lr
        rescale = self._h2_steps_since_rescale >= max(1, min(64, int(0.9 / (3.2 * self.lr))))
        self._h2_steps_since_rescale = 1 if rescale else self._h2_steps_since_rescale + 1
        return (p(self.PH), p(self.PTH), p(self.h2_scales), C.c_int(1 if rescale else 0))

    def pb_ptr(self):
        """Term planes for the UPDATE's forward (minibatch_grad); inference launches (rollout policy,
        critic rows, Net.pi / Net.v) stay on the fp32 MFMA body, which at one tile per CU is bound by
        latency, not by matrix time."""
        return C.c_void_p(self.PB.data_ptr()) if self.gemm == "bf16x3" else None

    def infer_pb_ptr(self):
        return C.c_void_p(self.PB.data_ptr()) if self.gemm_infer == "bf16x3" else None

    def ptb_ptr(self):
        return C.c_void_p(self.PTB.data_ptr()) if self.gemm == "bf16x3" else None

    def forward(self, x, want_mu=True, want_v=True, saves=None):
        """x f32 [..., 73] on the device -> (mu [..., 18] | None, v [..., 1] | None)."""
        lead = x.shape[:-1]
        x2 = x.reshape(-1, IN)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        n = x2.shape[0]
        mu = torch.empty((n, NACT), device=self.device) if want_mu else None
        v = torch.empty((n,), device=self.device) if want_v else None
        s = saves or {}
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
        _lib.check(self._lib.mlp_forward(ptr(self.P), ptr(self.PF), ptr(x2), C.c_int64(n), ptr(mu), ptr(v), ptr(s.get("out")),
                                         ptr(s.get("h1")), ptr(s.get("h2")), ptr(s.get("h3")),
                                         self.pb_ptr() if saves else self.infer_pb_ptr(), _lib.stream_ptr()),
                   "mlp_forward")
        return (mu.view(*lead, NACT) if want_mu else None), (v.view(*lead, 1) if want_v else None)

    # ---- training on the MFMA kernels (ppo.py:184-199) ------------------------------------------
    def init_training(self, max_rows, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0):
        dev = self.device
        self.lr, self.betas, self.eps, self.max_norm = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(max_norm)
        self.G = torch.zeros(PACKED, device=dev)
        self.exp_avg = torch.zeros(PACKED, device=dev)
        self.exp_avg_sq = torch.zeros(PACKED, device=dev)
        self._step2 = torch.zeros(2, dtype=torch.int32, device=dev)    # the step counter (two words: see adam_step)
        self._step_idx = 0
        self.steps_issued = 0              # adam_step calls so far; the device counter lags iff a step was refused (fail closed)
        self._norm_ws = torch.zeros(1280, device=dev)      # [0] = pre-clip gradient norm, rest partial sums
        self.grad_norm = self._norm_ws[:1]
        self.workspace = torch.empty(int(self._lib.mlp_grad_workspace_floats()), device=dev)
        self.max_rows = int(max_rows)
        r = (self.max_rows + 31) // 32 * 32        # whole 32-row tiles: the saved tensors are stored tile by tile
        self.saves = {"out": torch.empty(r, OUT, device=dev), "h1": torch.empty(r, H1, device=dev),
                      "h2": torch.empty(r, H2, device=dev), "h3": torch.empty(r, H3, device=dev)}
        self.dz = {"dz4": torch.empty(r, OUT, device=dev), "dz3": torch.empty(r, H3, device=dev),
                   "dz2": torch.empty(r, H2, device=dev), "dz1": torch.empty(r, H1, device=dev)}
        self.loss_part = torch.zeros((r + 31) // 32, 2, device=dev)
        # mlp_forward_backward: per-tile "forward done" words (compared against a per-call epoch,
        # never reset) and the word a workgroup sets if it ever gives up waiting (stays 0)
        self._tile_flags = torch.zeros((r + 31) // 32, dtype=torch.int32, device=dev)
        self.tile_wait_error = torch.zeros(1, dtype=torch.int32, device=dev)
        self._epoch = 0
        # bf16x3 only: the whole minibatch gradient in ONE persistent launch (csrc/mlp_fused_step.inc: forward, loss, dX chain and
        # dW of a tile in the workgroup that owns it; nothing but x, the per-row scalars and one partial slab per CU touches
        # HBM).  FLY_FUSED_STEP=0 keeps the three-launch path (the A/B).
        self.fused_step = os.environ.get("FLY_FUSED_STEP", "1") != "0"
        self._fused_ws = None
        self.fuse_fwd_bwd = os.environ.get("FLY_FUSE_FWD_BWD", "1") != "0"
        # how a tile travels from its forward to its backward workgroup inside the one launch
        # (csrc/mlp_backward.inc): "sc1" = write-through stores + L1-bypassing loads, no placement
        # assumption (default); "xcd" = plain accesses, producer and consumer must share an XCD
        self.handoff = os.environ.get("FLY_FWD_BWD_HANDOFF", "sc1")
        assert self.handoff in ("sc1", "xcd")
        if self.fuse_fwd_bwd:
            self._probe_fused_launch()

    def _probe_fused_launch(self):
        """One small fused launch over 64 tiles (8 per XCD) at start-up: if a backward workgroup cannot
        get its tile (no in-order dispatch: err 1; "xcd" hand-off on a device that places workgroups
        differently: err 2) the two-launch path is used from the start -- same results."""
        n = min(self.max_rows, 64 * 32)
        z = lambda *shape: torch.zeros(*shape, device=self.device)   # noqa: E731
        was, self.h2_suspended = self.h2_suspended, True        # (not a launch whose values the fp16x2 scales should be judged on)
        try:
            self.minibatch_grad(z(n, IN), z(n, NACT), z(n), z(n), z(n), torch.full((NACT,), 0.2, device=self.device), 0.2)
        finally:
            self.h2_suspended = was
        self.check_fused_launch()

    def update_can_be_refused(self):
        """True when a launch of the current update path can leave an invalid gradient that the optimizer kernels then refuse on
        the device (the fused forward+backward launch's tile hand-off; a value of the fp16x2 step that did not fit fp16): the caller
        must read the step counter after the update and redo what was refused.  The bf16x3 fused optimizer step (`mlp_fused_grad`)
        and the two-launch path cannot."""
        if self.fused_step and self.gemm == "bf16x3":
            return self.h2_live() and os.environ.get("FLY_H2_DIAG_NO_STEP_READ") != "1"     # (diagnostic: what the blocking read costs)
        return bool(self.fuse_fwd_bwd)

    def check_fused_launch(self):
        """Host sync.  Returns the err word of the fused launches since the last check (0 = all fine) and
        clears it; a nonzero word switches this policy to the two-launch path.  The optimizer kernels
        have already refused every step whose gradient came from a failed launch (they fail closed on
        the device), so the caller only has to redo those minibatches (PPO._update_hip)."""
        err = int(self.tile_wait_error.item())
        if err != 0:
            self.fuse_fwd_bwd = False
            self.fused_launch_failures = getattr(self, "fused_launch_failures", 0) + 1
            self.tile_wait_error.zero_()
        return err

    def minibatch_grad(self, x, action, old_logp, adv, target, var, clip, global_rows=None, fuse_norm=False, dump=False):
        """Forward + loss + backward of one minibatch; leaves the packed gradient in `self.G`.
        `fuse_norm` (single rank): the partial reduction also prepares the clip norm and advances the
        step, so `adam_step(norm_ready=True)` is one launch."""
        n = x.shape[0]
        assert n <= self.max_rows and x.is_contiguous() and action.is_contiguous()
        for t in (old_logp, adv, target):
            assert t.is_contiguous() and t.numel() == n
        s, d = self.saves, self.dz
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        st = _lib.stream_ptr()
        inv_b = 1.0 / float(global_rows if global_rows else n)
        if self.fused_step and self.gemm == "bf16x3":
            return self._fused_grad(x, action, old_logp, adv, target, var, clip, inv_b, fuse_norm, dump)
        if dump:
            raise ValueError("dump=True is a debugging aid of the fused step")
        if self.fuse_fwd_bwd:
            # one launch: the backward workgroup of a row tile starts when that tile's forward is done
            if self._epoch >= (1 << 27) - 1:
                self._tile_flags.zero_()
                self._epoch = 0
            self._epoch += 1
            _lib.check(self._lib.mlp_forward_backward(
                p(self.P), p(self.PF), p(self.PT), p(x), C.c_int64(n), p(s["out"]), p(s["h1"]), p(s["h2"]), p(s["h3"]),
                p(action), p(old_logp), p(adv), p(target), p(var), C.c_float(inv_b), C.c_float(clip),
                p(d["dz4"]), p(d["dz3"]), p(d["dz2"]), p(d["dz1"]), p(self.loss_part), p(self._tile_flags),
                C.c_int(self._epoch), p(self.tile_wait_error), self.pb_ptr(), self.ptb_ptr(),
                C.c_int(1 if self.handoff == "sc1" else 0), st), "mlp_forward_backward")
        else:
            _lib.check(self._lib.mlp_forward(p(self.P), p(self.PF), p(x), C.c_int64(n), None, None, p(s["out"]), p(s["h1"]),
                                             p(s["h2"]), p(s["h3"]), self.pb_ptr(), st), "mlp_forward")
            _lib.check(self._lib.mlp_backward_dx(p(self.PT), p(s["out"]), p(s["h1"]), p(s["h2"]), p(s["h3"]), p(action),
                                                 p(old_logp), p(adv), p(target), p(var), C.c_int64(n), C.c_float(inv_b),
                                                 C.c_float(clip), p(d["dz4"]), p(d["dz3"]), p(d["dz2"]), p(d["dz1"]),
                                                 p(self.loss_part), self.ptb_ptr(), st), "mlp_backward_dx")
        nm = (p(self.grad_mask), p(self._norm_ws), p(self.step)) if fuse_norm else (None, None, None)
        _lib.check(self._lib.mlp_grad_w(p(x), p(s["h1"]), p(s["h2"]), p(s["h3"]), p(d["dz1"]), p(d["dz2"]),
                                        p(d["dz3"]), p(d["dz4"]), C.c_int64(n), p(self.workspace), p(self.G), *nm,
                                        p(self.tile_wait_error), C.c_int(1 if self.gemm == "bf16x3" else 0), st),
                   "mlp_grad_w")

    def _fused_grad(self, x, action, old_logp, adv, target, var, clip, inv_b, fuse_norm, dump):
        """`mlp_fused_grad`: one persistent launch + the slab reduction.  dump=True (tests) also writes the chain's values into
        self.saves / self.dz exactly as the three-launch path leaves them."""
        n = x.shape[0]
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        if self._fused_ws is None:
            self._fused_ws = torch.empty(int(max(self._lib.mlp_fused_workspace_floats(), self._lib.mlp_fused_h2_workspace_floats())),
                                         device=self.device)
        nm = (p(self.grad_mask), p(self._norm_ws), p(self.step)) if fuse_norm else (None, None, None)
        dptr = None
        if dump:
            s, d = self.saves, self.dz
            arr = (C.c_void_p * 8)(*[t.data_ptr() for t in (s["out"], s["h1"], s["h2"], s["h3"], d["dz4"], d["dz3"], d["dz2"], d["dz1"])])
            dptr = arr
        if self.h2_live() and not self.h2_suspended:
            _lib.check(self._lib.mlp_fused_grad_h2(p(self.P), p(self.PH), p(self.PTH), p(self.h2_scales), p(self.h2_overflow),
                                                   C.c_int(1 if self.h2_freeze else 0), p(x), C.c_int64(n), p(action), p(old_logp),
                                                   p(adv), p(target), p(var), C.c_float(inv_b), C.c_float(clip), p(self._fused_ws),
                                                   p(self.G), *nm, p(self.loss_part), dptr, _lib.stream_ptr()), "mlp_fused_grad_h2")
            return
        _lib.check(self._lib.mlp_fused_grad(p(self.P), p(self.PB), p(self.PTB), p(x), C.c_int64(n), p(action), p(old_logp), p(adv),
                                            p(target), p(var), C.c_float(inv_b), C.c_float(clip), p(self._fused_ws), p(self.G), *nm,
                                            p(self.loss_part), dptr, _lib.stream_ptr()), "mlp_fused_grad")

    def calibrate_h2(self, x, action, old_logp, adv, target, var, clip, global_rows=None, max_launches=12):
        """Bring the activation / gradient scales of the fp16x2 step to the data (host-synchronising; before the first update on a new
        network or after an overflow): launch the gradient on one minibatch -- the result is discarded, the step counter untouched --
        until a launch fits fp16 and leaves the scales it found.  Every launch sets each class's scale from the maximum it saw, and an
        overflowing class garbles only what lies downstream of it, so the table settles in at most one launch per class.  Returns
        the number of launches."""
        assert self.h2_live()
        frozen, self.h2_freeze = self.h2_freeze, False
        try:
            for k in range(1, max_launches + 1):
                self.h2_overflow.zero_()
                before = self.h2_scales[:8].clone()
                self.minibatch_grad(x, action, old_logp, adv, target, var, clip, global_rows=global_rows, fuse_norm=False)
                fit = int(self.h2_overflow.item()) == 0
                if fit and torch.equal(before, self.h2_scales[:8]):
                    self.h2_calibrated = True
                    return k
            raise _lib.FlyHipError("calibrate_h2: the fp16x2 scales did not settle in %d launches (maxima %s)"
                                   % (max_launches, self.h2_scales[32:40].tolist()))
        finally:
            self.h2_overflow.zero_()
            self.h2_freeze = frozen

    def update_path(self):
        """Which launches one optimizer step is made of (for the bench line)."""
        if self.fused_step and self.h2_live():
            return ("mlp_fused_grad_h2 (ONE persistent launch: forward + loss + dX chain + dW per tile in fp16x2, slabs reduced, scales "
                    "tracked) + mlp_adam_step, gemm=bf16x3, step_gemm=f16x2")
        if self.fused_step and self.gemm == "bf16x3":
            return "mlp_fused_grad (ONE persistent launch: forward + loss + dX chain + dW per tile, slabs reduced) + mlp_adam_step, gemm=bf16x3"
        if self.fuse_fwd_bwd:
            return "mlp_forward_backward (one launch, per-tile flags) + mlp_grad_w (+reduce) + mlp_adam_step, gemm=" + self.gemm
        return "mlp_forward + mlp_backward_dx + mlp_grad_w (+reduce) + mlp_adam_step, gemm=" + self.gemm

    def loss_value(self, n):
        """ppo.py:194/:197 scalar of the last minibatch_grad call (diagnostics; one small reduction)."""
        parts = self.loss_part[: (n + 31) // 32].sum(0)
        return (parts[0] + parts[1]) / n

    @property
    def step(self):
        """Device int32 [1]: optimizer steps applied so far."""
        return self._step2[self._step_idx:self._step_idx + 1]

    def adam_step(self, grad_scale=1.0, norm_ready=False, self_norm=False, grad_invalid=None):
        """clip_grad_norm_ + Adam on the packed parameters.  norm_ready: mlp_grad_w has already left the norm
        partials and advanced the step (single rank).  self_norm: ONE launch that also sums the gradient
        (data-parallel ranks: G comes out of the all-reduce); the step counter then ping-pongs between two
        device words so that no workgroup reads a value another one has already advanced.  grad_invalid: optional int32 device
        word (the err word of the peer-to-peer exchange); nonzero = the launch refuses the step (fail closed)."""
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        self.version += 1
        self.steps_issued += 1
        step_in = self.step
        step_out = None
        if self_norm:
            self._step_idx ^= 1
            step_out = p(self.step)
        _lib.check(self._lib.mlp_adam_step(p(self.P), p(self.PF), p(self.PT), p(self.idx_f), p(self.idx_t), p(self.G),
                                           p(self.grad_mask), p(self.exp_avg),
                                           p(self.exp_avg_sq), p(step_in), C.c_float(self.lr),
                                           C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                                           C.c_float(self.max_norm), C.c_float(grad_scale), p(self._norm_ws),
                                           C.c_int(1 if norm_ready else 0), *self._plane_args(), step_out,
                                           p(grad_invalid) if grad_invalid is not None else None, *self._h2_args(), _lib.stream_ptr()),
                   "mlp_adam_step")
