"""Packed parameter storage of `Net` and the MFMA forward (`mlp_forward`, include/flyhip.h).

`PackedPolicy` owns ONE flat fp32 buffer in the layout of csrc/mlp_layout.h and re-points the
`.data` of every `Net` parameter at a strided view into it, so torch (checkpoint I/O, autograd,
optimizers) and the HIP kernels see the same memory.  Padding and the structural zeros of the
stacked last layer never receive gradient (`grad_mask`).
"""
import ctypes as C

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
OFF_WT2 = 0
OFF_WT3 = OFF_WT2 + H1 * H2
OFF_WT4 = OFF_WT3 + H2 * H3
PACKED_T = OFF_WT4 + H3 * OUT              # 53248


class PackedPolicy:
    def __init__(self, net, device):
        self.device = torch.device(device)
        self._lib = _lib.load()
        self.P = torch.zeros(PACKED, dtype=torch.float32, device=self.device)
        self.PT = torch.zeros(PACKED_T, dtype=torch.float32, device=self.device)
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
        self.refresh_transposes()

    def refresh_transposes(self):
        """W^T copies streamed by the backward dX chain (call after the weights change)."""
        with torch.no_grad():
            self.PT[OFF_WT2:OFF_WT3].view(H1, H2).copy_(self.W2.t())
            self.PT[OFF_WT3:OFF_WT4].view(H2, H3).copy_(self.W3.t())
            self.PT[OFF_WT4:PACKED_T].view(H3, OUT).copy_(self.W4.t())

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
        _lib.check(self._lib.mlp_forward(ptr(self.P), ptr(x2), C.c_int64(n), ptr(mu), ptr(v), ptr(s.get("out")),
                                         ptr(s.get("h1")), ptr(s.get("h2")), ptr(s.get("h3")), _lib.stream_ptr()),
                   "mlp_forward")
        return (mu.view(*lead, NACT) if want_mu else None), (v.view(*lead, 1) if want_v else None)

    # ---- training on the MFMA kernels (ppo.py:184-199) ------------------------------------------
    def init_training(self, max_rows, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0):
        dev = self.device
        self.lr, self.betas, self.eps, self.max_norm = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(max_norm)
        self.G = torch.zeros(PACKED, device=dev)
        self.exp_avg = torch.zeros(PACKED, device=dev)
        self.exp_avg_sq = torch.zeros(PACKED, device=dev)
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)
        self._norm_ws = torch.zeros(128, device=dev)       # [0] = pre-clip gradient norm, rest scratch
        self.grad_norm = self._norm_ws[:1]
        self.workspace = torch.empty(int(self._lib.mlp_grad_workspace_floats()), device=dev)
        self.max_rows = int(max_rows)
        r = self.max_rows
        self.saves = {"out": torch.empty(r, OUT, device=dev), "h1": torch.empty(r, H1, device=dev),
                      "h2": torch.empty(r, H2, device=dev), "h3": torch.empty(r, H3, device=dev)}
        self.dz = {"dz4": torch.empty(r, OUT, device=dev), "dz3": torch.empty(r, H3, device=dev),
                   "dz2": torch.empty(r, H2, device=dev), "dz1": torch.empty(r, H1, device=dev)}
        self.loss_part = torch.zeros((r + 31) // 32, 2, device=dev)

    def minibatch_grad(self, x, action, old_logp, adv, target, var, clip, global_rows=None):
        """Forward + loss + backward of one minibatch; leaves the packed gradient in `self.G`."""
        n = x.shape[0]
        assert n <= self.max_rows and x.is_contiguous() and action.is_contiguous()
        for t in (old_logp, adv, target):
            assert t.is_contiguous() and t.numel() == n
        s, d = self.saves, self.dz
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        st = _lib.stream_ptr()
        _lib.check(self._lib.mlp_forward(p(self.P), p(x), C.c_int64(n), None, None, p(s["out"]), p(s["h1"]),
                                         p(s["h2"]), p(s["h3"]), st), "mlp_forward")
        inv_b = 1.0 / float(global_rows if global_rows else n)
        _lib.check(self._lib.mlp_backward_dx(p(self.PT), p(s["out"]), p(s["h1"]), p(s["h2"]), p(s["h3"]), p(action),
                                             p(old_logp), p(adv), p(target), p(var), C.c_int64(n), C.c_float(inv_b),
                                             C.c_float(clip), p(d["dz4"]), p(d["dz3"]), p(d["dz2"]), p(d["dz1"]),
                                             p(self.loss_part), st), "mlp_backward_dx")
        _lib.check(self._lib.mlp_grad_w(p(x), p(s["h1"]), p(s["h2"]), p(s["h3"]), p(d["dz1"]), p(d["dz2"]),
                                        p(d["dz3"]), p(d["dz4"]), C.c_int64(n), p(self.workspace), p(self.G), st),
                   "mlp_grad_w")

    def loss_value(self, n):
        """ppo.py:194/:197 scalar of the last minibatch_grad call (diagnostics; one small reduction)."""
        parts = self.loss_part[: (n + 31) // 32].sum(0)
        return (parts[0] + parts[1]) / n

    def adam_step(self, grad_scale=1.0):
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        _lib.check(self._lib.mlp_adam_step(p(self.P), p(self.PT), p(self.G), p(self.grad_mask), p(self.exp_avg),
                                           p(self.exp_avg_sq), p(self.step), C.c_float(self.lr),
                                           C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                                           C.c_float(self.max_norm), C.c_float(grad_scale), p(self._norm_ws),
                                           _lib.stream_ptr()), "mlp_adam_step")
