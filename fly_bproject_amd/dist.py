"""Data-parallel glue: one process per GPU, envs sharded embarrassingly, policy replicated.

The only exchange on the path is the policy gradient: one all-reduce (RCCL over xGMI on the GPU
box; gloo in the CPU tests) of ONE flat fp32 buffer holding all 69 587 gradient elements
(278 KB) per optimizer step, averaged, before `clip_grad_norm_` so that the clip sees the global
gradient.  At this size the collective is latency-bound, so it is issued as a single call on a
single buffer: parameter `.grad`s are views into the flat buffer, nothing is packed or unpacked.
(The reference has no distributed code; ppo.py:196-199 is where the call sits.)
"""
import os

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    def __init__(self, params, world_size, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.world_size = int(world_size)
        self.group = group
        total = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(total, dtype=p0.dtype, device=p0.device)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)     # autograd accumulates in place into the view
            off += n

    def zero(self):
        self.flat.zero_()

    def allreduce_mean(self):
        for p in self.params:                              # zero_grad(set_to_none=True) would break the aliasing
            if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or \
                    p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * self.flat.element_size():
                raise RuntimeError("a parameter's .grad no longer aliases the flat buffer; "
                                   "use optimizer.zero_grad(set_to_none=False)")
        if self.world_size > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(self.world_size)


class P2PUnavailable(RuntimeError):
    """Raised by P2PAllReduce() on EVERY rank alike when any rank could not allocate, export or open a window."""


class P2PAllReduce:
    """One-shot peer-to-peer all-reduce (sum) of ONE flat fp32 device buffer across the ranks of a node
    (`dp_allreduce_p2p`, csrc/dp_p2p.hip): every rank publishes its buffer in a fine-grained window the peers
    have opened through hipIpc, and sums all windows in rank order -- one launch, one xGMI hop, bit-identical
    results on every rank.  The window handles travel once, at construction, through torch.distributed
    (any backend).

    Construction is a sequence of STAGES that every rank walks in the same order whatever happens locally: each
    fallible local step (allocate + export; open the peers' windows) runs under try/except and is followed by a
    collective in which every rank takes part and learns every rank's outcome.  If any rank failed a stage, every
    rank releases what it holds and raises `P2PUnavailable` -- no rank is ever left inside a collective its peers
    have skipped.

    `allreduce_(t)` is asynchronous on the current stream.  A rank whose bounded wait for a peer expires leaves `t`
    un-reduced, marks it invalid (`fail_slot`: the optimizer kernels then refuse it on that rank) and sets the
    error word; `check()` (a host sync) reports that, and the caller must stop the run: there is no in-flight
    fall-back to the collective, because the peers cannot know that this rank gave up."""

    def __init__(self, n_floats, device, group=None, fail_slot=-1):
        import ctypes as C
        from . import _lib
        self._C, self._lib, self._libmod = C, _lib.load(), _lib
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._group = group
        self.n = int(n_floats)
        self.fail_slot = int(fail_slot)
        if self.n % 4 or self.world > 16:
            raise ValueError("P2PAllReduce: n_floats must be a multiple of 4 and world <= 16")   # same verdict on every rank
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self._mine = C.c_void_p()
        self._opened = []
        self._windows = (C.c_void_p * self.world)()
        # stage 1: allocate + export, then everybody learns everybody's handle (or that there is none)
        mine, why = None, ""
        try:
            _lib.check(self._lib.dp_p2p_alloc(C.c_int64(self.n), C.byref(self._mine)), "dp_p2p_alloc")
            handle = (C.c_uint8 * 64)()
            _lib.check(self._lib.dp_ipc_export(self._mine, handle), "dp_ipc_export")
            mine = bytes(handle)
        except Exception as e:      # noqa: BLE001
            why = "rank %d: %r" % (self.rank, e)
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=group)
        if any(h is None for h in handles):
            self._release(collective=False)
            raise P2PUnavailable(why or "rank(s) %s could not allocate or export a window" %
                                 [r for r, h in enumerate(handles) if h is None])
        # stage 2: open the peers' windows, then everybody learns whether everybody could
        ok = True
        try:
            for r, h in enumerate(handles):
                if r == self.rank:
                    self._windows[r] = self._mine.value
                else:
                    ptr = C.c_void_p()
                    buf = (C.c_uint8 * 64).from_buffer_copy(h)
                    _lib.check(self._lib.dp_ipc_import(buf, C.byref(ptr)), "dp_ipc_import (rank %d)" % r)
                    self._windows[r] = ptr.value
                    self._opened.append(ptr)
        except Exception as e:      # noqa: BLE001
            ok, why = False, "rank %d: %r" % (self.rank, e)
        votes = [None] * self.world
        dist.all_gather_object(votes, (ok, why), group=group)           # doubles as the barrier: every window is open before anybody publishes
        if not all(v[0] for v in votes):
            self._release(collective=True)
            raise P2PUnavailable("; ".join(v[1] for v in votes if not v[0]))
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.epoch = 0

    def _release(self, collective):
        """Close what this rank opened; free its own window only after EVERY rank has closed (collective=True: all
        ranks are here together, so a barrier is safe; False: nobody opened anything yet)."""
        for ptr in self._opened:
            self._lib.dp_ipc_close(ptr)
        self._opened = []
        if collective and dist.is_initialized():
            dist.barrier(group=self._group)
        if self._mine:
            self._lib.dp_p2p_free(self._mine)
            self._mine = self._C.c_void_p()

    def allreduce_(self, t):
        assert t.is_contiguous() and t.dtype == torch.float32 and t.numel() == self.n and t.device == self.device
        self.epoch += 1
        C = self._C
        self._libmod.check(self._lib.dp_allreduce_p2p(C.c_void_p(t.data_ptr()), C.c_int64(self.n), self._windows, self.rank,
                                                      self.world, C.c_uint32(self.epoch), C.c_void_p(self.err.data_ptr()),
                                                      C.c_int64(self.fail_slot), self._libmod.stream_ptr()), "dp_allreduce_p2p")
        return t

    def check(self):
        """Host sync: True if every all-reduce since the last check saw all ranks."""
        bad = int(self.err.item())
        if bad:
            self.err.zero_()
        return bad == 0

    def close(self):
        """Collective: every rank closes the windows it opened, THEN (behind a barrier) frees its own."""
        if self._mine:
            torch.cuda.synchronize(self.device)
            self._release(collective=True)


def init_from_env(device_type="cuda"):
    """Read RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torch.distributed.run) and join the group."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FLY_SINGLE_GPU"):          # rehearse N ranks on one GPU (tests; needs FLY_DIST_BACKEND=gloo)
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        import datetime
        # a rank that never arrives (bad LOCAL_RANK, fewer GPUs than ranks) fails the rendezvous in 2 minutes, not 10
        timeout = datetime.timedelta(seconds=int(os.environ.get("FLY_DIST_TIMEOUT_S", "120")))
        backend = os.environ.get("FLY_DIST_BACKEND") or ("nccl" if device_type == "cuda" else "gloo")   # "nccl" is RCCL on ROCm
        if device_type == "cuda" and backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
    return rank, local_rank, world


def broadcast_parameters(module, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            if t.data.is_contiguous():
                dist.broadcast(t.data, src=src)
            else:                                   # a strided view (e.g. of a packed buffer)
                tmp = t.data.contiguous()
                dist.broadcast(tmp, src=src)
                t.data.copy_(tmp)


def broadcast_policy(agent, src=0):
    """Replicate rank `src`'s policy: ONE broadcast of the packed parameter buffer, then rebuild
    the fragment-ordered copies the kernels stream."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(agent.policy.P, src=src)
        agent.policy.refresh()
