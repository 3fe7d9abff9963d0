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
        backend = os.environ.get("FLY_DIST_BACKEND") or ("nccl" if device_type == "cuda" else "gloo")   # "nccl" is RCCL on ROCm
        if device_type == "cuda" and backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
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
