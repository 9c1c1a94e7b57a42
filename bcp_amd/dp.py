"""Data parallelism for the BCP step: one process per GPU, every rank holds a full
student + teacher replica and its own micro-batch / box / dropout stream; the ONLY exchange is one
all-reduce (sum) of the flat fp32 gradient buffer per step over RCCL (torch.distributed backend "nccl" on
ROCm; "gloo" in the CPU tests), scaled by 1/world inside the fused SGD launch.  BatchNorm statistics stay
rank-local (DDP convention; the reference's only multi-GPU code, nn.DataParallel in
pancreas/dataloaders.py:14, also normalises per replica).  Teachers stay identical because the students
do.  N ranks == N sequential micro-batches with averaged gradients (tests/test_dp_gloo.py)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, backend=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.enabled = self.world > 1
        if self.enabled and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)

    def broadcast_params(self, model):
        """rank 0's weights (and BN buffers) everywhere, once at start"""
        if not self.enabled:
            return
        flat = model.flat_params()
        dist.broadcast(flat, src=0)
        for b in model.buffers():
            dist.broadcast(b, src=0)
        model.bump()

    def allreduce_grads(self, model, optimizer=None):
        """sum the flat trainable-gradient buffer over ranks (one collective, 37.8 MB for the V-Net);
        the 1/world average is folded into the optimiser's grad_scale when given, else applied here."""
        if not self.enabled:
            return
        _, g = model.flat_trainable()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        if optimizer is not None and hasattr(optimizer, "grad_scale"):
            optimizer.grad_scale = 1.0 / self.world
        else:
            g.mul_(1.0 / self.world)

    def barrier(self):
        if self.enabled:
            dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if not self.enabled:
            return value
        t = torch.tensor([value], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def shutdown(self):
        if self.enabled and dist.is_initialized():
            dist.destroy_process_group()
