"""Data parallelism for the BCP step: one process per GPU, every rank holds a full
student + teacher replica and its own micro-batch / box / dropout stream; the ONLY exchange is the
all-reduce (sum) of the flat fp32 gradient buffer per step over RCCL (torch.distributed backend "nccl" on
ROCm; "gloo" in the CPU tests) -- a few >= 8 MB buckets in reverse layer order, started while the backward pass
is still running -- scaled by 1/world inside the fused SGD launch.  BatchNorm statistics stay
rank-local (DDP convention; the reference's only multi-GPU code, nn.DataParallel in
pancreas/dataloaders.py:14, also normalises per replica).  Teachers stay identical because the students
do.  N ranks == N sequential micro-batches with averaged gradients (tests/test_dp_gloo.py)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, backend=None, force=False):
        """force: build the process group even for WORLD_SIZE=1 (a one-rank RCCL communicator: the collective is the identity,
        but the stream ordering of the bucketed exchange is the real one -- tests/test_gpu_scripts.py)"""
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.enabled = self.world > 1 or force
        self.bucket_bytes = int(float(os.environ.get("BCP_DP_BUCKET_MB", "8")) * (1 << 20))
        self.n_collectives = 0
        self._works, self._armed, self._hi = [], False, 0
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

    # ---- gradient exchange.  Default: buckets of >= bucket_mb MB, reduced in reverse layer order while the backward pass is
    # still running; BCP_DP_BUCKET_MB=0 (or a step with several backward calls) falls back to ONE all-reduce of
    # the whole flat buffer after the backward.  Either way every element is summed exactly once over the same ranks, so the
    # two modes give bit-identical gradients.
    def arm(self, model):
        """call right before the step's single loss.backward(): the network reports, layer by layer, how much of the flat
        gradient buffer is final, and suffixes of at least `bucket_mb` are all-reduced asynchronously from then on.  On a GPU
        the collective is ordered after the weight-gradient side stream and runs on the process group's own stream
        underneath the remaining dgrad / norm-backward kernels (xGMI ring time hidden behind the shallow, expensive levels:
        the deep levels hold 80 % of the V-Net's parameters and are differentiated first)."""
        self._works, self._armed = [], False
        if not self.enabled or self.bucket_bytes <= 0:
            return
        model._ensure_flat()
        self._hi = model._n_trainable_flat
        self._armed = True
        model._grad_bucket_hook = self._on_grads_final

    def _on_grads_final(self, model, lo, like):
        if lo >= self._hi or (self._hi - lo) * 4 < self.bucket_bytes:
            return          # (lo >= hi: a parameter registered out of layer order -- it simply joins a later bucket)
        self._launch(model, lo, self._hi, like, overlapped=True)
        self._hi = lo

    def _launch(self, model, lo, hi, like, overlapped):
        g = model.flat_grads()[lo:hi]
        side = model._side_streams.get(like.device) if (overlapped and like.is_cuda and model.overlap_wgrad) else None
        if side is not None:
            # weight gradients of the layers in this bucket were enqueued on the side stream, norm / bias gradients on the main
            # stream: order the collective after both (the side stream's later weight gradients need later dy anyway)
            side.wait_stream(torch.cuda.current_stream(like.device))
            with torch.cuda.stream(side):
                self._works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))
        else:
            self._works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))
        self.n_collectives += 1

    def allreduce_grads(self, model, optimizer=None):
        """after loss.backward(): sum the (rest of the) flat trainable-gradient buffer over ranks -- 37.8 MB for the V-Net in
        total -- and make the current stream wait for every bucket; the 1/world average is folded into the optimiser's
        grad_scale when given, else applied here."""
        if not self.enabled:
            return
        _, g = model.flat_trainable()
        if getattr(self, "_armed", False):
            model._grad_bucket_hook = None
            self._armed = False
            if self._hi > 0:
                self._launch(model, 0, self._hi, g, overlapped=False)
            for w in self._works:
                w.wait()                 # nccl: the current stream waits for the collective; gloo: the host does
            self._works = []
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            self.n_collectives += 1
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
