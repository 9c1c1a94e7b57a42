"""Data parallelism for the BCP step: one process per GPU, every rank holds a full
student + teacher replica and its own micro-batch / box / dropout stream; the ONLY exchange is the
all-reduce (sum) of the flat fp32 gradient buffer per step over RCCL -- a few >= 8 MB buckets in reverse layer
order, started while the backward pass is still running -- scaled by 1/world inside the fused SGD launch.
BatchNorm statistics stay rank-local (DDP convention; the reference's only multi-GPU code, nn.DataParallel in
pancreas/dataloaders.py:14, also normalises per replica).  Teachers stay identical because the students
do.  N ranks == N sequential micro-batches with averaged gradients (tests/test_dp_gloo.py).

Transports (BCP_DP_BACKEND, default "rccl" on a GPU):
  * "rccl": the library's own communicator -- bcp_comm_init_rank / bcp_allreduce_f32 (include/bcp_hip.h, csrc/comm.hip:
    ncclAllReduce on a stream we own); the 128-byte unique id travels from rank 0 through the launch's c10d store at
    MASTER_ADDR:MASTER_PORT (multi-node capable; _exchange_id), torch.distributed's process groups are not used.
  * "nccl" / "gloo": torch.distributed process groups ("nccl" IS RCCL on ROCm; "gloo" runs the CPU tests).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import time

import torch


_ID_SEQ = [0]      # communicators built by this process so far (the ranks of a launch build them in the same order)


def _exchange_id(ident, world, rank):
    """rank 0's 128-byte RCCL unique id -> every rank, over the launch's own rendezvous: the c10d store at MASTER_ADDR:MASTER_PORT
    exactly as torch.distributed's env:// method finds it (the elastic agent's store under torchrun, else a TCPStore served by
    rank 0) -- works across nodes, needs no shared filesystem, and a key is written once per (launch, communicator), so a crashed
    earlier launch cannot leave a stale id behind (round-2 ADVICE: the id used to travel through a file under /tmp).
    BCP_DP_ID_DIR=<dir> selects the file transport instead (single node, no TCP): rank 0 unlinks any old file first and the file
    carries a per-launch nonce (BCP_DP_LAUNCH_ID or the launcher's pid) in its name."""
    _ID_SEQ[0] += 1
    if world == 1:
        return ident                                 # a one-rank communicator (tests): nothing to exchange
    d = os.environ.get("BCP_DP_ID_DIR")
    if not d and not (os.environ.get("MASTER_ADDR") and os.environ.get("MASTER_PORT")):
        # launched without torchrun's rendezvous variables (only RANK / WORLD_SIZE set): the store transport cannot work -- say so
        # instead of failing inside the TCP rendezvous
        raise RuntimeError("bcp_amd.dp: the RCCL unique id travels through the torch.distributed rendezvous store and needs MASTER_ADDR / "
                           "MASTER_PORT (python -m torch.distributed.run sets them); on a single node without them set BCP_DP_ID_DIR=<a "
                           "directory all ranks see> to exchange the id through a file instead")
    if not d:
        from datetime import timedelta
        from torch.distributed import rendezvous
        store, _, _ = next(iter(rendezvous("env://", rank=rank, world_size=world, timeout=timedelta(seconds=180))))
        key = f"bcp_rccl_id/{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}/{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}/{_ID_SEQ[0]}"
        if rank == 0:
            store.set(key, ident)
        out = bytes(store.get(key))                  # blocks until rank 0 has set it (store timeout: 180 s)
        assert len(out) == 128, "malformed RCCL unique id in the rendezvous store"
        _STORES.append(store)                        # rank 0 may be serving it: keep it alive for the life of the process
        return out
    tag = (f"{os.environ.get('MASTER_PORT', '29500')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{world}_"
           f"{os.environ.get('BCP_DP_LAUNCH_ID', os.getppid())}_{_ID_SEQ[0]}")
    path = os.path.join(d, f"bcp_rccl_id_{tag}")
    if rank == 0:
        try:
            os.remove(path)                          # nothing older may be mistaken for this launch's id
        except OSError:
            pass
        tmp = path + f".{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(ident)
        os.replace(tmp, path)                        # atomic: readers never see a half-written id
        _NONCE[0] = hashlib.sha1(ident).hexdigest()[:12]
        return ident
    t_launch = time.time() - float(os.environ.get("BCP_DP_ID_MAX_AGE", "120"))
    t0 = time.time()
    while True:
        try:
            if os.path.getmtime(path) >= t_launch and os.path.getsize(path) == 128:
                out = open(path, "rb").read()
                _NONCE[0] = hashlib.sha1(out).hexdigest()[:12]
                return out
        except OSError:
            pass
        if time.time() - t0 > 120:
            raise RuntimeError(f"rank {rank}: no RCCL unique id at {path} after 120 s")
        time.sleep(0.01)


_STORES = []


_NONCE = [None]   # file transport: a per-launch nonce = digest of the exchanged unique id (rank 0 makes a new id per launch): flag files named by it
                  # can never be mistaken for an earlier launch's; a rank that never got the id writes a ".noid" flag instead


_AGREE_SEQ = [0]   # agreements held by this process so far (one per communicator attempt, the same count on every rank)


def _agree(ok, world, rank, timeout_s=None):
    """True iff EVERY rank reports ok.  Runs between the rank-local half of building a communicator (dlopen, id exchange) and the
    collective half (ncclCommInitRank): a failure that hits one rank only -- an id file that never appeared, a rendezvous time-out --
    used to send that rank into the torch.distributed fall-back while the others sat in ncclCommInitRank for ever.
    Flags travel the way the id did: keys in the launch's c10d store, or files next to the id file (BCP_DP_ID_DIR).  A flag that
    does not arrive within the time-out counts as a failure, so every rank takes the same way out."""
    _AGREE_SEQ[0] += 1
    if world == 1:
        return bool(ok)
    seq = _AGREE_SEQ[0]      # (not _ID_SEQ: a rank that failed before its id exchange has not counted that one)
    timeout_s = float(os.environ.get("BCP_DP_AGREE_TIMEOUT", "180")) if timeout_s is None else timeout_s
    d = os.environ.get("BCP_DP_ID_DIR")
    if d:
        tag = (f"{os.environ.get('MASTER_PORT', '29500')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{world}_"
               f"{os.environ.get('BCP_DP_LAUNCH_ID', os.getppid())}_{seq}")
        nonce = _NONCE[0]
        flag = lambda n, r: os.path.join(d, f"bcp_rccl_ok_{tag}.{n}.{r}")
        mine = flag(nonce or "noid", rank)
        if nonce:
            try:
                os.remove(flag("noid", rank))        # an earlier launch of this tag that failed here: its flag must not speak for this one
            except OSError:
                pass
        tmp = mine + f".{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(b"1" if ok else b"0")
        os.replace(tmp, mine)                        # atomic
        # a peer's flag counts when it carries THIS launch's nonce; a ".noid" flag (the peer failed before it had the id) only when it is
        # younger than the launch can be, as _exchange_id treats the id file.  Flags are left behind (a slower peer may still have to read
        # them when this rank is done): nonce'd ones never match again, ".noid" ones age out / are unlinked by their owner's next launch
        t0, all_ok = time.time(), bool(ok)
        t_launch = t0 - float(os.environ.get("BCP_DP_ID_MAX_AGE", "120"))
        for r in range(world):
            cands = ([flag(nonce, r)] if nonce else []) + [flag("noid", r)]
            if not nonce:                            # this rank failed early and reports False whatever it reads; it only waits for its peers
                import glob
            while True:
                v = b""
                for path in (cands if nonce else glob.glob(flag("*", r))):
                    try:
                        if path.endswith(f".noid.{r}") and os.path.getmtime(path) < t_launch:
                            continue
                        v = open(path, "rb").read()
                    except OSError:
                        continue
                    if v in (b"0", b"1"):
                        break
                if v in (b"0", b"1"):
                    all_ok = all_ok and v == b"1"
                    break
                if time.time() - t0 > timeout_s:
                    return False
                time.sleep(0.01)
        return all_ok
    try:
        store = _STORES[-1] if _STORES else None
        if store is None:
            from datetime import timedelta
            from torch.distributed import rendezvous
            store, _, _ = next(iter(rendezvous("env://", rank=rank, world_size=world, timeout=timedelta(seconds=timeout_s))))
            _STORES.append(store)
        base = f"bcp_rccl_ok/{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}/{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}/{seq}"
        store.set(f"{base}/{rank}", b"1" if ok else b"0")
        all_ok = bool(ok)
        for r in range(world):
            all_ok = all_ok and bytes(store.get(f"{base}/{r}")) == b"1"      # blocks until rank r has reported (store time-out: a failure)
        return all_ok
    except Exception:
        return False


class _RcclAbi:
    """RCCL through the C ABI: communicator + one stream for the collectives"""

    def __init__(self, world, rank, local_rank):
        """the RANK-LOCAL half: library, device, the unique id (rank 0 makes it, everyone fetches it).  init() is the collective half;
        DataParallel calls _agree() between the two"""
        from . import _lib
        _NONCE[0] = None                             # set again by this communicator's id exchange
        self.b = _lib.product()
        if not self.b.call("bcp_comm_available"):
            raise RuntimeError("librccl.so could not be loaded")
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self._ident = (C.c_char * 128)()
        if rank == 0:
            self.b.call("bcp_comm_unique_id", C.cast(self._ident, C.c_void_p))
        C.memmove(self._ident, _exchange_id(bytes(self._ident) if rank == 0 else None, world, rank), 128)
        self.world, self.rank = world, rank
        self.comm = None

    def init(self):
        """ncclCommInitRank: COLLECTIVE -- call only when every rank got through __init__ (_agree); a failure in here is raised, there
        is no common way out of a half-built communicator"""
        comm = C.c_void_p()
        self.b.call("bcp_comm_init_rank", C.byref(comm), self.world, self.rank, C.cast(self._ident, C.c_void_p))
        self.comm = comm
        self.stream = torch.cuda.Stream(device=self.dev)
        self.barrier()
        return self

    def count(self):
        n = C.c_int(0)
        self.b.call("bcp_comm_count", self.comm, C.byref(n))
        return int(n.value)

    def all_reduce_async(self, t):
        """sum `t` in place over the ranks on the communicator's stream, ordered after the current stream; returns an event"""
        cur = torch.cuda.current_stream(t.device)
        self.stream.wait_stream(cur)
        t.record_stream(self.stream)
        self.b.call("bcp_allreduce_f32", self.comm, t.data_ptr(), t.numel(), self.stream.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return ev

    def all_reduce(self, t):
        self.all_reduce_async(t).wait()              # the CURRENT STREAM waits (no host sync)

    def barrier(self):
        t = torch.ones(1, dtype=torch.float32, device=self.dev)
        self.all_reduce(t)
        torch.cuda.current_stream(self.dev).synchronize()

    def broadcast(self, t, src=0):
        f = t if t.dtype == torch.float32 else t.to(torch.float32)
        if self.rank != src:
            f.zero_()
        self.all_reduce(f.view(-1))
        if f is not t:
            t.copy_(f)

    def max_f64(self, v):
        t = torch.zeros(self.world, dtype=torch.float32, device=self.dev)
        t[self.rank] = float(v)
        self.all_reduce(t)
        return float(t.max().item())

    def shutdown(self):
        torch.cuda.synchronize(self.dev)
        self.b.call("bcp_comm_destroy", self.comm)
        self.comm = None


class DataParallel:
    def __init__(self, backend=None, force=False):
        """force: build the communicator even for WORLD_SIZE=1 (a one-rank RCCL communicator: the collective is the identity,
        but the stream ordering of the bucketed exchange is the real one -- tests/test_gpu_scripts.py)"""
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.enabled = self.world > 1 or force
        self.bucket_bytes = int(float(os.environ.get("BCP_DP_BUCKET_MB", "8")) * (1 << 20))
        self.n_collectives = 0
        self._works, self._armed, self._hi = [], False, 0
        self._exposed, self._buckets = [], []      # (event before, event after) of allreduce_grads' final wait; bucket sizes of the last step
        self.abi = None
        self.backend = None
        if not self.enabled:
            return
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("BCP_DP_BACKEND") or ("rccl" if torch.cuda.is_available() else "gloo")
        self.backend = backend
        if backend == "rccl":
            from . import _lib
            import sys
            # Two halves with an agreement in between: the rank-local half (dlopen of librccl, the id exchange) may fail on ONE rank only;
            # every rank then learns it (_agree) and all of them take the same way out.  The collective half (ncclCommInitRank) runs only
            # when every rank got that far, and a failure inside it is raised.
            abi, err = None, None
            try:
                if not _lib.product().call("bcp_comm_available"):
                    raise RuntimeError("librccl.so not loadable through the C ABI")
                abi = _RcclAbi(self.world, self.rank, self.local_rank)
            except Exception as e:
                err = e
            if _agree(err is None, self.world, self.rank):
                self.abi = abi.init()
                return
            if os.environ.get("BCP_DP_BACKEND") == "rccl":      # asked for by name: no substitute
                raise err if err is not None else RuntimeError("bcp_amd.dp: another rank could not prepare its RCCL communicator")
            why = f"{type(err).__name__}: {err}" if err is not None else "another rank failed before ncclCommInitRank"
            print(f"[bcp_amd.dp] rank {self.rank}: RCCL communicator through the C ABI not built ({why}) -- every rank falls back to "
                  "torch.distributed 'nccl' (the same RCCL, torch's process group)", file=sys.stderr, flush=True)
            backend = self.backend = "nccl"
        import torch.distributed as dist
        if not dist.is_initialized():
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)

    def ranks_seen(self) -> int:
        """how many ranks the communicator really spans (bench.py reports it so that a multi-GPU number can be checked)"""
        if not self.enabled:
            return 1
        if self.abi is not None:
            t = torch.ones(1, dtype=torch.float32, device=self.abi.dev)
            self.abi.all_reduce(t)
            n = int(round(float(t.item())))
            assert n == self.abi.count(), "all-reduce of ones and ncclCommCount disagree"
            return n
        import torch.distributed as dist
        t = torch.ones(1, dtype=torch.float32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t)
        return int(round(float(t.item())))

    def broadcast_params(self, model):
        """rank 0's weights (and BN buffers) everywhere, once at start"""
        if not self.enabled:
            return
        flat = model.flat_params()
        if self.abi is not None:
            self.abi.broadcast(flat)
            for b in model.buffers():
                self.abi.broadcast(b)
        else:
            import torch.distributed as dist
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
        the collective is ordered after the weight-gradient side stream and runs on the communicator's own stream
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

    def _all_reduce_async(self, g):
        if self.abi is not None:
            return self.abi.all_reduce_async(g)
        import torch.distributed as dist
        return dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True)

    def _launch(self, model, lo, hi, like, overlapped):
        g = model.flat_grads()[lo:hi]
        self._buckets.append(((hi - lo) * 4, bool(overlapped)))
        side = model._side_streams.get(like.device) if (overlapped and like.is_cuda and model.overlap_wgrad) else None
        if side is not None:
            # weight gradients of the layers in this bucket were enqueued on the side stream, norm / bias gradients on the main
            # stream: order the collective after both (the side stream's later weight gradients need later dy anyway)
            side.wait_stream(torch.cuda.current_stream(like.device))
            with torch.cuda.stream(side):
                self._works.append(self._all_reduce_async(g))
        else:
            self._works.append(self._all_reduce_async(g))
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
            ev0 = ev1 = None
            if g.is_cuda and self._measure:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            for w in self._works:
                w.wait()                 # rccl / nccl: the current stream waits for the collective; gloo: the host does
            if ev0 is not None:
                ev1.record()
                self._exposed.append((ev0, ev1))
            self._works = []
            self._last_buckets, self._buckets = self._buckets, []
        else:
            # not armed (ungrouped steps: several backward passes): ONE collective after the last of them, nothing hidden -- the whole
            # exchange is exposed, and is measured as such (this branch used to leave exposed_ms_per_step at None)
            ev0 = ev1 = None
            if g.is_cuda and self._measure:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            if self.abi is not None:
                self.abi.all_reduce(g)
            else:
                import torch.distributed as dist
                dist.all_reduce(g, op=dist.ReduceOp.SUM)
            if ev0 is not None:
                ev1.record()
                self._exposed.append((ev0, ev1))
            self.n_collectives += 1
        if optimizer is not None and hasattr(optimizer, "grad_scale"):
            optimizer.grad_scale = 1.0 / self.world
        else:
            g.mul_(1.0 / self.world)

    # ---- measurement (bench.py --gpus N): how long the optimiser's stream really waited for the exchange
    _measure = False
    _last_buckets = ()

    def reset_exposed(self):
        self._exposed, self._measure = [], self.enabled

    def exposed_ms_per_step(self, steps):
        """stream time between `the backward pass is enqueued` and `every bucket has arrived`, averaged over the timed steps:
        the part of the gradient exchange the backward pass did not hide (0 when not data-parallel)"""
        self._measure = False
        if not self._exposed:
            return 0.0 if not self.enabled else None
        torch.cuda.synchronize()
        tot = sum(a.elapsed_time(b) for a, b in self._exposed)
        self._exposed = []
        return round(tot / max(steps, 1), 4)

    def bucket_report(self):
        return {"bucket_mb_min": self.bucket_bytes / (1 << 20), "transport": self.backend,
                "buckets_bytes": [b for b, _ in self._last_buckets], "overlapped_with_backward": [o for _, o in self._last_buckets]}

    def barrier(self):
        if not self.enabled:
            return
        if self.abi is not None:
            self.abi.barrier()
        else:
            import torch.distributed as dist
            dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if not self.enabled:
            return value
        if self.abi is not None:
            return self.abi.max_f64(value)
        import torch.distributed as dist
        t = torch.tensor([value], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def shutdown(self):
        if not self.enabled:
            return
        if self.abi is not None:
            self.abi.shutdown()
            self.abi = None
            return
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
