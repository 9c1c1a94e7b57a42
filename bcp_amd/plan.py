"""Recorded launch plans: the host side of a network pass without the Python.

A forward (or backward) pass of one of the networks is a fixed sequence of C-ABI launches over tensors whose shapes never change
from step to step.  The first pass with a given (network, input shape, groups, mode, stream) runs the ordinary Python code once
while `Binding.call` records every launch -- the ctypes function and its argument list -- and every tensor the pass allocates is
kept alive by the plan, so all recorded device pointers stay valid.  Every later pass replays the list: one `fn(*args)` per launch,
no shape arithmetic, no `torch.empty`, no wrapper layers (bench: host enqueue of the LA step 5.3 -> see DESIGN.md section 8).

What varies between passes lives in device memory: the Dropout3d / Dropout seeds (the recorded draws are `bcp_bernoulli_dev`
launches reading the plan's seed table, which one `bcp_store_u64` launch refills from the network's seed stream before a replay)
and the input, which is copied into the plan's static input buffer (one device copy).  So a recorded pass is a constant launch
sequence -- and IS captured: with GRAPHS >= 1 (the default) and the caller on a real (non-null) stream, the first replay of a FORWARD pass runs
under HIP stream capture and every later one is ONE `hipGraphLaunch` (`bcp_graph_launch`); the backward pass can be captured too (event fork /
join onto the weight-gradient side stream included; GRAPHS = 2, not the default: see GRAPHS below).  Stream-ordering calls (`wait_stream`) and data-parallel bucket hooks are recorded as Python callables in
place.  Weight packing is NOT recorded: it depends on the weights' version and runs eagerly before a replay.

The reference has no counterpart (its host path is PyTorch's eager dispatch); this is the "launch plan" of VERDICT r01 item 4.
"""
import contextlib

import torch

ENABLED = True          # module switch (tests compare a replayed step with an eager one)
# Capture a plan's second pass in a HIP graph when the caller runs on a real (non-null) stream.  0: never; 1: forward passes (the default
# again since round 5); 2: also the backward pass.
# History.  Round 4 switched capture off: with the teacher's forward graph on its side stream BESIDE the student's work, the teacher's logits
# deviated from the eager path's in 24 of 150 runs of a small ACDC step -- and the driver's box then showed the same signature WITHOUT graphs
# (per-launch replays, GPUTEST_r04).  Round 5 found the cause outside this module: ONE instruction of the round-4 build of k_bilinear2x_fwd --
# v_pk_mul_f32 with the halves of its second source crossed -- returns +-0 in lanes 48..63 whenever another kernel's wave on the same CU issues
# 16-bit MFMAs, i.e. whenever the other network's bf16-pipe convs ran beside the upsample (DESIGN.md section 4.0,
# tools/probe/pkmul_mfma_repro.hip) -- a graph merely packed the two streams' kernels more tightly.  With the rebuilt kernel: graphs = 1 and
# graphs = 2 beside the teacher stream under the load generator 0 of 150 runs each, LA 0 of 80 (tools/probe/replay_stress.py,
# gpurun_out/r05_s11), per-launch replays 0 of 450.
# What capture buys (round 5, gpurun_out/r05_s13/graphs_ab.txt, two interleaved pairs): nothing on the GPU -- LA 5.47 / 5.47 ms, ACDC 3.45 /
# 3.44, pancreas 5.01 / 5.01 -- and 0.6-1.1 ms of host time per step (LA 2.7 -> 1.6 ms of enqueue, ACDC 1.9 -> 1.4, pancreas 2.5 -> 1.8): room
# for the input pipeline and the Python around the step.  The backward pass stays a per-launch replay: inside a graph the weight-gradient
# branch no longer overlaps the dgrad -> norm chain the way the side stream does (round 4: LA 5.81-5.92 vs 5.34-5.38 ms).
GRAPHS = 1
PROFILE = None          # bench.py's per-op table: an Ops whose profile_begin() is live -- replays then go launch by launch through
                        # bcp_replay_run_timed with HIP events around every recorded op (LaunchPlan.spans), graphs bypassed; the host still
                        # runs ahead of the GPU, so the brackets hold the ops as they run inside the step
PROFILE_ONLY = None      # None: every recorded op gets its pair of events; a set of (op name, first tensor's shape): only those -- every event is a
                        # marker packet between two kernels (the fully bracketed LA step runs 6.7 instead of 5.4 ms and a 55 us conv reads
                        # 72 us), so bench.py ranks the ops with everything bracketed and then times the dominant ones ALONE in an otherwise
                        # undisturbed step
C_REPLAY = True         # replay runs of recorded launches from C (bcp_replay_run: one foreign call per run); False: one ctypes call per launch
_EPOCH = [0]            # bumped when library options change: every plan recorded before is dropped


def invalidate_all():
    _EPOCH[0] += 1


def epoch():
    return _EPOCH[0]


class LaunchPlan:
    __slots__ = ("entries", "keep", "n_seeds", "seed_dev", "static_in", "result", "ticks", "n_calls", "busy", "graph", "graph_state",
                 "capturable", "forks", "_owner", "segs", "_handles", "spans", "_seg_of", "_timed")

    MAX_SEEDS = 16

    def __init__(self):
        self.entries = []        # [callable, [args...], entry-point name | None for a Python callable]
        self.keep = []           # every tensor allocated while recording (pointers inside `entries` refer to them)
        self.n_seeds = 0         # Dropout3d / Dropout draws of the pass; their seeds live in `seed_dev` (device memory) so that the
        self.seed_dev = None     #   recorded launches -- and a graph captured from them -- never change: bcp_bernoulli_dev
        self.static_in = None
        self.result = None
        self.ticks = 0
        self.n_calls = 0
        self.busy = False        # a forward plan whose saved activations a pending backward still needs
        self.graph = None        # hipGraphExec_t of the captured pass (GRAPHS)
        self.graph_state = 0     # 0: capture not tried yet, 1: captured, -1: stays a per-launch replay
        self.capturable = True
        self.forks = False       # the pass forks onto a side stream (backward: weight gradients)
        self._owner = None
        self.segs = None         # compiled form of `entries` (compile)
        self._handles = []
        self.spans = []          # (op name, shapes, ints, operands carrying |max|, first entry, one past the last entry) of every profiled op of
                                 # the pass, noted while recording (hip_ops._profiled): what bench.py's per-op events bracket in a replay
        self._seg_of = None      # entry index -> (segment index, index inside the segment's C handle) (compile)
        self._timed = None

    # called by Binding.call while recording
    def add_call(self, name, fn, args):
        self.entries.append([fn, list(args), name])
        self.n_calls += 1
        if name == "bcp_stream_wait_stream":
            self.forks = True

    def add_py(self, fn, *args, capturable=True):
        """a Python callable in launch order; capturable=False: something a graph must not swallow (the data-parallel bucket
        hook: it changes per step).  Stream ordering is NOT one of these: it is a recorded bcp_stream_wait_stream launch."""
        self.entries.append([fn, list(args), None])
        if not capturable:
            self.capturable = False

    def seed_slot(self, device):
        """device address of the next dropout seed of this pass (recording)"""
        if self.seed_dev is None:
            self.seed_dev = torch.zeros(self.MAX_SEEDS, dtype=torch.int64, device=device)
        assert self.n_seeds < self.MAX_SEEDS, "more dropout draws per pass than a plan holds seeds for"
        self.n_seeds += 1
        return self.seed_dev.data_ptr() + 8 * (self.n_seeds - 1)

    def compile(self, b):
        """runs of consecutive C-ABI launches -> one bcp_replay_run each (csrc/replay.hip); Python callables stay in between"""
        import ctypes as C
        from . import _lib
        segs, cur = [], None
        seg_of, n_in = {}, 0
        run = b._fns["bcp_replay_run"][0]
        for ei, (fn, args, name) in enumerate(self.entries):
            if name is None or not C_REPLAY:
                cur = None
                segs.append((fn, args))
                continue
            if cur is None:
                cur = C.c_void_p()
                if b.cdll.bcp_replay_create(C.byref(cur)):
                    raise _lib.BcpError(b.last_error())
                self._handles.append(cur)
                self._owner = b
                segs.append((run, [cur]))
                n_in = 0
            shape = _lib.shape_of(name)
            if b.cdll.bcp_replay_add(cur, C.cast(fn, C.c_void_p), shape.encode(), _lib.pack_slots(shape, args), len(args)):
                raise _lib.BcpError(f"{name}: {b.last_error()}")
            seg_of[ei] = (len(segs) - 1, n_in)
            n_in += 1
        self.segs = segs
        self._seg_of = seg_of

    def run_entries(self, check):
        for fn, args in self.segs:
            rc = fn(*args)
            if rc:
                check(rc)

    def run_entries_timed(self, ops, check):
        """one pass, launch by launch from C, with a HIP event in front of and behind every recorded op (self.spans) on the stream the op's
        first / last launch uses; the (op, shapes, ints, events) records go to ops._prof like the eager wrappers' (hip_ops._profiled)"""
        import ctypes as C
        b = ops.b
        if self._timed is None:
            # per C segment: entry count and the stream of every entry (the last argument of every launch entry point is its stream;
            # bcp_stream_wait_stream has none and is never the first / last launch of an op)
            per = {}
            for ei, (si, li) in self._seg_of.items():
                per.setdefault(si, []).append((li, self.entries[ei][1][-1] if self.entries[ei][2] != "bcp_stream_wait_stream" else None))
            self._timed = {si: [st for _, st in sorted(v)] for si, v in per.items()}
        arrays = {}
        for si, streams in self._timed.items():
            n = len(streams)
            arrays[si] = ((C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)(*[C.c_void_p(s or 0) for s in streams]))
        for name, shapes, ints, namax, n0, n1 in self.spans:
            first, last = self._seg_of.get(n0), self._seg_of.get(n1 - 1)
            if first is None or last is None or n1 <= n0:
                continue                     # (an op that starts or ends with a Python callable: not bracketed)
            if PROFILE_ONLY is not None and (name, shapes[0] if shapes else ()) not in PROFILE_ONLY:
                continue
            if arrays[first[0]][0][first[1]] or arrays[last[0]][1][last[1]]:
                continue                     # (a wrapper that only forwards to another profiled op, e.g. a dgrad served by conv3_fwd: the inner
                                             #  op has the slot -- one event per launch boundary, and no launch is counted twice)
            e0, e1 = ops._prof_event(), ops._prof_event()
            arrays[first[0]][0][first[1]] = e0
            arrays[last[0]][1][last[1]] = e1
            ops._prof.append((name, shapes, ints, e0, e1, namax))
        timed = b._fns["bcp_replay_run_timed"][0]
        for si, (fn, args) in enumerate(self.segs):
            if si in arrays:
                rc = timed(args[0], arrays[si][0], arrays[si][1], arrays[si][2])
            else:
                rc = fn(*args)
            if rc:
                check(rc)

    def replay(self, ops, seeds, like):
        """one pass: refresh the seeds, then either the captured graph (one call) or the recorded launches one by one"""
        b = ops.b
        if self.segs is None:
            self.compile(b)
        if self.n_seeds:
            ops.store_u64(self.seed_dev, seeds, like)
        if PROFILE is not None and PROFILE._prof is not None and C_REPLAY:
            self.run_entries_timed(PROFILE, b.check_replayed)
            return
        if self.graph is not None:
            b.call("bcp_graph_launch", self.graph, ops.stream(like))
            return
        if self.graph_state == 0 and GRAPHS and self.capturable and like.is_cuda and (GRAPHS >= 2 or not self.forks):
            self.graph_state = -1
            stream = ops.stream(like)
            if stream:                                   # the null stream cannot be captured: run the step on a real stream to get graphs
                self.graph = _capture(self, b, stream)
                if self.graph is not None:
                    self.graph_state = 1
                    self._owner = b
                    b.call("bcp_graph_launch", self.graph, stream)
                    return
        self.run_entries(b.check_replayed)

    def __del__(self):
        g, b = getattr(self, "graph", None), getattr(self, "_owner", None)
        try:
            if g is not None and b is not None:
                b.cdll.bcp_graph_destroy(g)
            for h in getattr(self, "_handles", ()):
                b.cdll.bcp_replay_destroy(h)
        except Exception:
            pass


def _capture(pl, b, stream):
    """replay the recorded launches under stream capture -> hipGraphExec_t, or None (the plan stays a per-launch replay)"""
    import ctypes as C
    if b.cdll.bcp_graph_begin_capture(stream):
        return None
    err = None
    try:
        pl.run_entries(b.check_replayed)
    except Exception as e:          # the capture must be closed whatever happened
        err = e
    handle = C.c_void_p()
    rc = b.cdll.bcp_graph_end_capture(stream, C.byref(handle))
    if err is not None:
        if not rc and handle.value:
            b.cdll.bcp_graph_destroy(handle)
        raise err
    if rc or not handle.value:
        return None
    return handle


@contextlib.contextmanager
def recording(ops, plan):
    """run a pass eagerly while collecting its launches and allocations into `plan`"""
    b = ops.b
    real_empty, real_empty_like = torch.empty, torch.empty_like

    def empty(*a, **k):
        t = real_empty(*a, **k)
        plan.keep.append(t)
        return t

    def empty_like(*a, **k):
        t = real_empty_like(*a, **k)
        plan.keep.append(t)
        return t

    prev = b._rec
    b._rec = plan
    ops._rec_plan = plan
    torch.empty, torch.empty_like = empty, empty_like
    ok = False
    try:
        yield plan
        ok = True
    finally:
        torch.empty, torch.empty_like = real_empty, real_empty_like
        ops._rec_plan = None
        b._rec = prev
    if ok:
        _assert_no_dangling_pointers(plan)


def _assert_no_dangling_pointers(plan):
    """Every device pointer a recorded launch carries must still be allocated when the recording ends.  The plan keeps what the pass
    allocated through torch.empty / torch.empty_like alive; a tensor made any other way inside a recorded pass (torch.zeros, clone,
    .contiguous(), .to()) would be freed after the pass and the replay -- or the HIP graph -- would silently read or write memory the
    caching allocator has handed to someone else.  Checked against the allocator's own block table (CUDA tensors only)."""
    import ctypes as C
    if not (torch.cuda.is_available() and torch.cuda.is_initialized()):
        return
    ptrs = []
    for fn, args, name in plan.entries:
        at = getattr(fn, "argtypes", None)
        if name is None or not at:
            continue
        for a, t in zip(args, at):
            if t is C.c_void_p and isinstance(a, int) and a:
                ptrs.append((a, name))
    if not ptrs:
        return
    free = []                                   # [lo, hi) of every block the allocator holds but has not handed out
    for seg in torch.cuda.memory_snapshot():
        addr = seg["address"]
        for blk in seg["blocks"]:
            if blk["state"] != "active_allocated":
                free.append((addr, addr + blk["size"]))
            addr += blk["size"]
    if not free:
        return
    free.sort()
    import bisect
    los = [f[0] for f in free]
    for a, name in ptrs:
        i = bisect.bisect_right(los, a) - 1
        if i >= 0 and free[i][0] <= a < free[i][1]:
            raise RuntimeError(f"launch plan: {name} was recorded with a device pointer ({a:#x}) into memory that was freed before the "
                               "recording ended -- a tensor allocated inside the recorded pass by something other than torch.empty / "
                               "torch.empty_like (the plan cannot keep it alive)")


@contextlib.contextmanager
def suspended(ops):
    """inside a recording: run something eagerly WITHOUT recording it (weight packing)"""
    b = ops.b
    prev = b._rec
    b._rec = None
    try:
        yield
    finally:
        b._rec = prev


def use_real_stream(device):
    """make a non-null stream the current one on `device` (no-op when it already is): the null stream cannot be captured, so the
    training scripts and bench.py call this once before their loop to get graph replays"""
    if torch.cuda.is_available() and torch.device(device).type == "cuda":
        if torch.cuda.current_stream(device).cuda_stream == 0:
            s = torch.cuda.Stream(device=device)
            s.wait_stream(torch.cuda.current_stream(device))
            torch.cuda.set_stream(s)
