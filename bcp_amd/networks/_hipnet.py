"""Shared machinery of the HIP-backed networks: reference-exact parameter trees living in ONE flat fp32
buffer (so EMA / SGD / the DP all-reduce are single launches), packed-weight caches, and the
autograd bridge (one Function node per network call; the whole backward is scheduled by hand).
"""
from __future__ import annotations

import os
import weakref

import torch
import torch.nn as nn

from ..hip_ops import Ops


class ConvP(nn.Module):
    """parameter holder with the reference's `weight` / `bias` names (torch default init bounds)"""

    def __init__(self, wshape, fan_in, bias_n):
        super().__init__()
        import math
        bound = 1.0 / math.sqrt(max(fan_in, 1))
        self.weight = nn.Parameter(torch.empty(wshape).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(bias_n).uniform_(-bound, bound))


class BNP(nn.Module):
    """BatchNorm parameter / buffer holder (state_dict-compatible with nn.BatchNorm{2,3}d)"""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class Seq(nn.Module):
    """numbered children, like nn.Sequential's naming (indices may have gaps for ReLU / Dropout slots)"""

    def __init__(self, items):
        super().__init__()
        for idx, m in items:
            self.add_module(str(idx), m)


class Holder(nn.Module):
    pass


def contrastive_heads(owner: nn.Module, n_sel: int):
    """the reference's never-called heads (networks/VNet.py:250-278, networks/unet.py:216-247): kept so
    state_dict()/parameters() match key for key"""
    owner.projection_head = nn.Sequential(nn.Linear(16, 32), nn.BatchNorm1d(32), nn.ReLU(inplace=True), nn.Linear(32, 32))
    owner.prediction_head = nn.Sequential(nn.Linear(32, 32), nn.BatchNorm1d(32), nn.ReLU(inplace=True), nn.Linear(32, 32))
    for fam in ("contrastive_class_selector_", "contrastive_class_selector_memory"):
        for c in range(n_sel):
            sel = nn.Sequential(nn.Linear(32, 32), nn.BatchNorm1d(32), nn.LeakyReLU(negative_slope=0.2, inplace=True), nn.Linear(32, 1))
            owner.__setattr__(fam + str(c), sel)


class HipNet(nn.Module):
    """Base class: flat parameter / gradient storage + autograd bridge."""

    def __init__(self):
        super().__init__()
        self._ops = None
        self._flat = None
        self._flat_grad = None
        self._n_trainable_flat = 0   # prefix of the flat buffer that takes part in the optimiser (enc + dec)
        self._pack_cache = {}
        self._bump = 0               # incremented when the flat buffer is modified outside torch (fused SGD / EMA)
        self.drop_masks = None       # injectable dropout keep-masks (parity runs)
        self._drop_seed = None       # dropout stream of THIS instance: seeded lazily from torch's generator (see next_seed)

    # ------------------------------------------------------------------ ops / storage
    @property
    def ops(self) -> Ops:
        if self._ops is None:
            self._ops = Ops.product()
        return self._ops

    def set_ops(self, ops: Ops):
        """tests only: run this network on another binding (the host simulator)"""
        self._ops = ops
        return self

    def _ordered_params(self):
        # the module tree is fixed after construction: walk it once (nn.Module.parameters() re-walks every submodule and
        # hashes every tensor -- 36 % of the host time of a training step when called 7x per step)
        ps = self.__dict__.get("_plist")
        if ps is None:
            ps = list(self.parameters())
            object.__setattr__(self, "_plist", ps)
            object.__setattr__(self, "_first_fbuf", next((b for b in self.buffers() if b.is_floating_point()), None))
        return ps

    def flatten_(self):
        """(re)locate every parameter AND every float buffer (BN running stats) inside one contiguous fp32 buffer:
        [ optimiser parameters | other parameters (unused heads) | float buffers ], registration order inside
        each part.  One launch then covers SGD (first part), the parameter EMA (first two) or the state_dict EMA
        of the ACDC script (all three)."""
        ps = self._ordered_params()
        bufs = [b for b in self.buffers() if b.is_floating_point()]
        dev = ps[0].device
        items = list(ps) + bufs
        offs, off = [], 0
        for t in items:
            offs.append(off)
            off += (t.numel() + 3) // 4 * 4      # 16-B alignment of every view
        flat = torch.zeros(off, dtype=torch.float32, device=dev)
        for t, o in zip(items, offs):
            flat[o:o + t.numel()].copy_(t.data.reshape(-1))
            t.data = flat[o:o + t.numel()].view(t.shape)
        self._flat = flat
        self._offs = {id(t): o for t, o in zip(items, offs)}
        self._n_param_flat = offs[len(ps)] if bufs else off
        self._flat_grad = torch.zeros(self._n_param_flat, dtype=torch.float32, device=dev)
        self._grad_views = {id(p): self._flat_grad[o:o + p.numel()].view(p.shape) for p, o in zip(ps, offs)}
        self._opt_plist = [p for p in ps if id(p) in self._opt_param_ids]
        self._grads_stale = True
        n_train = 0
        for p, o in zip(ps, offs):
            if id(p) in self._opt_param_ids:
                n_train = o + (p.numel() + 3) // 4 * 4
        self._n_trainable_flat = n_train
        self._pack_cache.clear()
        return self

    def _ensure_flat(self):
        ps = self._ordered_params()
        if self._flat is None or self._flat.device != ps[0].device or ps[0].data_ptr() != self._flat.data_ptr() + 4 * self._offs.get(id(ps[0]), -1):
            self.flatten_()
        else:
            last = ps[-1]
            if last.data_ptr() != self._flat.data_ptr() + 4 * self._offs[id(last)]:
                self.flatten_()
            else:
                b = self._first_fbuf
                if b is not None and b.data_ptr() != self._flat.data_ptr() + 4 * self._offs.get(id(b), -1):
                    self.flatten_()

    def flat_params(self):
        """all parameters (optimiser part + unused heads): what `for p in model.parameters()` walks"""
        self._ensure_flat()
        return self._flat[:self._n_param_flat]

    def flat_state(self):
        """parameters + float buffers: what state_dict() holds, minus the int64 num_batches_tracked"""
        self._ensure_flat()
        return self._flat

    def flat_grads(self):
        self._ensure_flat()
        return self._flat_grad

    def flat_trainable(self):
        """(params, grads) prefix views covering exactly the encoder+decoder parameters"""
        self._ensure_flat()
        n = self._n_trainable_flat
        return self._flat[:n], self._flat_grad[:n]

    def grad_view(self, p):
        return self._grad_views[id(p)]

    def attach_grad(self, p):
        """-> (grad tensor, accumulate?)  first touch after zero_grad() re-attaches the flat view"""
        if p.grad is None:
            p.grad = self.grad_view(p)
            return p.grad, False
        return p.grad, True

    def bump(self):
        self._bump += 1
        object.__setattr__(self, "_wver", None)

    def begin_backward(self):
        """first backward after zero_grad(): clear the flat gradient buffer with ONE memset and re-attach every
        optimiser parameter's .grad view; from then on all kernels accumulate (+=)."""
        self._ensure_flat()
        ps = self._opt_plist
        if ps[0].grad is None or ps[0].grad.data_ptr() != self.grad_view(ps[0]).data_ptr():
            self._flat_grad.zero_()        # a foreign optimiser set the grads to None: re-attach the flat views
            for p in ps:
                p.grad = self.grad_view(p)
        elif self._grads_stale:
            self._flat_grad.zero_()        # FlatSGD / FlatAdam.zero_grad(): the views stay attached, one memset clears them
        self._grads_stale = False

    def clear_grads_now(self):
        """the memset a zero_grad() owes, issued NOW on the current stream instead of at the head of the next backward pass (the fused step
        functions call it in front of the student's forward, where it runs under the teacher's pass: 7 us + a launch gap off the step's
        critical path).  Only when the flat views are attached (otherwise begin_backward does everything, as before)"""
        self._ensure_flat()
        ps = self._opt_plist
        if self._grads_stale and ps[0].grad is not None and ps[0].grad.data_ptr() == self.grad_view(ps[0]).data_ptr():
            self._flat_grad.zero_()
            self._grads_stale = False

    def mark_grads_stale(self):
        """zero_grad() of the flat optimisers: O(1) on the host -- the next backward clears the flat buffer with one memset"""
        self._grads_stale = True

    # num_batches_tracked: every live BN layer is bumped once per training forward, so ONE host-side integer
    # stands for all of them and is written to the buffers only when a state_dict is taken (29 tiny device ops
    # per forward otherwise).  momentum is fixed (0.1), so the value never feeds back into the arithmetic.
    def _nbt_tick(self):
        self._nbt = getattr(self, "_nbt", 0) + 1
        self._nbt_dirty = True

    def flush_nbt(self):
        if getattr(self, "_nbt_dirty", False):
            for m in self.modules():
                if isinstance(m, BNP) and getattr(m, "_live", False):
                    m.num_batches_tracked.fill_(self._nbt)
            self._nbt_dirty = False

    def state_dict(self, *a, **k):
        self.flush_nbt()
        return super().state_dict(*a, **k)

    # ---- all 3x3(x3) conv weights are packed by ONE launch per (network, weight version)
    def register_conv3(self, key, weight, KD):
        """called by the network constructors for every MFMA conv layer; returns nothing, see conv3_packed()"""
        if not hasattr(self, "_c3"):
            self._c3 = []
        self._c3.append((key, weight, KD))

    def _build_pack_tables(self):
        import struct
        dev = self._flat.device
        sizes = []
        for _, w, KD in self._c3:
            Cout, Cin = w.shape[0], w.shape[1]
            sizes.append(self.ops.conv3_packed_floats(Cin, Cout, KD))      # == the dgrad pack's size (K16 / N16 swap roles)
        total = sum(sizes)
        self._pack_buf = torch.empty(2 * total, dtype=torch.float32, device=dev)
        self._pack_views = {}
        self._pack_items = []
        self._pack_need = {}          # (key, direction) -> sections the launches observed so far read (1 fp32 pack, 2 bf16 planes, 4 fp16 planes)
        self._pack_need_ver = 0
        self._pack_partial = None     # (need version, fwd descriptors, fwd + dgrad descriptors) with per-layer section masks
        self._pack_full = True        # the packs of _pack_state hold every section
        fwd_desc, all_desc, dg_desc = b"", b"", b""
        off = 0
        for (key, w, KD), n in zip(self._c3, sizes):
            Cout, Cin = w.shape[0], w.shape[1]
            K16, N16 = (Cin + 15) // 16 * 16, (Cout + 15) // 16 * 16
            wf, wd = self._pack_buf[off:off + n], self._pack_buf[total + off:total + off + n]
            self._pack_views[key] = (wf, wd)
            wf._bcp_pack, wd._bcp_pack = (weakref.ref(self), key, 0), (weakref.ref(self), key, 1)      # (Ops notes which section a launch read)
            self._pack_items.append((key, w.data_ptr(), wf.data_ptr(), wd.data_ptr(), Cout, Cin, KD * 9, K16, N16))
            d_f = struct.pack("<QQiiiiii", w.data_ptr(), wf.data_ptr(), Cout, Cin, KD * 9, K16, N16, 0)
            d_d = struct.pack("<QQiiiiii", w.data_ptr(), wd.data_ptr(), Cout, Cin, KD * 9, N16, K16, 1)
            d_t = struct.pack("<QQiiiiii", w.data_ptr(), wd.data_ptr(), Cout, Cin, KD * 9, N16, K16, 1 | 0x800)      # ... right behind its forward twin: |max| from there
            fwd_desc += d_f
            all_desc += d_f + d_t
            dg_desc += d_d
            off += n
        self._desc_fwd = torch.frombuffer(bytearray(fwd_desc), dtype=torch.uint8).to(dev)
        self._desc_all = torch.frombuffer(bytearray(all_desc), dtype=torch.uint8).to(dev)
        self._desc_dg = torch.frombuffer(bytearray(dg_desc), dtype=torch.uint8).to(dev)
        self._pack_state = None
        self._pack_ptr = self._flat.data_ptr()

    def _weights_version(self):
        """(bump counter, sum of tensor versions of every packed weight): recomputed once per network call
        (refresh_weights_version(), called by forward()), not once per layer"""
        v = self.__dict__.get("_wver")
        if v is None:
            v = self.refresh_weights_version()
        return v

    def refresh_weights_version(self):
        tot = 0
        for _, w, _ in getattr(self, "_c3", ()):
            tot += w._version
        for _, w, *_ in getattr(self, "_k2", ()):
            tot += w._version
        v = (self._bump, tot)
        object.__setattr__(self, "_wver", v)
        return v

    def conv3_packed(self, key, need_dgrad):
        """(wp_fwd, wp_dgrad) of layer `key`, repacking EVERY layer in one launch when any weight changed"""
        if getattr(self, "_pack_ptr", None) != self._flat.data_ptr():
            self._build_pack_tables()
        ver = self._weights_version()
        st = self._pack_state
        if st is not None and not self._pack_full:
            st = None                 # this version was packed PARTIALLY (in front of a replay): anything else gets every section
        if st is None or st[0] != ver or (need_dgrad and not st[1]):
            n = len(self._c3)
            fresh = st is not None and st[0] == ver            # the forward packs of this weight version exist already
            self._pack_full = True
            if not need_dgrad:
                self.ops.conv3_pack_many(self._desc_fwd, n)
            elif self._defer_dgrad_pack():
                # the forward needs the forward packs NOW and the dgrad packs only in the backward pass: the latter go to the
                # weight-gradient side stream (idle during the forward) and the backward waits for them (_wait_dgrad_packs)
                if not fresh:
                    self.ops.conv3_pack_many(self._desc_fwd, n)
                with self._on_pack_stream():
                    self.ops.conv3_pack_many(self._desc_dg, n)
            elif fresh:
                self.ops.conv3_pack_many(self._desc_dg, n)
            else:
                self.ops.conv3_pack_many(self._desc_all, 2 * n)
            self._pack_state = (ver, need_dgrad or (fresh and st[1]))
        return self._pack_views[key]

    # ---- dgrad packs off the forward's critical path (round 4): 37.8 MB of V-Net weights -> ~95 MB of packs per direction; the student's
    # forward used to wait for both directions (k_pack_conv3_many 2 x n descriptors: 65-100 us at the head of its stream)
    DEFER_DGRAD_PACK = False      # measured, not adopted (round 4, tools/sessions/r04_s18.sh): LA 5.29 vs 5.33 ms, ACDC / pancreas inside the noise, and the
                                  # captured-backward mode (plan.GRAPHS = 2) lost its bit-identity with the eager path in test_graph_replays_equal_eager_path

    def _defer_dgrad_pack(self):
        # (asked for only with need_dgrad, i.e. from a forward that saves for backward; NOT torch.is_grad_enabled(): that is False inside
        #  autograd.Function.forward, where the network passes run -- the first version of this switch never fired)
        return self.DEFER_DGRAD_PACK and self.overlap_wgrad and self._flat.is_cuda and self.ops.b._rec is None and self.training

    def _on_pack_stream(self):
        dev = self._flat.device
        main = torch.cuda.current_stream(dev)
        side = self._side_streams.get(dev)
        if side is None:
            side = self._side_streams[dev] = torch.cuda.Stream(device=dev, priority=HipNet.WGRAD_STREAM_PRIORITY)
        side.wait_stream(main)                 # the optimiser's update of the weights is ordered before the pack
        self._dgrad_pack_pending = (dev, side)
        return torch.cuda.stream(side)

    def _wait_dgrad_packs(self):
        """called at the head of the backward pass: the main stream joins the side stream's dgrad packs"""
        pend = self.__dict__.get("_dgrad_pack_pending")
        if pend is not None:
            dev, side = pend
            torch.cuda.current_stream(dev).wait_stream(side)
            self._dgrad_pack_pending = None

    # ---- same for the k2s2 / 1x1 GEMM weights: (fwd kind, dgrad kind) per layer, one launch per weight version
    def register_k2(self, key, weight, Cin, Cout, kind_fwd, kind_dgrad):
        if not hasattr(self, "_k2"):
            self._k2 = []
        self._k2.append((key, weight, Cin, Cout, kind_fwd, kind_dgrad))

    def _build_k2_tables(self):
        dev = self._flat.device
        total = sum(w.numel() for _, w, *_ in self._k2)
        self._k2_buf = torch.empty(2 * total, dtype=torch.float32, device=dev)
        self._k2_views = {}
        fwd_desc, all_desc, dg_desc = b"", b"", b""
        off = 0
        for key, w, Cin, Cout, kf, kd in self._k2:
            n = w.numel()
            bf, bd = self._k2_buf[off:off + n], self._k2_buf[total + off:total + off + n]
            self._k2_views[key] = (bf, bd)
            d_f = self.ops.k2_pack_desc(w.data, bf, Cin, Cout, kf)
            d_d = self.ops.k2_pack_desc(w.data, bd, Cin, Cout, kd)
            fwd_desc += d_f
            all_desc += d_f + d_d
            dg_desc += d_d
            off += n
        self._k2_desc_fwd = torch.frombuffer(bytearray(fwd_desc), dtype=torch.uint8).to(dev)
        self._k2_desc_all = torch.frombuffer(bytearray(all_desc), dtype=torch.uint8).to(dev)
        self._k2_desc_dg = torch.frombuffer(bytearray(dg_desc), dtype=torch.uint8).to(dev)
        self._k2_state = None
        self._k2_ptr = self._flat.data_ptr()

    def k2_packed(self, key, need_dgrad):
        """(B_fwd, B_dgrad) of k2 / 1x1 layer `key`, repacking every such layer in one launch when any weight changed"""
        if getattr(self, "_k2_ptr", None) != self._flat.data_ptr():
            self._build_k2_tables()
        ver = self._weights_version()
        st = self._k2_state
        if st is None or st[0] != ver or (need_dgrad and not st[1]):
            n = len(self._k2)
            fresh = st is not None and st[0] == ver
            if not need_dgrad:
                self.ops.k2_pack_many(self._k2_desc_fwd, n)
            elif self._defer_dgrad_pack():            # (see conv3_packed)
                if not fresh:
                    self.ops.k2_pack_many(self._k2_desc_fwd, n)
                with self._on_pack_stream():
                    self.ops.k2_pack_many(self._k2_desc_dg, n)
            elif fresh:
                self.ops.k2_pack_many(self._k2_desc_dg, n)
            else:
                self.ops.k2_pack_many(self._k2_desc_all, 2 * n)
            self._k2_state = (ver, need_dgrad or (fresh and st[1]))
        return self._k2_views[key]

    def _packed(self, key, p, fn):
        ver = (p._version, self._bump, p.data_ptr())
        hit = self._pack_cache.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, fn())
            self._pack_cache[key] = hit
        return hit[1]

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.bump()
        for m in self.modules():
            if isinstance(m, BNP) and getattr(m, "_live", False):
                self._nbt = int(m.num_batches_tracked)
                self._nbt_dirty = False
                break
        return r

    # ---- side stream for weight gradients (no consumer inside the backward pass)
    class _OnSide:
        def __init__(self, net, tensors):
            self.net, self.tensors = net, tensors
            self.ctx = None

        def __enter__(self):
            t = self.tensors[0]
            if not t.is_cuda or not self.net.overlap_wgrad:
                return self
            main = torch.cuda.current_stream(t.device)
            side = self.net._side_streams.get(t.device)
            if side is None:
                side = self.net._side_streams[t.device] = torch.cuda.Stream(device=t.device, priority=HipNet.WGRAD_STREAM_PRIORITY)
            b = self.net.ops.b
            if b._rec is not None:                 # recorded pass: the fork is an entry of the launch list (bcp_stream_wait_stream)
                b.call("bcp_stream_wait_stream", side.cuda_stream, main.cuda_stream)
            else:
                side.wait_stream(main)             # dy (and the zeroed gradient buffer) are ready
            for x in self.tensors:
                x.record_stream(side)              # keep the allocator from recycling them under the side stream
                am = getattr(x, "_bcp_amax", None)
                if am is not None:
                    am.record_stream(side)         # ... and the |max| slots the side stream's kernels read next to them (round 4: an eager
                                                   # step's slot buffer was recycled under the weight-gradient kernel -- plans keep theirs)
            self.ctx = torch.cuda.stream(side)
            self.ctx.__enter__()
            return self

        def __exit__(self, *a):
            if self.ctx is not None:
                self.ctx.__exit__(*a)
            return False

    overlap_wgrad = True
    WGRAD_STREAM_PRIORITY = 0      # measurement switch (bench.py --opt wgrad_prio=..)
    _side_streams = {}

    def _wgrad_stream(self, *tensors):
        return HipNet._OnSide(self, tensors)

    def _join_wgrad_stream(self, like):
        if like.is_cuda and self.overlap_wgrad:
            side = self._side_streams.get(like.device)
            if side is not None:
                main = torch.cuda.current_stream(like.device)
                if self.ops.b._rec is not None:
                    self.ops.b.call("bcp_stream_wait_stream", main.cuda_stream, side.cuda_stream)
                else:
                    main.wait_stream(side)

    # ---- data parallelism: bucketed gradient all-reduce underneath the rest of the backward pass (bcp_amd/dp.py).
    # The backward walks the layers in reverse registration order, so once layer L is done the flat gradient buffer is final
    # from L's first parameter to its end; the hook (set by DataParallel.arm for the step's single backward) may start
    # reducing that suffix while the shallower layers are still being differentiated.
    _grad_bucket_hook = None

    def _grad_stages(self):
        """[(marker parameter, [parameters whose gradients are final once the marker's layer is done]), ...] in backward order"""
        raise NotImplementedError

    def _bucket_plan(self):
        """marker parameter id -> first element of the flat gradient buffer that is final at that point of the backward pass.
        Checked, not assumed: a suffix [lo, end) is handed out only if the parameters living there are exactly the ones
        processed so far; a network whose registration order is not its layer order gets an empty plan (one all-reduce at the end)."""
        cached = self.__dict__.get("_bplan")
        if cached is not None and cached[0] is self._offs:
            return cached[1]
        offs = sorted(self._offs[id(p)] for p in self._opt_plist)
        plan, done, lo = {}, 0, None
        for marker, ps in self._grad_stages():
            done += len(ps)
            lo = min([self._offs[id(q)] for q in ps] + ([lo] if lo is not None else []))
            if len(offs) - done < 0 or offs[len(offs) - done] != lo:
                plan = {}
                break
            plan[id(marker)] = lo
        object.__setattr__(self, "_bplan", (self._offs, plan))
        return plan

    def _grads_final_from(self, p, like):
        if self._grad_bucket_hook is None:
            return
        lo = self._bucket_plan().get(id(p))
        if lo is None:
            return
        self._grad_bucket_hook(self, lo, like)
        rec = self.ops.b._rec
        if rec is not None:
            rec.add_py(self._replay_bucket_hook, lo, like, capturable=False)

    def _replay_bucket_hook(self, lo, like):
        hook = self._grad_bucket_hook          # the hook armed for THIS step, not the one seen while recording
        if hook is not None:
            hook(self, lo, like)

    _instances = 0   # per-process instance counter: every network gets its own dropout stream

    def seed_dropout(self, seed=None):
        """(re)seed this network's dropout stream.  Default: torch's seed (so --seed / torch.manual_seed govern dropout as in
        the reference, which draws its masks from the torch RNG), mixed with a per-process instance number (student and
        teacher -- and the pre-training / self-training models -- draw INDEPENDENT masks, as two nn.Dropout modules do) and with
        the data-parallel rank (one stream per replica, bcp_amd/dp.py)."""
        if seed is None:
            HipNet._instances += 1
            seed = (torch.initial_seed() & 0xFFFFFFFFFFFFFFFF) ^ (HipNet._instances * 0x9E3779B97F4A7C15) ^ (int(os.environ.get("RANK", "0")) * 0xD1B54A32D192ED03)
        z = (int(seed) + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF        # splitmix64 finaliser: nearby seeds -> unrelated streams
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        self._drop_seed = z ^ (z >> 31)
        return self

    def next_seed(self):
        if self._drop_seed is None:
            self.seed_dropout()
        self._drop_seed = (self._drop_seed * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return self._drop_seed


    # ---- recorded launch plans (bcp_amd/plan.py): a training-mode pass is recorded once per (shape, groups, mode, stream) and
    # replayed afterwards -- one ctypes call per launch, no Python per layer
    use_plans = True

    def _plan_ok(self, x):
        from .. import plan
        return (plan.ENABLED and self.use_plans and self.training and self.drop_masks is None and not getattr(self, "_keep_saved", False)
                and (x.is_cuda or self.ops.allow_cpu) and self.ops.b._rec is None)

    # ---- round 6: partial packs in front of REPLAYS.  A pack holds three sections -- the fp32 pack (conv3.hip's kernels), three bf16 planes and
    # two fp16 planes (conv3b.hip) -- and a launch reads exactly one; every network repacks every weight every step (500 MB per LA step, of which
    # the step's launches read 130).  Ops notes after every eager / recorded forward or dgrad launch which section it read
    # (bcp_conv3_last_section) for the (layer, direction) its packed weight belongs to; a REPLAY of a recorded pass issues exactly the launches
    # that were observed when it was recorded, so in front of a replay only the observed sections are written (per-descriptor masks,
    # bcp_conv3_pack_many).  Everything else -- eager passes, the recording of a new plan, a validation forward -- asks conv3_packed(), which
    # repacks every section when the current version was packed partially.  LA 757.1 -> 764.3 volumes/s as an upper bound (gpurun_out/r06_s18).
    PACK_PARTIAL = os.environ.get("BCP_PACK_PARTIAL", "1") != "0"

    def pack_section_used(self, key, direction, sec):
        k = (key, direction)
        m = self._pack_need.get(k, 0)
        if sec & ~m:
            self._pack_need[k] = m | sec
            self._pack_need_ver += 1

    def _partial_descs(self):
        import struct
        hit = self._pack_partial
        if hit is None or hit[0] != self._pack_need_ver:
            fwd, both = b"", b""
            for key, wptr, wfp, wdp, Cout, Cin, T, K16, N16 in self._pack_items:
                sf, sd = self._pack_need.get((key, 0), 0), self._pack_need.get((key, 1), 0)      # 0: never observed -> every section
                d_f = struct.pack("<QQiiiiii", wptr, wfp, Cout, Cin, T, K16, N16, 0 | (sf << 8))
                d_d = struct.pack("<QQiiiiii", wptr, wdp, Cout, Cin, T, N16, K16, 1 | (sd << 8) | 0x800)      # (0x800: |max| from the forward twin in front)
                fwd += d_f
                both += d_f + d_d
            dev = self._flat.device
            hit = (self._pack_need_ver, torch.frombuffer(bytearray(fwd), dtype=torch.uint8).to(dev), torch.frombuffer(bytearray(both), dtype=torch.uint8).to(dev))
            self._pack_partial = hit
        return hit[1], hit[2]

    def _ensure_packed_replay(self, need_dgrad):
        """_ensure_packed in front of the REPLAY of a recorded pass: only the sections its launches were seen reading"""
        c3 = getattr(self, "_c3", None)
        if c3 and self.PACK_PARTIAL and getattr(self, "_pack_ptr", None) == self._flat.data_ptr() and self._pack_need:
            ver = self._weights_version()
            st = self._pack_state
            if st is None or st[0] != ver or (need_dgrad and not st[1]):
                fwd, both = self._partial_descs()
                n = len(self._c3)
                if need_dgrad:
                    self.ops.conv3_pack_many(both, 2 * n)
                else:
                    self.ops.conv3_pack_many(fwd, n)
                self._pack_state = (ver, bool(need_dgrad))
                self._pack_full = False
        elif c3:
            self.conv3_packed(c3[0][0], need_dgrad)
        k2 = getattr(self, "_k2", None)
        if k2:
            self.k2_packed(k2[0][0], need_dgrad)

    def _ensure_packed(self, need_dgrad):
        """weight packs depend on the weights' version, not on the pass: (re)pack eagerly, outside any plan"""
        c3 = getattr(self, "_c3", None)
        if c3:
            self.conv3_packed(c3[0][0], need_dgrad)
        k2 = getattr(self, "_k2", None)
        if k2:
            self.k2_packed(k2[0][0], need_dgrad)

    def _plans_for(self):
        from .. import plan
        st = self.__dict__.get("_plan_state")
        tag = (plan.epoch(), self._flat.data_ptr(), self._flat_grad.data_ptr() if self._flat_grad is not None else 0)
        if st is None or st[0] != tag:
            st = (tag, {})
            object.__setattr__(self, "_plan_state", st)
        return st[1]

    # volatile_io (the training scripts and bench.py set it; off by default): the caller consumes a replayed pass's results before the
    # NEXT pass of this network and fills its inputs in place -- the logits are handed out as an alias of the plan's static tensor instead
    # of a copy, input_buffer() / dout_buffer() hand the plan's static input tensors to the producers (the copy-paste mix, the loss
    # backward), and a pass whose input already IS that tensor skips its copy: three device copies per student pass (8 + 16 + 16 MB at the
    # LA size, ~30 us with their launch gaps) and one per teacher pass.  Everything else -- launch lists, arithmetic, results -- is the same.
    volatile_io = False

    def _io_plan(self, kind, shape):
        if not self.volatile_io or self.__dict__.get("_plan_state") is None:
            return None
        self._ensure_flat()
        pl = self._plans_for().get((kind, tuple(shape)))
        return None if (pl is None or pl.busy) else pl

    def input_buffer(self, xshape):
        """a logical view [N, 1, ...] of the static input of the training-forward plan for inputs of this shape (None: no such plan yet, or
        busy, or volatile_io off): what is written there needs no copy into the plan"""
        xshape = tuple(xshape)
        if len(xshape) not in (4, 5) or xshape[1] != 1:
            return None
        cl = (xshape[0],) + ((1,) if len(xshape) == 4 else ()) + xshape[2:] + (1,)
        pl = self._io_plan("in", cl)
        return None if pl is None else pl.static_in.view(xshape)

    def dout_buffer(self, clshape):
        """the static input of the backward plan for logits gradients of this (channels-last) shape, or None"""
        pl = self._io_plan("bin", clshape)
        return None if pl is None else pl.static_in

    def dout_buffer_for(self, logits):
        """dout_buffer for the gradient of `logits` -- only when `logits` IS the alias this network's last training pass handed out (same
        storage): a loss on some other tensor of the same shape (another network's output, a clone) gets None and allocates its own"""
        if logits is None or logits.data_ptr() != self.__dict__.get("_volatile_logits_ptr"):
            return None
        return self.dout_buffer(tuple(logits.shape))

    def _run_forward(self, xcl, save):
        if not self._plan_ok(xcl):
            r = self._forward_impl(xcl, save)
            self._feat_out = getattr(r[0], "_bcp_feat", None)
            return r
        from .. import plan
        self._ensure_flat()
        plans = self._plans_for()
        # (every switch that changes the launch list is part of the key: a toggled switch must never replay the other sequence)
        key = ("f", tuple(xcl.shape), getattr(self, "_groups", 1), bool(save), bool(getattr(self, "_turnoff_drop", False)), self.ops.stream(xcl),
               bool(getattr(self, "fuse_head", False)), bool(getattr(self, "fuse_c1", False)), bool(self.overlap_wgrad),
               bool(getattr(self, "_keep_saved", False)), bool(getattr(self, "_want_feat", False)), bool(getattr(self, "skip_in_concat", False)), bool(getattr(self, "inline_dropout", False)))
        pl = plans.get(key)
        if pl is not None and pl.busy:
            # the plan's static activations belong to a forward whose backward has not run yet (the unfused loop calls the student
            # twice per step): this call must not overwrite them
            r = self._forward_impl(xcl, save)
            self._feat_out = getattr(r[0], "_bcp_feat", None)
            return r
        if pl is None:
            self._ensure_packed(save)              # a pass about to be RECORDED runs eagerly: every section
        else:
            self._ensure_packed_replay(save)
        if pl is None:
            pl = plan.LaunchPlan()
            pl.static_in = torch.empty_like(xcl)
            pl.static_in.copy_(xcl)
            t0 = getattr(self, "_nbt", 0)
            with plan.recording(self.ops, pl):
                pl.result = self._forward_impl(pl.static_in, save)
            pl.ticks = getattr(self, "_nbt", 0) - t0
            plans[key] = pl
        else:
            if xcl.data_ptr() != pl.static_in.data_ptr():       # (volatile_io: the producer wrote into input_buffer())
                pl.static_in.copy_(xcl)
            pl.replay(self.ops, [self.next_seed() for _ in range(pl.n_seeds)], xcl)
            for _ in range(pl.ticks):
                self._nbt_tick()
        if save:
            plans[("in", tuple(xcl.shape))] = pl
        out, saved = pl.result
        f = getattr(out, "_bcp_feat", None)
        self._feat_out = None if f is None else f.clone()
        # the logits leave the plan as a COPY (16 MB at the LA size): callers may hold them across the next replay (logging, the
        # reference's unfused loop), and a replay overwrites the plan's static tensors in place
        res = out.detach() if (self.volatile_io and self.training) else out.clone()      # volatile_io (training passes only): an alias, valid until the next pass of this network
        object.__setattr__(self, "_volatile_logits_ptr", res.data_ptr() if (self.volatile_io and self.training and save) else None)
        if save:
            ps = _PlanSaved(saved, pl)
            pl.busy = ps.token          # cleared by the backward pass -- or when `ps` dies without one (a discarded loss, an exception)
            return res, ps
        return res, None

    def _run_backward(self, saved, dout):
        self._wait_dgrad_packs()
        if not isinstance(saved, _PlanSaved):
            return self._backward_impl(saved, dout)
        from .. import plan
        fwd, tok, saved = saved.plan, saved.token, saved.saved
        try:
            if not self._plan_ok(dout):
                return self._backward_impl(saved, dout)
            plans = self._plans_for()
            key = ("b", id(saved), tuple(dout.shape), self.ops.stream(dout), self._grad_bucket_hook is not None,
                   bool(self.overlap_wgrad))
            pl = plans.get(key)
            if pl is None:
                self._ensure_packed(True)
            else:
                self._ensure_packed_replay(True)
            if pl is None:
                pl = plan.LaunchPlan()
                pl.static_in = torch.empty(tuple(dout.shape), dtype=dout.dtype, device=dout.device)
                pl.static_in.copy_(dout)
                pl.keep.append(saved)           # the forward plan's tensors this pass reads
                with plan.recording(self.ops, pl):
                    self._backward_impl(saved, pl.static_in)
                plans[key] = pl
            else:
                if dout.data_ptr() != pl.static_in.data_ptr():      # (volatile_io: the loss backward wrote into dout_buffer())
                    pl.static_in.copy_(dout)
                else:
                    object.__setattr__(self, "_dout_in_place", self.__dict__.get("_dout_in_place", 0) + 1)      # (tests: the copy really is gone)
                self.begin_backward()
                pl.replay(self.ops, (), dout)
            plans[("bin", tuple(dout.shape))] = pl
        finally:
            if fwd.busy is tok:
                fwd.busy = False
        return None


class _PlanSaved:
    """what NetFn keeps between forward and backward when the forward came from a launch plan"""
    __slots__ = ("saved", "plan", "token")

    def __init__(self, saved, plan):
        self.saved, self.plan, self.token = saved, plan, object()

    def __del__(self):
        # a training-mode forward that is never backpropagated (loss discarded, an exception, a logging forward with grad enabled)
        # must not leave its plan busy for ever: every later pass of the net would silently take the eager path
        try:
            if self.plan.busy is self.token:
                self.plan.busy = False
        except Exception:
            pass


class NetFn(torch.autograd.Function):
    """One autograd node for a whole network call.  `anchor` is any trainable parameter: it makes the
    output require grad; parameter gradients are written by the kernels straight into the flat
    gradient buffer (p.grad views), so backward returns None for it."""

    @staticmethod
    def forward(ctx, x, anchor, net):
        out, saved = net._run_forward(x, True)
        if getattr(net, "_keep_saved", False):      # parity tests: expose (y, stats) per layer -> activation patterns (tests/net_checks.py)
            net._last_saved = saved
        ctx.net = net
        ctx.saved = saved
        return out

    @staticmethod
    def backward(ctx, dout):
        ctx.net._run_backward(ctx.saved, dout)
        ctx.saved = None
        return None, None, None
