"""tools/attic/overlap_step.py -- NOT part of the package since round 4 (kept for the record).
OverlapStep: optimiser + EMA + weight re-pack applied range by range UNDERNEATH the backward pass (bit-identical to the plain step,
parity-tested in round 3).  Measured no faster: LA 6.26-6.30 vs 6.16-6.22 ms, pancreas 5.66-5.71 vs 5.59-5.62 -- the tail it hides is
shorter than the launches and host callbacks it adds.  The text below is the code as it stood at the end of round 3 (class, the range
helpers of FlatSGD / FlatAdam and the pack bookkeeping of HipNet it used); it is not importable on its own."""

# ---- FlatSGD methods (bcp_amd/train_step.py)
    # ---- the same update a range at a time (OverlapStep): elementwise, so bit-identical to step()
    def begin_step(self):
        p, _ = self.model.flat_trainable()
        if self.buf is None:
            self.buf = torch.zeros_like(p)
        return self.steps == 0

    def step_range(self, lo, hi, first):
        g = self.param_groups[0]
        p, gr = self.model.flat_trainable()
        _ops_for(p).sgd(p[lo:hi], gr[lo:hi], self.buf[lo:hi], g["lr"], g["momentum"], g["weight_decay"], first_step=first, grad_scale=self.grad_scale)

    def end_step(self):
        self.steps += 1



# ---- FlatAdam methods (bcp_amd/train_step.py)
    # ---- the same update a range at a time (OverlapStep)
    def begin_step(self):
        p, _ = self.model.flat_trainable()
        if self.m is None:
            self.m, self.v = torch.zeros_like(p), torch.zeros_like(p)
        self.steps += 1
        return self.steps

    def step_range(self, lo, hi, step_no):
        g = self.param_groups[0]
        p, gr = self.model.flat_trainable()
        _ops_for(p).adam(p[lo:hi], gr[lo:hi], self.m[lo:hi], self.v[lo:hi], g["lr"], step_no, g["betas"][0], g["betas"][1], g["eps"], grad_scale=self.grad_scale)

    def end_step(self):
        pass



# ---- bcp_amd/train_step.py
class OverlapStep:
    """The optimiser step, the EMA and the weight re-pack UNDERNEATH the backward pass (round 3).

    The backward pass walks the layers in reverse registration order, so once layer L is done the flat gradient buffer is final from
    L's first parameter to its end (the signal the data-parallel bucket exchange uses, bcp_amd/dp.py).  Every time >= `bucket_mb` of
    new final gradients exist, this object applies -- on the weight-gradient side stream, ordered after the main stream -- the
    optimiser update of that range (bcp_sgd / bcp_adam: elementwise, so a range at a time is bit-identical to one launch), the
    teacher's EMA of the same range, and the re-pack of the conv / k2 weights living there (student: forward + dgrad packs, teacher:
    forward).  What is left for `finish()` after the backward pass is the shallow remainder (a few hundred KB) plus the EMA of the
    non-trainable tail.  The deep levels hold 80 % of the parameters and are differentiated first, so SGD + EMA + 2 x pack (0.2 ms at
    the head of the next step, where nothing overlapped them) mostly disappear from the critical path.
    Only for ONE backward pass per step (grouped student batches) and without data parallelism (its buckets must be reduced first)."""

    def __init__(self, optimizer, model, ema_model, alpha, ema_whole_state, bucket_mb=None, first_frac=0.6):
        """bucket_mb None (the product): ONE early range -- as soon as `first_frac` of the trainable buffer is final (the end of the deep
        levels) -- then the remainder in finish(): every range costs a Python callback in the middle of the replayed backward pass, and
        with 4 MB ranges (nine callbacks) the host fell behind the GPU at the deep levels (LA step 6.33 vs 6.11 ms).  A number: that many
        MB per range (tests)."""
        self.opt, self.model, self.ema, self.alpha, self.whole = optimizer, model, ema_model, alpha, ema_whole_state
        model._ensure_flat()
        ema_model._ensure_flat()
        self.hi = model._n_trainable_flat
        self.bucket = int(bucket_mb * (1 << 20)) // 4 if bucket_mb is not None else int(first_frac * self.hi)
        self.once = bucket_mb is None
        self.ok = (ema_model._n_trainable_flat == self.hi and ema_model.flat_state().numel() == model.flat_state().numel())
        self.first = None
        self.stream = None

    def arm(self):
        if self.ok:
            self.first = self.opt.begin_step()
            self.model._opt_bucket_hook = self._on_final
        return self.ok

    def _apply(self, lo, hi):
        from . import plan
        ops = self.model.ops
        with plan.suspended(ops):               # never part of a recorded pass: lr / step count change from step to step
            self.opt.step_range(lo, hi, self.first)
            ops.ema(self.ema.flat_state()[lo:hi], self.model.flat_state()[lo:hi], self.alpha)
            self.model.pack_range(lo, hi, True)
            self.ema.pack_range(lo, hi, False)

    def _on_final(self, model, lo, like):
        if lo >= self.hi or self.hi - lo < self.bucket:
            return
        if like.is_cuda:
            # a stream of its own: ordered after the main stream (norm / bias gradients) and the weight-gradient side stream, it runs
            # next to both -- HBM-bound launches underneath MFMA-bound ones -- and delays neither
            dev = like.device
            st = _OPT_STREAMS.get(dev)
            if st is None:
                st = _OPT_STREAMS[dev] = torch.cuda.Stream(device=dev)
            st.wait_stream(torch.cuda.current_stream(dev))
            side = model._side_streams.get(dev) if model.overlap_wgrad else None
            if side is not None:
                st.wait_stream(side)
            with torch.cuda.stream(st):
                self._apply(lo, self.hi)
            self.stream = st
        else:
            self._apply(lo, self.hi)
        self.hi = lo
        if self.once:
            self.bucket = 1 << 62                # one early range only

    def finish(self):
        """after loss.backward() (the side stream has been joined): the remainder, the EMA of the non-trainable tail, bookkeeping"""
        self.model._opt_bucket_hook = None
        if self.stream is not None:
            torch.cuda.current_stream(self.stream.device).wait_stream(self.stream)
        if self.hi > 0:
            self._apply(0, self.hi)
        n_tr = self.model._n_trainable_flat
        src = self.model.flat_state() if self.whole else self.model.flat_params()
        dst = self.ema.flat_state() if self.whole else self.ema.flat_params()
        if src.numel() > n_tr:
            self.model.ops.ema(dst[n_tr:], src[n_tr:], self.alpha)        # unused heads (and, for the state-dict EMA, the BN buffers)
        self.opt.end_step()
        self.model.mark_packed(True)
        self.ema.mark_packed(False)
        if self.whole:
            a, b = float(getattr(self.ema, "_nbt", 0)), float(getattr(self.model, "_nbt", 0))
            self.ema._nbt = int(np.float32(np.float32(self.alpha) * np.float32(a)) + np.float32(np.float32(1 - self.alpha) * np.float32(b)))
            self.ema._nbt_dirty = True


_OPT_STREAMS = {}
OVERLAP_STEP = False       # module switch (bench.py --opt overlap_step=1; parity tests compare both).  OFF: measured no faster (LA 6.26-6.30 vs
                           # 6.16-6.22 ms, pancreas 5.66-5.71 vs 5.59-5.62): optimiser / EMA / pack launches at the step boundary are already
                           # hidden -- the step is bound by its MFMA kernels and the deep levels' dependency chains


def _overlap_for(optimizer, model, ema_model, alpha, whole, dp, grouped):
    from .networks._hipnet import HipNet
    from . import plan
    if not (OVERLAP_STEP and grouped and dp is None and isinstance(optimizer, (FlatSGD, FlatAdam)) and isinstance(model, HipNet)
            and isinstance(ema_model, HipNet) and optimizer.model is model and plan.GRAPHS < 2):      # (a captured backward takes no callbacks)
        return None
    ov = OverlapStep(optimizer, model, ema_model, alpha, whole)
    return ov if ov.arm() else None




# ---- HipNet methods (bcp_amd/networks/_hipnet.py)
    # ---- packed-weight bookkeeping for the overlapped optimiser step: which conv / k2 layers have their weight inside a flat range,
    # and "everything is packed for the current weights" without a launch
    def _pack_index(self):
        idx = self.__dict__.get("_pidx")
        if idx is None or idx[0] is not self._offs:
            c3 = [self._offs[id(w)] for _, w, _ in getattr(self, "_c3", ())]
            k2 = [self._offs[id(w)] for _, w, *_ in getattr(self, "_k2", ())]
            assert c3 == sorted(c3) and k2 == sorted(k2), "layers are registered in flat-buffer order"
            idx = (self._offs, c3, k2)
            object.__setattr__(self, "_pidx", idx)
        return idx[1], idx[2]

    def pack_range(self, lo, hi, need_dgrad):
        """re-pack the conv / k2 weights living in flat elements [lo, hi) (their layers form a contiguous run of the descriptor tables)"""
        import bisect
        if getattr(self, "_pack_ptr", None) != self._flat.data_ptr():
            self._build_pack_tables()
        if getattr(self, "_k2", None) and getattr(self, "_k2_ptr", None) != self._flat.data_ptr():
            self._build_k2_tables()
        c3, k2 = self._pack_index()
        i0, i1 = bisect.bisect_left(c3, lo), bisect.bisect_left(c3, hi)
        if i1 > i0:
            if need_dgrad:
                self.ops.conv3_pack_many(self._desc_all[80 * i0:80 * i1], 2 * (i1 - i0))
            else:
                self.ops.conv3_pack_many(self._desc_fwd[40 * i0:40 * i1], i1 - i0)
        j0, j1 = bisect.bisect_left(k2, lo), bisect.bisect_left(k2, hi)
        if j1 > j0:
            if need_dgrad:
                self.ops.k2_pack_many(self._k2_desc_all[128 * j0:128 * j1], 2 * (j1 - j0))
            else:
                self.ops.k2_pack_many(self._k2_desc_fwd[64 * j0:64 * j1], j1 - j0)

    def mark_packed(self, has_dgrad):
        """the flat weights changed (bump) AND every pack was refreshed range by range: the next forward must not pack again"""
        self.bump()
        ver = self.refresh_weights_version()
        if getattr(self, "_c3", None):
            self._pack_state = (ver, bool(has_dgrad))
        if getattr(self, "_k2", None):
            self._k2_state = (ver, bool(has_dgrad))

