"""2-D U-Net on the MI355X kernels -- drop-in for the reference's networks/unet.py:UNet_2d (ACDC).

Same constructor, same 226 state_dict keys / parameter order, `net(x) -> logits [N,4,H,W]`.  Layout is NHWC
fp32 ([N,1,H,W,C] physically: the 3-D kernels run with D = 1, KD = 1).

Topology (networks/unet.py:15-57, 80-86, 104-116): ConvBlock = conv3x3 -> BN -> LeakyReLU(0.01) -> Dropout(p) ->
conv3x3 -> BN -> LeakyReLU; encoder in_conv(1->16, p=.05), down1..4 = MaxPool2d(2) + ConvBlock (32,64,128,256;
p = .1,.2,.3,.5); decoder up1..4 = conv1x1 (halve C) -> bilinear x2 (align_corners=True) -> cat([skip, up]) ->
ConvBlock(p=0); out_conv 3x3 16->4.  The upsample writes directly into its half of the concat buffer.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import hip_ops as H
from ._hipnet import BNP, ConvP, HipNet, Holder, NetFn, Seq, contrastive_heads

FT = [16, 32, 64, 128, 256]
DROP = [0.05, 0.1, 0.2, 0.3, 0.5]


class _CB:
    """one ConvBlock: two (conv, bn) pairs + its dropout probability and mask key"""

    def __init__(self, cin, cout, p, dkey):
        self.cin, self.cout, self.p, self.dkey = cin, cout, p, dkey
        self.c1, self.b1 = ConvP((cout, cin, 3, 3), cin * 9, cout), BNP(cout)
        self.c2, self.b2 = ConvP((cout, cout, 3, 3), cout * 9, cout), BNP(cout)
        self.b1._live = self.b2._live = True

    def module(self):
        h = Holder()
        h.conv_conv = Seq([(0, self.c1), (1, self.b1), (4, self.c2), (5, self.b2)])
        return h


class UNet_2d(HipNet):
    fuse_c1 = True       # first layer: conv + norm + LeakyReLU (+ dropout) with recompute (networks/VNet.py)
    skip_in_concat = True   # encoder outputs are written into the decoder's concat buffers (no torch.cat copy; pool backward joins the skip gradient)
    inline_dropout = True   # nn.Dropout keep bits evaluated inside the norm kernels from a device seed (hip_ops.SeedMask): no mask tensors
    def __init__(self, in_chns, class_num):
        super().__init__()
        assert in_chns == 1, "the ACDC hot path is single-channel"
        self.n_classes = class_num
        enc, dec = Holder(), Holder()
        self._enc = [_CB(in_chns, FT[0], DROP[0], "d0")] + [_CB(FT[i - 1], FT[i], DROP[i], f"d{i}") for i in range(1, 5)]
        enc.in_conv = self._enc[0].module()
        for i in range(1, 5):
            d = Holder()
            d.maxpool_conv = Seq([(1, self._enc[i].module())])
            setattr(enc, f"down{i}", d)
        self._up = []
        for i in range(1, 5):
            c1, c2 = FT[5 - i], FT[4 - i]
            u = Holder()
            pw = ConvP((c2, c1, 1, 1), c1, c2)
            cb = _CB(2 * c2, c2, 0.0, None)
            u.conv1x1 = pw
            u.conv = cb.module()
            setattr(dec, f"up{i}", u)
            self._up.append((pw, cb, c1, c2))
        dec.out_conv = ConvP((class_num, FT[0], 3, 3), FT[0] * 9, class_num)
        self.encoder, self.decoder = enc, dec
        contrastive_heads(self, 4)
        object.__setattr__(self, "_out", dec.out_conv)
        ids = set()
        for cb in self._enc + [u[1] for u in self._up]:
            for m in (cb.c1, cb.b1, cb.c2, cb.b2):
                ids.add(id(m.weight)); ids.add(id(m.bias))
        for pw, _, _, _ in self._up:
            ids.add(id(pw.weight)); ids.add(id(pw.bias))
        ids.add(id(self._out.weight)); ids.add(id(self._out.bias))
        self._opt_param_ids = ids
        for tag, cb in [(f"e{i}", c) for i, c in enumerate(self._enc)] + [(f"u{i}", u[1]) for i, u in enumerate(self._up, start=1)]:
            if cb.cin != 1:
                self.register_conv3((tag, 1), cb.c1.weight, 1)
            self.register_conv3((tag, 2), cb.c2.weight, 1)
        self.register_conv3(("out", 0), self._out.weight, 1)
        for i, (pw, _, c1, c2) in enumerate(self._up, start=1):
            self.register_k2(("pw", i), pw.weight, c1, c2, H.PACK_PW_FWD, H.PACK_PW_DGRAD)
        self._prenorm_bias_ids = set(id(m.bias) for cb in self._enc + [u[1] for u in self._up] for m in (cb.c1, cb.c2))

    # ------------------------------------------------------------------ public call
    def forward(self, x, groups=1):
        """groups: see networks/VNet.py:forward -- `groups` sub-batches normalised separately, launched together"""
        N = x.shape[0]
        assert N % groups == 0
        self._groups = int(groups)
        assert x.dim() == 4 and x.shape[1] == 1, "expected [N,1,H,W]"
        self._ensure_flat()
        self.refresh_weights_version()
        xcl = x.contiguous().view(N, 1, x.shape[2], x.shape[3], 1)
        anchor = self._enc[0].c1.weight
        if self.training and torch.is_grad_enabled() and anchor.requires_grad:
            out = NetFn.apply(xcl, anchor, self)
        else:
            out, _ = self._run_forward(xcl, False)
        return out.permute(0, 4, 1, 2, 3).squeeze(2)   # logical [N,4,H,W]

    # ------------------------------------------------------------------ pieces
    def _elem_mask(self, cb, shape, like):
        """the block's nn.Dropout keep mask: injected (parity runs: a uint8 tensor), or drawn -- as a SeedMask the norm kernels evaluate in
        place (inline_dropout, round 4: no mask tensor, no launch), else as a uint8 tensor from bcp_bernoulli with the SAME bits"""
        dev = like.device
        if cb.p <= 0.0 or not self.training:
            return None
        if self.drop_masks is not None:
            m = self.drop_masks[cb.dkey]                       # logical [N,C,H,W] keep mask
            return m.to(dev).permute(0, 2, 3, 1).unsqueeze(1).contiguous().to(torch.uint8)
        if self.inline_dropout:
            return self.ops.seed_mask(shape, 1.0 - cb.p, self.next_seed(), like)
        m = torch.empty(shape, dtype=torch.uint8, device=dev)
        return self.ops.bernoulli(m, 1.0 - cb.p, 1.0, self.next_seed())

    def _convblock_fwd(self, cb, tag, h, save, saved, out=None):
        """out: where the block's output goes when the caller has a place for it -- the leading channels of the decoder's concat buffer
        (Ops.channel_slab).  Returned as is when it was used; otherwise the block returns a tensor of its own (eval mode, the raw-slab
        norm of the deepest level) and the caller copies."""
        ops = self.ops
        G = getattr(self, "_groups", 1)
        if not self.training:      # model.eval(): running statistics, no dropout (val_2d / test scripts)
            if cb.cin == 1:
                y1 = ops.conv3_c1_fwd(h, cb.c1.weight.data, cb.c1.bias.data, 1)
            else:
                y1 = ops.conv3_fwd(h, self.conv3_packed((tag, 1), False)[0], cb.c1.bias.data, cb.cout, 1)
            a1 = ops.norm_eval(y1, cb.b1.weight.data, cb.b1.bias.data, cb.b1.running_mean, cb.b1.running_var, H.ACT_LRELU)
            y2 = ops.conv3_fwd(a1, self.conv3_packed((tag, 2), False)[0], cb.c2.bias.data, cb.cout, 1)
            return ops.norm_eval(y2, cb.b2.weight.data, cb.b2.bias.data, cb.b2.running_mean, cb.b2.running_var, H.ACT_LRELU)
        N, _, Hh, Ww, _ = h.shape
        oshape = (N, 1, Hh, Ww, cb.cout)
        # deepest level (<= 4096 rows per group) with split-K convs: the conv leaves its raw slabs and the norm's statistics pass sums them
        # on its way in (networks/VNet.py, bcp_norm_fwd_slabs) -- no slab-sum launch
        slabs_ok = ops.norm_slabs_ok(G, N * Hh * Ww // G, cb.cout)
        es = 1.0 / (1.0 - cb.p) if cb.p > 0 else 1.0
        em = self._elem_mask(cb, oshape, h)
        b1, b2 = cb.b1, cb.b2
        part1, nb1 = None, 0
        src, nsl, bsrc = None, 1, None
        fused_c1 = (cb.cin == 1 and self.fuse_c1 and not getattr(self, "_keep_saved", False) and ops.conv3_c1_norm_ok(h.shape, 1, G))
        if fused_c1:
            y1 = None      # conv + norm + LeakyReLU + dropout with recompute (bcp_conv3_c1_norm_fwd): y1 is never written
        elif cb.cin == 1:
            y1, part1, nb1 = ops.conv3_c1_fwd_stats(h, cb.c1.weight.data, cb.c1.bias.data, 1, G)
        else:
            wf, _ = self.conv3_packed((tag, 1), save)
            sk = ops.conv3_nslabs(h.shape, cb.cout, 1) if slabs_ok else 0
            if sk > 1:
                src, nsl, bsrc = ops.conv3_fwd_raw(h, wf, cb.cout, 1, sk), sk, cb.c1.bias.data
            else:
                y1, part1, nb1 = ops.conv3_fwd_stats(h, wf, cb.c1.bias.data, cb.cout, 1, G)
        if fused_c1:
            a1, st1 = ops.conv3_c1_norm_fwd(h, cb.c1.weight.data, cb.c1.bias.data, 1, G, b1.weight.data, b1.bias.data, b1.running_mean, b1.running_var,
                                            H.ACT_LRELU, elem_mask=em, elem_scale=es)
        elif src is not None:
            a1, st1, y1 = ops.norm_fwd_slabs(src, nsl, bsrc, G, b1.weight.data, b1.bias.data, b1.running_mean, b1.running_var,
                                             H.ACT_LRELU, elem_mask=em, elem_scale=es)
        else:
            a1, st1 = ops.norm_fwd(y1, G, b1.weight.data, b1.bias.data, b1.running_mean, b1.running_var, H.ACT_LRELU,
                                   elem_mask=em, elem_scale=es, partial=part1, nb=nb1)
        wf2, _ = self.conv3_packed((tag, 2), save)
        sk = ops.conv3_nslabs(a1.shape, cb.cout, 1) if slabs_ok else 0
        if sk > 1:
            a2, st2, y2 = ops.norm_fwd_slabs(ops.conv3_fwd_raw(a1, wf2, cb.cout, 1, sk), sk, cb.c2.bias.data, G, b2.weight.data, b2.bias.data,
                                             b2.running_mean, b2.running_var, H.ACT_LRELU)
        else:
            y2, part2, nb2 = ops.conv3_fwd_stats(a1, wf2, cb.c2.bias.data, cb.cout, 1, G)
            a2, st2 = ops.norm_fwd(y2, G, b2.weight.data, b2.bias.data, b2.running_mean, b2.running_var, H.ACT_LRELU, partial=part2, nb=nb2,
                                   out=out)
        if save:
            saved[tag] = (h, y1, st1, em, a1, y2, st2, G)
        return a2

    # (round 6) weight gradients cross to the side stream in batches: every fork is an event record between two kernels of the main stream
    # (networks/VNet.py WGRAD_DEFER); launches on >= 2^22 elements of dy fork at once
    WGRAD_DEFER = 2

    _wg_pend = None

    def _wgrad_later(self, launch, dy, x_in):
        if self._wg_pend is None:
            self._wg_pend = []
        self._wg_pend.append((launch, dy, x_in))
        if dy.numel() >= (1 << 22) or len(self._wg_pend) >= self.WGRAD_DEFER:
            self._wgrad_flush()

    def _wgrad_flush(self):
        pend = self._wg_pend
        if not pend:
            return
        with self._wgrad_stream(*[t for j in pend for t in (j[1], j[2])]):
            for j in pend:
                j[0]()
        pend.clear()

    def _final_from(self, p, like):
        if self._grad_bucket_hook is not None:      # (data parallelism: the bucket's weight gradients must be on the side stream before the hook orders behind it)
            self._wgrad_flush()
        self._grads_final_from(p, like)

    def _convblock_bwd(self, cb, tag, da2, saved, need_dx):
        ops = self.ops
        h, y1, st1, em, a1, y2, st2, G = saved[tag]
        dy2 = ops.norm_bwd(y2, da2, G, st2, H.ACT_LRELU, cb.b2.weight.grad, cb.b2.bias.grad, True)
        # weight gradients run underneath the dgrad -> norm_bwd chain (VNet.py)
        self._wgrad_later(lambda: ops.conv3_wgrad(a1, dy2, cb.c2.weight.grad, 1, accumulate=True), dy2, a1)
        _, wd2 = self.conv3_packed((tag, 2), True)
        es = 1.0 / (1.0 - cb.p) if cb.p > 0 else 1.0
        sk = (ops.conv3_nslabs(dy2.shape, cb.cout, 1) if y1 is not None and ops.norm_slabs_ok(G, y2.numel() // (cb.cout * G), cb.cout) else 0)
        if sk > 1:      # deepest level: the dgrad's raw split-K slabs are summed by the norm backward's statistics pass (bcp_norm_bwd_slabs)
            dy1, _ = ops.norm_bwd_slabs(y1, ops.conv3_fwd_raw(dy2, wd2, cb.cout, 1, sk), sk, G, st1, H.ACT_LRELU, cb.b1.weight.grad, cb.b1.bias.grad,
                                        True, elem_mask=em, elem_scale=es)
        else:
            bpart, bnb = None, 0
            if em is None and y1 is not None:       # no dropout between the two convs: the dgrad epilogue leaves b1's backward statistics
                da1, bpart, bnb = ops.conv3_dgrad_bwdstats(dy2, wd2, cb.cout, 1, y1, st1, H.ACT_LRELU, G)
            else:
                da1 = ops.conv3_fwd(dy2, wd2, None, cb.cout, 1)
            if y1 is None and ops.C1_BWD_FUSED:      # the fused first layer, its weight gradient in the same pass (bcp_conv3_c1_norm_bwd_wgrad): no dy1
                ops.conv3_c1_norm_bwd_wgrad(h, cb.c1.weight.data, cb.c1.bias.data, 1, G, st1, da1, H.ACT_LRELU, cb.c1.weight.grad, cb.b1.weight.grad,
                                            cb.b1.bias.grad, True, dw_accumulate=True, elem_mask=em, elem_scale=es)
                return None
            if y1 is None:     # the fused first layer: y1 is recomputed from the block's input (bcp_conv3_c1_norm_bwd)
                dy1 = ops.conv3_c1_norm_bwd(h, cb.c1.weight.data, cb.c1.bias.data, 1, G, st1, da1, H.ACT_LRELU, cb.b1.weight.grad, cb.b1.bias.grad, True,
                                            elem_mask=em, elem_scale=es)
            else:
                dy1 = ops.norm_bwd(y1, da1, G, st1, H.ACT_LRELU, cb.b1.weight.grad, cb.b1.bias.grad, True, elem_mask=em, elem_scale=es,
                                   partial=bpart, nb=bnb)
        if cb.cin == 1:
            self._wgrad_later(lambda: ops.conv3_c1_wgrad(h, dy1, cb.c1.weight.grad, 1, accumulate=True), dy1, h)
        else:
            self._wgrad_later(lambda: ops.conv3_wgrad(h, dy1, cb.c1.weight.grad, 1, accumulate=True), dy1, h)
        if cb.cin == 1 or not need_dx:
            return None
        _, wd1 = self.conv3_packed((tag, 1), True)
        return ops.conv3_fwd(dy1, wd1, None, cb.cin, 1)

    # ------------------------------------------------------------------ schedule
    def _forward_impl(self, xcl, save):
        ops = self.ops
        saved = {} if save else None
        N = xcl.shape[0]
        # torch.cat([skip, up]) without the copy (round 4): encoder block j writes its output straight into the leading channels of the
        # concat buffer its decoder block reads (bcp_norm_fwd out_ld), the pool reads it from there (bcp_maxpool2d_fwd ldx) -- the skip
        # tensor exists once.  Hh, Ww of level j = input >> j; channels FT[j].
        cats = [None] * 4
        Hh, Ww = xcl.shape[2], xcl.shape[3]
        if self.training and self.skip_in_concat:
            cats = [torch.empty((N, 1, Hh >> j, Ww >> j, 2 * FT[j]), dtype=torch.float32, device=xcl.device) for j in range(4)]
        slab = [None if c is None else ops.channel_slab(c, FT[j]) for j, c in enumerate(cats)]
        xs = [self._convblock_fwd(self._enc[0], "e0", xcl, save, saved, out=slab[0])]
        for i in range(1, 5):
            # (the pooled tensor carries xs[-1]'s |max| slots; the concat buffer xs[-1] lives in gets slots of its own, started as a copy)
            pooled = ops.maxpool2d_fwd(xs[-1], concat=cats[i - 1] if (slab[i - 1] is not None and xs[-1] is slab[i - 1]) else None)
            xs.append(self._convblock_fwd(self._enc[i], f"e{i}", pooled, save, saved, out=slab[i] if i < 4 else None))
        h = xs[4]
        for i, (pw, cb, c1, c2) in enumerate(self._up, start=1):
            bp, _ = self.k2_packed(("pw", i), save)
            z = ops.pw_fwd(h, bp, pw.bias.data, c2)
            skip = xs[4 - i]
            if slab[4 - i] is not None and skip is slab[4 - i]:
                cat = cats[4 - i]              # (its |max| slots: a copy of the skip's, made by the pool launch; the upsample max-reduces its half into them)
            else:
                cat = torch.empty((N, 1, skip.shape[2], skip.shape[3], 2 * c2), dtype=torch.float32, device=xcl.device)
                ops.copy_channels(skip, cat, c2, 0, 0, carry_amax=True)      # (+ the |max| of the concat buffer: skip's, then the upsampled half's)
            ops.bilinear2x_fwd(z, cat, c2)
            if save:
                saved[f"pw{i}"] = (h,)
            h = self._convblock_fwd(cb, f"u{i}", cat, save, saved)
        wf, _ = self.conv3_packed(("out", 0), save)
        logits = ops.conv3_fwd(h, wf, self._out.bias.data, self.n_classes, 1)
        if getattr(self, "_want_features", False):
            self._last_feature = h
        if self.training:
            for _ in range(getattr(self, "_groups", 1)):
                self._nbt_tick()
        if save:
            saved["out"] = (h,)
            saved["xs"] = xs
        return logits, saved

    def _grad_stages(self):
        def cbp(cb):
            return [m_.weight for m_ in (cb.c1, cb.b1, cb.c2, cb.b2)] + [m_.bias for m_ in (cb.c1, cb.b1, cb.c2, cb.b2)]
        st = []
        for i in range(4, 0, -1):
            pw, cb = self._up[i - 1][0], self._up[i - 1][1]
            st.append((pw.weight, [pw.weight, pw.bias] + cbp(cb) + ([self._out.weight, self._out.bias] if i == 4 else [])))
        for i in range(4, 0, -1):
            st.append((self._enc[i].c1.weight, cbp(self._enc[i])))
        return st

    def _backward_impl(self, saved, dout):
        ops = self.ops
        dlogits = dout if dout.is_contiguous() else dout.contiguous()
        self.begin_backward()
        (h_last,) = saved["out"]
        xs = saved["xs"]
        self._wg_pend = []

        def out_wgrad():
            ops.conv3_wgrad(h_last, dlogits, self._out.weight.grad, 1, accumulate=True)
            ops.colsum(dlogits, self._out.bias.grad, accumulate=True)
        self._wgrad_later(out_wgrad, dlogits, h_last)
        _, wd = self.conv3_packed(("out", 0), True)
        dh = ops.conv3_fwd(dlogits, wd, None, FT[0], 1)
        skip_grads = {}
        for i in range(4, 0, -1):
            pw, cb, c1, c2 = self._up[i - 1]
            dcat = self._convblock_bwd(cb, f"u{i}", dh, saved, True)          # [N,1,H,W,2*c2]
            skip_grads[4 - i] = (dcat, c2)                                       # first c2 channels belong to xs[4-i]
            dz = ops.bilinear2x_bwd(dcat, c2, c2)
            (h_in,) = saved[f"pw{i}"]
            def pw_wgrad(h_in=h_in, dz=dz, pw=pw):
                ops.k2_wgrad(h_in, dz, pw.weight.grad, H.WG_PW, accumulate=True)
                ops.colsum(dz, pw.bias.grad, accumulate=True)
            self._wgrad_later(pw_wgrad, dz, h_in)
            _, bpd = self.k2_packed(("pw", i), True)
            dh = ops.pw_fwd(dz, bpd, None, c1)
            self._final_from(pw.weight, dz)
        # dh = gradient w.r.t. x4; walk the encoder upwards
        for i in range(4, 0, -1):
            dpool = self._convblock_bwd(self._enc[i], f"e{i}", dh, saved, True)
            dx = torch.empty(tuple(xs[i - 1].shape), dtype=torch.float32, device=dpool.device)
            dcat, c2 = skip_grads[i - 1]
            # pool backward + the decoder-side skip gradient (the leading c2 channels of dcat) in one pass
            ops.maxpool2d_bwd(xs[i - 1], dpool, dx, add=ops.channel_slab(dcat, c2))
            dh = dx
            self._final_from(self._enc[i].c1.weight, dx)
        self._convblock_bwd(self._enc[0], "e0", dh, saved, False)
        self._wgrad_flush()
        self._join_wgrad_stream(dlogits)
        return None


class UNet(UNet_2d):
    """networks/unet.py:148-201 `UNet`: the SAME layers and state_dict keys as `UNet_2d`, but `forward` returns
    `(output, features)` with `features` the decoder's last 16-channel activation (:104-116).  The reference only builds it through
    `net_factory(net_type="unet")` in its offline test / demo scripts (test_ACDC.py:91, under `torch.no_grad()` after `.eval()`);
    no training loop uses the second output, so it is served on the no-grad path only."""

    def forward(self, x, groups=1):
        if self.training and torch.is_grad_enabled() and self._enc[0].c1.weight.requires_grad:
            raise NotImplementedError("UNet (net_factory('unet')): (output, features) is an inference-time interface in BCP; train with "
                                      "BCP_net() / UNet_2d, or call under torch.no_grad()")
        self._want_features, plans, self.use_plans = True, self.use_plans, False      # a replayed pass would not refresh _last_feature
        try:
            out = super().forward(x, groups)
            feat = self._last_feature
        finally:
            self._want_features, self.use_plans = False, plans
            self._last_feature = None
        return out, feat.permute(0, 4, 1, 2, 3).squeeze(2)
