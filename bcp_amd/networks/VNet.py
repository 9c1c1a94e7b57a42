"""3-D V-Net on the MI355X kernels -- drop-in for the reference's networks/VNet.py:VNet (LA, BatchNorm3d +
Dropout3d) and pancreas/Vnet.py:VNet (InstanceNorm3d, `branchs` head).

Same constructor arguments, same `state_dict()` keys / shapes (259 / 60), same `parameters()` order,
same call result: `(out_seg, features)` for the LA net, `[out]` for the pancreas net.  What differs is
what runs: every layer is a call into libbcp_hip.so (NDHWC fp32, MFMA implicit-GEMM convs, fused
norm+act(+dropout,+skip) streams), scheduled by `_forward_impl` / `_backward_impl` below; autograd
sees ONE node per call (networks/_hipnet.py:NetFn).

Topology (networks/VNet.py:167-186, 213-239): block_one(1->16) dw block_two(2x32) dw block_three(3x64) dw
block_four(3x128) dw block_five(3x256) [Dropout3d] up+x4 block_six(3x128) up+x3 block_seven(3x64) up+x2
block_eight(2x32) up+x1 block_nine(16) [Dropout3d] out_conv(16->n_classes, 1x1x1).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import hip_ops as H
from ._hipnet import BNP, ConvP, HipNet, Holder, NetFn, Seq, contrastive_heads


class _Layer:
    __slots__ = ("kind", "conv", "bn", "cin", "cout", "name", "skip_push", "skip_pop", "drop")

    def __init__(self, kind, conv, bn, cin, cout, name):
        self.kind, self.conv, self.bn, self.cin, self.cout, self.name = kind, conv, bn, cin, cout, name
        self.skip_push = False   # the INPUT of this (dw) layer is a skip source
        self.skip_pop = False    # this (up) layer adds the matching skip after its ReLU
        self.drop = None         # 'x5' / 'x9': Dropout3d on this layer's output


class VNet(HipNet):
    UP_RECOMPUTE_GRAD = False   # (round 6) the recomputing transposed conv + norm (library option up_recompute) also in forwards that save for a backward pass: measured slower there (both passes of the norm's backward recompute the conv); forwards WITHOUT a backward pass -- the teacher's -- take it wherever the library serves the shape
    WGRAD_DEFER = 2      # (round 6) small layers' weight gradients cross to the side stream in batches of this many layers (1: every layer forks, rounds 2-5)
    fuse_c1 = True       # first layer: conv + norm + ReLU with recompute (bcp_conv3_c1_norm_fwd / _bwd); False: conv -> y -> norm passes
    fuse_head = True     # the 1x1x1 head applies the last conv's norm + ReLU + Dropout3d itself (bcp_pw16_fwd_norm); False: separate apply pass

    def __init__(self, n_channels=3, n_classes=2, n_filters=16, normalization="none", has_dropout=False, has_residual=False,
                 variant="la"):
        super().__init__()
        assert n_filters == 16 and not has_residual, "only the configuration the BCP scripts use is implemented"
        assert n_channels == 1, "the hot path is single-channel (LA / pancreas); see DESIGN.md"
        la = variant == "la"
        if la:
            assert normalization == "batchnorm", "networks/net_factory.py builds the LA V-Net with batchnorm"
        else:
            assert normalization == "instancenorm"
        self.variant = variant
        self.norm = normalization
        self.has_dropout = has_dropout
        self.n_classes = n_classes
        nf = n_filters
        self._layers = []
        bn = la

        def block(owner, name, n, cin, cout, kind="c3"):
            items = []
            for i in range(n):
                ci = cin if i == 0 else cout
                if kind == "c3":
                    w = (cout, ci, 3, 3, 3)
                    fan = ci * 27
                elif kind == "dw":
                    w = (cout, ci, 2, 2, 2)
                    fan = ci * 8
                else:  # ConvTranspose3d weight [Cin, Cout, 2,2,2]; torch's fan_in uses dim 1
                    w = (ci, cout, 2, 2, 2)
                    fan = cout * 8
                conv = ConvP(w, fan, cout)
                items.append((3 * i, conv))
                b = BNP(cout) if bn else None
                if b is not None:
                    b._live = True
                    items.append((3 * i + 1, b))
                k = "c1" if (kind == "c3" and ci == 1) else kind
                self._layers.append(_Layer(k, conv, b, ci, cout, f"{name}.{3 * i}"))
            h = Holder()
            h.conv = Seq(items)
            setattr(owner, name, h)

        enc = Holder() if la else self
        dec = Holder() if la else self
        block(enc, "block_one", 1, n_channels, nf)
        block(enc, "block_one_dw", 1, nf, 2 * nf, "dw")
        block(enc, "block_two", 2, 2 * nf, 2 * nf)
        block(enc, "block_two_dw", 1, 2 * nf, 4 * nf, "dw")
        block(enc, "block_three", 3, 4 * nf, 4 * nf)
        block(enc, "block_three_dw", 1, 4 * nf, 8 * nf, "dw")
        block(enc, "block_four", 3, 8 * nf, 8 * nf)
        block(enc, "block_four_dw", 1, 8 * nf, 16 * nf, "dw")
        block(enc, "block_five", 3, 16 * nf, 16 * nf)
        if la:
            enc.dropout = nn.Dropout3d(p=0.5, inplace=False)  # parameter-free; kept for module-tree parity
        block(dec, "block_five_up", 1, 16 * nf, 8 * nf, "up")
        block(dec, "block_six", 3, 8 * nf, 8 * nf)
        block(dec, "block_six_up", 1, 8 * nf, 4 * nf, "up")
        block(dec, "block_seven", 3, 4 * nf, 4 * nf)
        block(dec, "block_seven_up", 1, 4 * nf, 2 * nf, "up")
        block(dec, "block_eight", 2, 2 * nf, 2 * nf)
        block(dec, "block_eight_up", 1, 2 * nf, nf, "up")
        if la:
            block(dec, "block_nine", 1, nf, nf)
            dec.out_conv = ConvP((n_classes, nf, 1, 1, 1), nf, n_classes)
            dec.dropout = nn.Dropout3d(p=0.5, inplace=False)
            self.encoder, self.decoder = enc, dec
            self.pool = nn.MaxPool3d(3, stride=2)
            contrastive_heads(self, 2)
            object.__setattr__(self, "_out", dec.out_conv)   # alias without a second registration
        else:
            br = Holder()
            block(br, "b0", 1, nf, nf)
            head = ConvP((n_classes, nf, 1, 1, 1), nf, n_classes)
            self.branchs = nn.ModuleList([Seq([(0, br.b0), (1, head)])])
            object.__setattr__(self, "_out", head)
        for L in self._layers:
            if L.kind == "dw":
                L.skip_push = True
            if L.kind == "up":
                L.skip_pop = True
        if la:
            # Dropout3d sites: after block_five's last conv (x5) and after block_nine (x9)
            idx5 = max(i for i, L in enumerate(self._layers) if L.name.startswith("block_five."))
            self._layers[idx5].drop = "x5"
            self._layers[-1].drop = "x9"
        # parameters that take part in the optimiser: everything the forward pass touches
        ids = set()
        for L in self._layers:
            ids.add(id(L.conv.weight)); ids.add(id(L.conv.bias))
            if L.bn is not None:
                ids.add(id(L.bn.weight)); ids.add(id(L.bn.bias))
        ids.add(id(self._out.weight)); ids.add(id(self._out.bias))
        self._opt_param_ids = ids
        for li, L in enumerate(self._layers):
            if L.kind == "c3":
                self.register_conv3(("c3", li), L.conv.weight, 3)
            elif L.kind == "dw":
                self.register_k2(("k2", li), L.conv.weight, L.cin, L.cout, H.PACK_DOWN_FWD, H.PACK_DOWN_DGRAD)
            elif L.kind == "up":
                self.register_k2(("k2", li), L.conv.weight, L.cin, L.cout, H.PACK_UP_FWD, H.PACK_UP_DGRAD)

    # ------------------------------------------------------------------ public call
    pool_features = True      # LA variant: also return pool(x5) as the reference does (networks/VNet.py:286-290)

    def forward(self, input, turnoff_drop=False, groups=1, features=None):
        """groups > 1: `input` holds `groups` consecutive sub-batches that the reference would push through the net in
        separate calls (LA_BCP_train.py:241-242, 252-253); they are normalised separately (grouped BatchNorm) but every
        layer is ONE launch over all of them -- identical results, half the launches, twice the work per launch."""
        x = input
        N = x.shape[0]
        assert x.dim() == 5 and x.shape[1] == 1, "expected [N,1,X,Y,Z]"
        self._ensure_flat()
        self.refresh_weights_version()
        xcl = x.contiguous().view(N, x.shape[2], x.shape[3], x.shape[4], 1)
        self._turnoff_drop = bool(turnoff_drop)
        assert N % groups == 0
        self._groups = int(groups)
        self._feat_out = None
        # `features=False` (the fused step functions of train_step.py, which discard it): no max-pool launch, no copy out of the plan
        self._want_feat = bool(self.pool_features if features is None else features) and self.variant == "la"
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in (self._layers[0].conv.weight,)):
            out = NetFn.apply(xcl, self._layers[0].conv.weight, self)   # eval() is forward-only (validation, test_3d_patch)
        else:
            out, _ = self._run_forward(xcl, False)
        logits = out.permute(0, 4, 1, 2, 3)  # logical [N,C,X,Y,Z], channels_last_3d strides
        if self.variant == "la":
            # second value: pool(x5) [N,256,3,3,2] at the LA size (networks/VNet.py:286-290; dead in every train script, LA_BCP_train.py:158,252;
            # forward only -- no gradient flows into it).  None when the deepest level is smaller than the 3x3x3 window (the reference raises there).
            f = self._feat_out
            return logits, (None if f is None else f.permute(0, 4, 1, 2, 3))
        return [logits]

    # ------------------------------------------------------------------ schedule
    def _chan_scale(self, L, N, dev):
        if L.drop is None or not self.has_dropout or not self.training or getattr(self, "_turnoff_drop", False):
            return None
        if self.drop_masks is not None:
            m = self.drop_masks[L.drop]
            return (m.to(device=dev, dtype=torch.float32) * 2.0).contiguous()
        cs = torch.empty((N, L.cout), dtype=torch.float32, device=dev)
        return self.ops.bernoulli(cs, 0.5, 2.0, self.next_seed())

    def _forward_impl(self, xcl, save):
        ops = self.ops
        N = xcl.shape[0]
        G = getattr(self, "_groups", 1) if self.norm == "batchnorm" else N
        h = xcl
        skips = []
        saved = []
        feat = None
        last = len(self._layers) - 1
        Ll = self._layers[last]
        fuse_head = (self.fuse_head and self.training and Ll.kind == "c3" and Ll.cout == 16 and not Ll.skip_pop and N <= 32)
        for li, L in enumerate(self._layers):
            w, b = L.conv.weight, L.conv.bias
            part, nb = None, 0
            if L.skip_push:
                skips.append(h)
            # deep levels (<= 4096 rows per normalisation group) whose conv runs split-K: the conv leaves its raw slabs and the norm's
            # statistics pass sums them (+ bias) on its way in (bcp_norm_fwd_slabs) -- no slab-sum launch; decided per layer below
            sp = h.shape[1] * h.shape[2] * h.shape[3]
            slabs_ok = (self.training and L.kind == "c3" and not (li == last and fuse_head) and ops.norm_slabs_ok(G, N * sp // G, L.cout))
            small = False
            src, nsl, bsrc = None, 1, None
            fused_c1 = False
            fused_up = False
            if L.kind == "c1":
                fused_c1 = (self.fuse_c1 and self.training and not small and not L.skip_pop and L.drop is None and not getattr(self, "_keep_saved", False)
                            and not (li == last and fuse_head) and ops.conv3_c1_norm_ok(h.shape, 3, G))
                if fused_c1:
                    y = None       # conv + norm + ReLU with recompute (bcp_conv3_c1_norm_fwd): the 16-channel pre-norm tensor is never written
                elif self.training and not small:
                    y, part, nb = ops.conv3_c1_fwd_stats(h, w.data, b.data, 3, G)      # statistics in the epilogue: no pass over the 16-channel y
                else:
                    y = ops.conv3_c1_fwd(h, w.data, b.data, 3)
            elif L.kind == "c3":
                wf, _ = self.conv3_packed(("c3", li), save)
                sk = ops.conv3_nslabs(h.shape, L.cout, 3) if slabs_ok else 0
                if sk > 1:
                    src, nsl, bsrc, small = ops.conv3_fwd_raw(h, wf, L.cout, 3, sk), sk, b.data, True
                elif self.training or L.bn is None:
                    y, part, nb = ops.conv3_fwd_stats(h, wf, b.data, L.cout, 3, G)
                else:
                    y = ops.conv3_fwd(h, wf, b.data, L.cout, 3)          # eval-mode BatchNorm needs no batch statistics
            elif (L.kind == "up" and self.training and L.drop is None and not (li == last and fuse_head) and not getattr(self, "_keep_saved", False)
                  and (not save or self.UP_RECOMPUTE_GRAD) and ops.up_norm_rows(h.shape, L.cout, G) > 0):
                # (round 6) transposed conv + norm + ReLU + skip add with recompute (bcp_up_fwd_norm): the pre-norm tensor is never written
                bp, _ = self.k2_packed(("k2", li), save)
                fused_up, y = True, None
            else:
                kind = 0 if L.kind == "dw" else 1
                bp, _ = self.k2_packed(("k2", li), save)
                # (round 6) the norm's statistics from the GEMM's epilogue where the shape has them: no statistics pass over y
                if self.training and ops.k2_stat_rows(kind, h.shape, L.cout, G) > 0:       # (library option k2_stats = 0: never)
                    y, part, nb = ops.k2_fwd_stats(kind, h, bp, b.data, L.cout, G)
                elif kind == 0:
                    y = ops.down_fwd(h, bp, b.data, L.cout)
                else:
                    y = ops.up_fwd(h, bp, b.data, L.cout)
            res = skips.pop() if L.skip_pop else None
            cs = self._chan_scale(L, N, xcl.device)
            if fused_up:
                bn = L.bn
                a, stats = ops.up_fwd_norm(h, bp, b.data, L.cout, G, *((bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var)
                                                                      if bn is not None else (None,) * 4), H.ACT_RELU, residual=res)
            elif fused_c1:
                bn = L.bn
                a, stats = ops.conv3_c1_norm_fwd(h, w.data, b.data, 3, G, *((bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var)
                                                                              if bn is not None else (None,) * 4), H.ACT_RELU)
            elif small:
                bn = L.bn
                a, stats, y = ops.norm_fwd_slabs(src, nsl, bsrc, G,
                                                 *((bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var) if bn is not None else (None,) * 4),
                                                 H.ACT_RELU, chan_scale=cs, residual=res)
            elif li == last and fuse_head:
                # the head normalises on its way in: block_nine's 16-channel activation is never written (statistics only here)
                bn = L.bn
                a, stats = ops.norm_fwd(y, G, *((bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var) if bn is not None else (None,) * 4),
                                        H.ACT_RELU, chan_scale=cs, partial=part, nb=nb, stats_only=True)
            elif L.bn is not None and not self.training:
                # model.eval(): running statistics, no update (validation / sliding-window inference)
                a, stats = ops.norm_eval(y, L.bn.weight.data, L.bn.bias.data, L.bn.running_mean, L.bn.running_var, H.ACT_RELU, residual=res), None
            elif L.bn is not None:
                a, stats = ops.norm_fwd(y, G, L.bn.weight.data, L.bn.bias.data, L.bn.running_mean, L.bn.running_var, H.ACT_RELU,
                                        chan_scale=cs, residual=res, partial=part, nb=nb)
            else:
                a, stats = ops.norm_fwd(y, G, None, None, None, None, H.ACT_RELU, chan_scale=cs, residual=res, partial=part, nb=nb)
            if save:
                saved.append((h, y, stats, cs, G))
            h = a
            if L.drop == "x5" and getattr(self, "_want_feat", False) and min(a.shape[1:4]) >= 3:
                feat = ops.maxpool3d_k3s2_fwd(a)      # pool(features[4]): the reference's second return value (networks/VNet.py:286-290)
        if self.norm == "batchnorm" and self.training:
            for _ in range(G):
                self._nbt_tick()
        if fuse_head:
            logits = ops.pw16_fwd_norm(y, stats, cs, G, H.ACT_RELU, self._out.weight.data, self._out.bias.data, self.n_classes)
        else:
            logits = ops.pw16_fwd(h, self._out.weight.data, self._out.bias.data, self.n_classes)
        if save:
            saved.append((h,))          # None when the head is fused: the backward recomputes it from saved[last]
        logits._bcp_feat = feat         # rides on the result object (launch plans keep it; _run_forward hands out a copy)
        return logits, saved

    def _grad_stages(self):
        st = []
        for li in range(len(self._layers) - 1, -1, -1):
            L = self._layers[li]
            ps = [L.conv.weight, L.conv.bias] + ([L.bn.weight, L.bn.bias] if L.bn is not None else [])
            if li == len(self._layers) - 1:
                ps += [self._out.weight, self._out.bias]
            st.append((L.conv.weight, ps))
        return st

    def _backward_impl(self, saved, dout):
        ops = self.ops
        G = saved[0][4]
        dlogits = dout if dout.is_contiguous() else dout.contiguous()
        self.begin_backward()
        (h_last,) = saved[-1]
        dy_head = None                   # the fused head went THROUGH the last layer's norm: that layer's dy, no norm_bwd of its own
        if h_last is None:
            Ll = self._layers[-1]
            _, y9, st9, cs9, _ = saved[len(self._layers) - 1]
            if ops.HEAD_BWD_FUSED and not Ll.skip_pop:
                dg9, db9 = (Ll.bn.weight.grad, Ll.bn.bias.grad) if Ll.bn is not None else (None, None)
                dy_head = ops.pw16_bwd_norm_bwd(y9, st9, cs9, G, H.ACT_RELU, dlogits, self._out.weight.data, self._out.weight.grad,
                                                self._out.bias.grad, dg9, db9, norm_accumulate=Ll.bn is not None, accumulate=True)
                dh = None
            else:
                dh = ops.pw16_bwd_norm(y9, st9, cs9, G, H.ACT_RELU, dlogits, self._out.weight.data, self._out.weight.grad, self._out.bias.grad,
                                       accumulate=True)
        else:
            dh = ops.pw16_bwd(h_last, dlogits, self._out.weight.data, self._out.weight.grad, self._out.bias.grad, accumulate=True)
        skip_grads = []
        pend = []

        def flush_wgrads():
            if not pend:
                return
            with self._wgrad_stream(*[t for j in pend for t in (j[2], j[1])]):
                for kind, x_in_, dy_, gw_, acc_, _ in pend:
                    if kind == "c1":
                        ops.conv3_c1_wgrad(x_in_, dy_, gw_, 3, accumulate=acc_)
                    elif kind == "c3":
                        ops.conv3_wgrad(x_in_, dy_, gw_, 3, accumulate=acc_)
                    elif kind == "dw":
                        ops.k2_wgrad(x_in_, dy_, gw_, H.WG_DOWN, accumulate=acc_)
                    else:
                        ops.k2_wgrad(x_in_, dy_, gw_, H.WG_UP, accumulate=acc_)
            for j in pend:       # (data parallelism: the flat gradient buffer is final from this layer's first parameter on -- bucket hook)
                self._grads_final_from(j[5], j[2])
            pend.clear()
        nsl = 1                          # dh is a plain gradient tensor (1) or the raw split-K slabs of the dgrad that produced it (> 1)
        bpart, bnb = None, 0             # backward-statistics partials of THIS layer's norm, left by the dgrad that produced dh
        for li in range(len(self._layers) - 1, -1, -1):
            L = self._layers[li]
            x_in, y, stats, cs, _ = saved[li]
            w = L.conv.weight
            da = dh
            dg, db = (L.bn.weight.grad, L.bn.bias.grad) if L.bn is not None else (None, None)
            if dy_head is not None:
                dy, dy_head = dy_head, None
            elif y is None and L.kind == "up":     # (round 6) the recomputing transposed conv: y = up(x_in) again, inside both passes of the norm's backward
                assert nsl == 1 and cs is None
                bpf, _ = self.k2_packed(("k2", li), True)      # (the FORWARD pack: y is recomputed)
                dy = ops.up_norm_bwd(x_in, bpf, L.conv.bias.data, L.cout, G, stats, da, H.ACT_RELU, dg, db, L.bn is not None)
            elif y is None:     # the fused first layer: its pre-norm tensor is recomputed from the input (bcp_conv3_c1_norm_bwd)
                assert L.kind == "c1" and nsl == 1
                if ops.C1_BWD_FUSED:      # ... and the layer's weight gradient in the same pass: no dy, no launch left behind the main stream's last one
                    ops.conv3_c1_norm_bwd_wgrad(x_in, w.data, L.conv.bias.data, 3, G, stats, da, H.ACT_RELU, w.grad, dg, db, L.bn is not None,
                                                dw_accumulate=True)
                    flush_wgrads()
                    self._grads_final_from(w, da)
                    break
                dy = ops.conv3_c1_norm_bwd(x_in, w.data, L.conv.bias.data, 3, G, stats, da, H.ACT_RELU, dg, db, L.bn is not None)
            elif nsl > 1:
                # deep levels: dh is the raw split-K slabs of the dgrad that produced it; the backward-statistics pass sums them (bcp_norm_bwd_slabs)
                dy, da = ops.norm_bwd_slabs(y, da, nsl, G, stats, H.ACT_RELU, dg, db, L.bn is not None, chan_scale=cs)
            else:
                dy = ops.norm_bwd(y, da, G, stats, H.ACT_RELU, dg, db, L.bn is not None, chan_scale=cs, partial=bpart, nb=bnb)
            nsl, bpart, bnb = 1, None, 0
            if L.skip_pop:
                skip_grads.append(da)       # d(out)/d(skip) = identity: the skip source gets `da` itself
            gw, acc = w.grad, True
            # conv biases feed a norm: their gradient is identically zero (DESIGN.md "bias gradients"); the flat
            # gradient buffer was cleared by begin_backward(), nothing to add.
            # The weight gradient of a layer has no consumer inside the backward pass: it runs on a side stream underneath
            # the dgrad -> norm_bwd critical path (the deep levels' kernels are too small to fill 256 CUs on their own).
            # (round 6) every fork onto the side stream is an event record between two kernels of the MAIN stream -- a 5-7 us gap in front of
            # the dgrad that follows (kernel trace, gpurun_out/r06_s31) -- so the small layers' weight gradients go over in batches of
            # WGRAD_DEFER layers behind ONE fork; the large layers (>= 2^22 elements of dy) fork at once, as before
            pend.append((L.kind, x_in, dy, gw, acc, w))
            if dy.numel() >= (1 << 22) or len(pend) >= self.WGRAD_DEFER:
                flush_wgrads()
            if L.kind == "c1":
                dh = None
            elif L.kind == "c3":
                _, wd = self.conv3_packed(("c3", li), True)
                # the consumer of this dgrad is the previous layer's norm backward: at the deep levels it takes the raw split-K slabs
                # (no slab-sum launch); the previous layer must have a pre-norm tensor of its own (not the recomputing first layer)
                sk = (ops.conv3_nslabs(dy.shape, L.cin, 3) if li > 0 and saved[li - 1][1] is not None
                      and ops.norm_slabs_ok(G, x_in.numel() // (L.cin * G), L.cin) else 0)
                if sk > 1:
                    dh, nsl = ops.conv3_fwd_raw(dy, wd, L.cin, 3, sk), sk
                elif li > 0 and saved[li - 1][3] is None and saved[li - 1][1] is not None:      # (a recomputing layer in front keeps no pre-norm tensor)
                    # conv -> conv edge without a dropout epilogue: the dgrad epilogue leaves the previous norm layer's backward statistics
                    # (bcp_conv3_dgrad_bwdstats; a plain dgrad when the shape is not served)
                    dh, bpart, bnb = ops.conv3_dgrad_bwdstats(dy, wd, L.cin, 3, saved[li - 1][1], saved[li - 1][2], H.ACT_RELU, G)
                else:
                    dh = ops.conv3_fwd(dy, wd, None, L.cin, 3)
            else:
                kind = 0 if L.kind == "dw" else 1
                _, bp = self.k2_packed(("k2", li), True)
                sg = skip_grads.pop() if kind == 0 else None      # down conv: x_in is a skip source, the decoder-side gradient is joined in place
                # (round 6) the layer in front has a pre-norm tensor of its own and no dropout epilogue: this dgrad's epilogue leaves its norm's
                # backward statistics (bcp_down_dgrad_bwdstats / bcp_up_dgrad_bwdstats) where the shape is served -- no k_col_partial<1> pass over (y, da)
                prev = saved[li - 1] if li > 0 else None
                if prev is not None and prev[1] is not None and prev[3] is None and ops.k2_bwdstat_rows(kind, dy.shape, L.cin, G) > 0:
                    dh, bpart, bnb = ops.k2_dgrad_bwdstats(kind, dy, bp, L.cin, prev[1], prev[2], H.ACT_RELU, G, out=sg, accumulate=sg is not None)
                elif kind == 0:
                    dh = ops.down_dgrad(dy, bp, L.cin, out=sg, accumulate=True)
                else:
                    dh = ops.up_dgrad(dy, bp, L.cin)
        flush_wgrads()
        self._join_wgrad_stream(dlogits)
        return None
