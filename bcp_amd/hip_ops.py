"""Tensor-level wrappers over the C ABI (include/bcp_hip.h).  PyTorch tensors are containers only:
every op here is one call into libbcp_hip.so on the tensor's data_ptr() and the current stream.

Physical layout convention ("cl" tensors): activations are contiguous [N, D, H, W, C] (2-D: D == 1),
labels / masks contiguous uint8 [N, D, H, W].  The logical NCDHW view the reference's scripts see is
`cl.permute(0, 4, 1, 2, 3)` (torch channels_last_3d strides) -- a view, never a copy.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
LOSS_LA, LOSS_ACDC = 0, 1
PACK_DOWN_FWD, PACK_DOWN_DGRAD, PACK_UP_FWD, PACK_UP_DGRAD, PACK_PW_FWD, PACK_PW_DGRAD = range(6)
WG_DOWN, WG_UP, WG_PW = range(3)
CAST_I64_U8, CAST_F32_U8, CAST_U8_F32, CAST_U8_I64 = range(4)


AMAX_FLOATS, AMAX_STRIDE = 1024, 32      # csrc/common.h kAmaxFloats / kAmaxStride


def amax_slots(value, device):
    """a |max| slot buffer holding `value` (tests, tools: tensors that did not come out of a norm pass)"""
    a = torch.zeros(AMAX_FLOATS, dtype=torch.float32, device=device)
    a[0] = value
    return a


def amax_value(a):
    """the |max| a slot buffer stands for"""
    v = a[::AMAX_STRIDE]
    return float("nan") if bool(torch.isnan(v).any()) else float(v.max())


def _p(t):
    # plain int: the bound functions declare c_void_p argtypes, ctypes converts (an explicit c_void_p object per pointer
    # was 10 % of the host time of a step)
    return None if t is None else t.data_ptr()


class SeedMask:
    """An elementwise-Dropout keep mask that exists only as its 64-bit seed in device memory (round 4): the norm kernels evaluate the keep
    bit of element i from the seed -- the bit `Ops.bernoulli` would have written at mask[i] -- in the forward AND the backward pass of the
    layer, so there is no mask tensor and no launch for it.  Pass it wherever a wrapper takes `elem_mask=`.  `ptr`: device address of the
    seed (a launch plan's seed slot, refilled before every replay, or a one-element tensor held here)."""
    __slots__ = ("ptr", "p_keep", "shape", "_hold")

    def __init__(self, ptr, p_keep, shape, hold=None):
        self.ptr, self.p_keep, self.shape, self._hold = int(ptr), float(p_keep), tuple(shape), hold


class Ops:
    """All kernels, bound to one library handle.  `Ops.product()` is the only constructor product
    code uses; tests build `Ops(Binding(emu_path), allow_cpu=True)` to run the same wrappers on the
    host simulator."""

    _product = None

    def __init__(self, binding: _lib.Binding, allow_cpu: bool = False):
        self.b = binding
        self.allow_cpu = allow_cpu
        self._ws = {}
        self._wsz = {}
        self._box_cache = {}
        self._rec_plan = None     # set by bcp_amd.plan.recording

    @classmethod
    def product(cls) -> "Ops":
        if cls._product is None:
            cls._product = cls(_lib.product(), allow_cpu=False)
        return cls._product

    # ------------------------------------------------------------------ helpers
    def set_option(self, name, value=""):
        """library tuning / test switch (bcp_set_option); cached shape queries depend on the options: drop them"""
        self.b.set_option(name, value)
        self._wsz.clear()
        from . import plan
        plan.invalidate_all()      # recorded launch plans carry kernel choices and workspace sizes of the old options

    @staticmethod
    def channel_slab(buf, Cc):
        """the first Cc channels of a channels-last buffer [..., ld] as a tensor view (row stride ld): what an encoder block of the U-Net
        writes its output into when the buffer is the decoder's concat buffer"""
        return buf[..., :Cc]

    @staticmethod
    def row_stride(t):
        """row stride (floats) of a channels-last tensor that is either contiguous or the leading channels of a contiguous buffer"""
        ld = t.stride(-2) if t.dim() >= 2 else t.shape[-1]
        exp = ld
        for d in range(t.dim() - 2, -1, -1):
            if t.shape[d] != 1 and t.stride(d) != exp:
                raise _lib.BcpError("HIP op needs a channels-last tensor or the leading channels of one")
            exp *= t.shape[d]
        if t.stride(-1) != 1 or ld < t.shape[-1]:
            raise _lib.BcpError("HIP op needs unit channel stride")
        return int(ld)

    def _chk_rows(self, t):
        """like _chk, for the arguments that may be channel slabs (row_stride)"""
        if not self.allow_cpu and not t.is_cuda:
            raise _lib.BcpError("HIP op called with a CPU tensor: the product path has no CPU fallback")
        return self.row_stride(t)

    def _chk(self, *ts):
        for t in ts:
            if t is None:
                continue
            if not self.allow_cpu and not t.is_cuda:
                raise _lib.BcpError("HIP op called with a CPU tensor: the product path has no CPU fallback")
            if not t.is_contiguous():
                raise _lib.BcpError("HIP op needs contiguous tensors (physical NDHWC)")

    def stream(self, t):
        if t.is_cuda:
            return torch._C._cuda_getCurrentRawStream(t.device.index)
        return None

    @staticmethod
    def _mask_split(elem_mask):
        """elem_mask= of the norm wrappers: a uint8 tensor, a SeedMask or None -> (tensor | None, seed address | None, p_keep)"""
        if isinstance(elem_mask, SeedMask):
            return None, elem_mask.ptr, elem_mask.p_keep
        return elem_mask, None, 0.0

    def seed_mask(self, shape, p_keep, seed, like):
        """the Dropout keep mask of a tensor of `shape` under `seed`, as a SeedMask (see there).  Inside a recorded pass the seed lives in
        the plan's device table (one more slot; refilled before every replay like the bernoulli launches' seeds)"""
        pl = self._rec_plan
        if pl is not None:
            slot = pl.seed_slot(like.device)
            fn, _ = self.b._fns["bcp_store_u64"]
            arr = (C.c_ulonglong * 1)(int(seed) & 0xFFFFFFFFFFFFFFFF)
            rc = fn(slot, 1, arr, self.stream(like))
            if rc:
                raise _lib.BcpError(f"bcp_store_u64 failed ({rc}): {self.b.last_error()}")
            return SeedMask(slot, p_keep, shape)
        t = torch.empty(1, dtype=torch.int64, device=like.device)
        self.store_u64(t, [seed], like)
        return SeedMask(t.data_ptr(), p_keep, shape, hold=t)

    def _ws_bytes(self, fn, *args):
        """size / shape queries: one ctypes round trip per distinct (options epoch, shape).  Several answers depend on the library's
        options (split-K slab counts, fused-statistics rows): the binding counts its set_option calls, so a query cached under other
        options -- also when someone went through Binding.set_option directly -- is never reused"""
        key = (self.b.options_epoch, fn) + args
        v = self._wsz.get(key)
        if v is None:
            v = self._wsz[key] = int(self.b.call(fn, *args))
        return v

    def workspace(self, key, nbytes, like):
        """grow-only scratch buffer per (key, device, stream): two streams never share scratch"""
        if self._rec_plan is not None:
            # a recorded pass owns its scratch: the shared grow-only buffers may be re-allocated later by another shape, and a
            # plan's pointers must stay valid for its whole life (torch.empty is the recording's capturing version)
            return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=like.device)
        k = (key, like.device, torch._C._cuda_getCurrentRawStream(like.device.index) if like.is_cuda else 0)
        w = self._ws.get(k)
        if w is None or w.numel() < nbytes:
            w = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=like.device)
            self._ws[k] = w
        return w

    # ---- per-tensor |max| (round 4): the norm apply pass that writes an activation leaves max |a| in a device float riding on the tensor
    # object (`_bcp_amax`); a conv that reads the tensor hands it to the library (x_amax: power-of-two pre-scale of the fp16 planes).
    # Views / slices do not carry the attribute: those launches take the three-plane bf16 kernels, which need no scale.
    AMAX = True

    # AMAX_BWD (round 5): False keeps the two-plane fp16 instances to the FORWARD operands -- the norm backward passes then leave
    # no |max| for dy, and every dgrad / weight-gradient launch that reads dy takes the three-plane bf16 kernels (no pre-scale, no range
    # assumption).  Default True: gradient tensors are heavy-tailed, but the planes' error is ABSOLUTE (2^-22 of the tensor's |max| per
    # element, the same as the forward operands'), and a 1000-step training run on the device shows the two settings' gradients -- same
    # forward bits, same activation patterns -- apart by 1.9e-6 .. 3.3e-6 rel-L2 (worst parameter tensor; median 1.3e-6) at steps 0, 10, 100,
    # 300, 600 and 999, with no trend (tests/net_checks.py check_fp16_backward_long_run, asserted <= 2e-5; gpurun_out/r05_s14/pytest.txt).
    AMAX_BWD = True

    def _amax_slot(self, out, backward=False):
        if out is None or not self.AMAX or (backward and not self.AMAX_BWD):
            return None
        a = torch.empty(AMAX_FLOATS, dtype=torch.float32, device=out.device)      # 32 slots, one per 128-byte line (csrc/common.h)
        out._bcp_amax = a
        return a

    # debug mode (BCP_AMAX_CHECK=1, or Ops.AMAX_CHECK = True): every launch that is handed a tensor's |max| slots first recomputes
    # max |x| (a host synchronisation per launch: tests and triage only) and fails when the slots promise LESS than the tensor holds --
    # a stale attribute (an in-place write after the producing pass, an attribute copied onto another tensor) would overflow the fp16
    # planes silently.  A NaN on either side passes: the kernels then take scale 1 and forward the NaN.
    AMAX_CHECK = os.environ.get("BCP_AMAX_CHECK", "0") == "1"

    def _amax_of(self, x):
        """the |max| slots of a tensor a launch READS as a conv operand.  (Slots a launch WRITES into -- bilinear2x_fwd's concat buffer, whose
        upsampled half is still torch.empty memory -- are fetched with a plain getattr: checking them against uninitialised contents fired
        spuriously)  The debug check is a host synchronisation and stays out of plan recording / graph capture"""
        a = getattr(x, "_bcp_amax", None)
        if a is not None and self.AMAX_CHECK and self.b._rec is None:
            have, want = amax_value(a), float(x.detach().abs().max())
            if have == have and want == want and want > have:
                raise _lib.BcpError(f"|max| slots of a {tuple(x.shape)} tensor promise {have:.9g} but the tensor holds {want:.9g}: the "
                                    "attribute is stale (in-place write after the producing pass?)")
        return a

    @staticmethod
    def _no_amax(out):
        """an op WITHOUT |max| slots of its own has written into `out`: whatever slots the tensor carried describe its old contents (ADVICE
        r04: an attribute that outlives an in-place write would overflow the fp16 planes of the next reader silently) -- drop them; the
        next conv that reads `out` then takes the three-plane bf16 instance, which needs no |max|"""
        if out is not None and getattr(out, "_bcp_amax", None) is not None:
            out._bcp_amax = None
        return out

    def box_arg(self, box6):
        key = tuple(int(v) for v in box6)
        a = self._box_cache.get(key)
        if a is None:
            a = (C.c_int * 6)(*key)
            if len(self._box_cache) > 4096:
                self._box_cache.clear()
            self._box_cache[key] = a
        return a

    # ------------------------------------------------------------------ BCP ops
    def mix_box(self, a, b, box6, out=None):
        """out = a outside the box, b inside.  a, b: [N,D,H,W,C] float32."""
        self._chk(a, b)
        N, D, H, W, Cc = a.shape
        if out is None:
            out = torch.empty_like(a)
        self.b.call("bcp_mix_box", _p(a), _p(b), _p(out), N, D, H, W, Cc, self.box_arg(box6), self.stream(a))
        return out

    def plabel_bin(self, logits, thres=0.5):
        self._chk(logits)
        N, D, H, W, Cc = logits.shape
        assert Cc == 2
        out = torch.empty((N, D, H, W), dtype=torch.uint8, device=logits.device)
        self.b.call("bcp_plabel_bin", _p(logits), _p(out), N * D * H * W, float(thres), self.stream(logits))
        return out

    def plabel_argmax4(self, logits):
        self._chk(logits)
        N, D, H, W, Cc = logits.shape
        assert Cc == 4
        out = torch.empty((N, D, H, W), dtype=torch.uint8, device=logits.device)
        self.b.call("bcp_plabel_argmax4", _p(logits), _p(out), N * D * H * W, self.stream(logits))
        return out

    def cc_largest(self, seg, nclass=1, connectivity=3, want_f32=False):
        """seg uint8 [N,D,H,W] -> (uint8 same shape, optional float32 copy)"""
        self._chk(seg)
        N, D, H, W = seg.shape
        nbytes = self._ws_bytes("bcp_cc_workspace_bytes", N, D, H, W, nclass)
        ws = self.workspace("cc", nbytes, seg)
        out = torch.empty_like(seg)
        outf = torch.empty(seg.shape, dtype=torch.float32, device=seg.device) if want_f32 else None
        self.b.call("bcp_cc_largest", _p(seg), _p(out), _p(outf), N, D, H, W, nclass, connectivity, _p(ws), self.stream(seg))
        return (out, outf) if want_f32 else out

    def plabel_cc_largest(self, logits, thres=0.5, connectivity=3, want_f32=False, want_seg=False):
        """(round 6) channel-last logits [N,D,H,W,C] -> largest-CC-filtered pseudo-label uint8 [N,D,H,W]: plabel_bin (C = 2, one class) or
        plabel_argmax4 (C = 4, three classes) + cc_largest as one chain (bcp_plabel_cc_largest: the first CC kernel labels from the
        logits itself) -- the same bits as the two calls"""
        self._chk(logits)
        N, D, H, W, Cc = logits.shape
        assert Cc in (2, 4)
        nclass = 1 if Cc == 2 else 3
        ws = self.workspace("cc", self._ws_bytes("bcp_cc_workspace_bytes", N, D, H, W, nclass), logits)
        seg = torch.empty((N, D, H, W), dtype=torch.uint8, device=logits.device)
        out = torch.empty_like(seg)
        outf = torch.empty(seg.shape, dtype=torch.float32, device=seg.device) if want_f32 else None
        self.b.call("bcp_plabel_cc_largest", _p(logits), Cc, float(thres), _p(seg), _p(out), _p(outf), N, D, H, W, nclass, connectivity, _p(ws),
                    self.stream(logits))
        res = (out, outf) if want_f32 else out
        return (res, seg) if want_seg else res

    def mixloss_fwd(self, logits, img_l, patch_l, box6, flavour, w_img, w_patch, mask=None, prev=None, total=None):
        """-> (out3 float32[3] on device, workspace tensor to hand to mixloss_bwd).  prev + total: the step's second call hands in the
        first call's out3 and a float32[1] that receives the step's total loss (the reference's sum order, bcp_hip.h)"""
        self._chk(logits, img_l, patch_l, mask, prev, total)
        N, D, H, W, Cc = logits.shape
        nbytes = self._ws_bytes("bcp_mixloss_workspace_bytes", N, Cc)
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=logits.device)  # kept alive for backward
        out3 = torch.empty(3, dtype=torch.float32, device=logits.device)
        self.b.call("bcp_mixloss_fwd", _p(logits), _p(img_l), _p(patch_l), _p(mask), self.box_arg(box6), N, D, H, W, Cc, flavour,
                    float(w_img), float(w_patch), _p(ws), _p(out3), _p(prev), _p(total), self.stream(logits))
        return out3, ws

    # utils/BCP_utils.mix_loss_pair: both mix_loss calls of a step as ONE launch pair (bcp_mixloss_pair_fwd / _bwd); False
    # (BCP_MIXLOSS_PAIR=0, a measurement switch): two calls each way, the second one forming the step's total
    MIXLOSS_PAIR = os.environ.get("BCP_MIXLOSS_PAIR", "1") != "0"

    def mixloss_pair_fwd(self, logits, img_l, patch_l, img_l2, patch_l2, box6, flavour, w1, w2, mask=None):
        """logits [2N, ...]: call 1 on the first N samples, call 2 on the rest -> (out6 float32[2, 3], total float32[1], workspace)"""
        self._chk(logits, img_l, patch_l, img_l2, patch_l2, mask)
        N2, D, H, W, Cc = logits.shape
        N = N2 // 2
        ws = torch.empty(int(self._ws_bytes("bcp_mixloss_pair_workspace_bytes", N, Cc)), dtype=torch.uint8, device=logits.device)  # kept alive for backward
        out6 = torch.empty((2, 3), dtype=torch.float32, device=logits.device)
        total = torch.empty(1, dtype=torch.float32, device=logits.device)
        self.b.call("bcp_mixloss_pair_fwd", _p(logits), _p(img_l), _p(patch_l), _p(img_l2), _p(patch_l2), _p(mask), self.box_arg(box6), N, D, H, W, Cc,
                    flavour, float(w1[0]), float(w1[1]), float(w2[0]), float(w2[1]), _p(ws), _p(out6), _p(total), self.stream(logits))
        return out6, total, ws

    def mixloss_pair_bwd(self, logits, img_l, patch_l, img_l2, patch_l2, box6, flavour, ws, g_dice, g_ce, mask=None, g_dev=None, out=None):
        self._chk(logits, img_l, patch_l, img_l2, patch_l2, mask, out, g_dev)
        N2, D, H, W, Cc = logits.shape
        dlogits = torch.empty_like(logits) if out is None else out
        self.b.call("bcp_mixloss_pair_bwd", _p(logits), _p(img_l), _p(patch_l), _p(img_l2), _p(patch_l2), _p(mask), self.box_arg(box6), N2 // 2, D, H, W,
                    Cc, flavour, _p(ws), float(g_dice), float(g_ce), _p(g_dev), 0 if g_dev is None else int(g_dev.numel()), _p(dlogits),
                    self.stream(logits))
        return dlogits

    def mixloss_bwd(self, logits, img_l, patch_l, box6, flavour, ws, g_dice, g_ce, mask=None, g_dev=None, out=None):
        """g_dev: device float32 of upstream gradients -- two elements {dice, ce} or ONE for both terms"""
        self._chk(logits, img_l, patch_l, mask, out, g_dev)
        N, D, H, W, Cc = logits.shape
        dlogits = torch.empty_like(logits) if out is None else out
        self.b.call("bcp_mixloss_bwd", _p(logits), _p(img_l), _p(patch_l), _p(mask), self.box_arg(box6), N, D, H, W, Cc, flavour,
                    _p(ws), float(g_dice), float(g_ce), _p(g_dev), 0 if g_dev is None else int(g_dev.numel()), _p(dlogits), self.stream(logits))
        return dlogits

    # ------------------------------------------------------------------ DiceLoss class on probabilities (utils/losses.py:113-134)
    def _dice_prob_args(self, probs, target, mask, mode, box6):
        """probs: logical [N,C,*sp] float32, any DENSE layout whose spatial dims are jointly contiguous"""
        N, Cc = probs.shape[0], probs.shape[1]
        sp = tuple(probs.shape[2:])
        V = 1
        for e in sp:
            V *= e
        st = probs.stride()
        vs = st[-1]
        exp = vs
        for e, s_ in zip(reversed(sp), reversed(st[2:])):      # spatial dims must collapse into one index with stride vs
            if e != 1 and s_ != exp:
                raise _lib.BcpError("DiceLoss: probabilities must be dense (NCHW- or NHWC-contiguous)")
            exp *= e
        D, H, W = (1,) * (3 - len(sp)) + sp
        if not self.allow_cpu and not probs.is_cuda:
            raise _lib.BcpError("HIP op called with a CPU tensor: the product path has no CPU fallback")
        self._chk(target, mask)
        return (_p(probs), int(st[1]), int(vs), int(st[0]), _p(target), _p(mask), int(mode),
                self.box_arg(box6 if box6 is not None else (0,) * 6), N, D, H, W, Cc)

    def dice_prob_fwd(self, probs, target, mask=None, mode=0, box6=None, weight=None):
        """-> (out float32[1 + C] on device {loss, class-wise dice}, workspace for dice_prob_bwd)"""
        args = self._dice_prob_args(probs, target, mask, mode, box6)
        Cc = probs.shape[1]
        ws = torch.empty(self._ws_bytes("bcp_dice_prob_workspace_bytes", Cc), dtype=torch.uint8, device=probs.device)
        out = torch.empty(1 + Cc, dtype=torch.float32, device=probs.device)
        w = None if weight is None else (C.c_float * Cc)(*[float(v) for v in weight])
        self.b.call("bcp_dice_prob_fwd", *args, w, _p(ws), _p(out), self.stream(probs))
        return out, ws

    def dice_prob_bwd(self, probs, target, ws, mask=None, mode=0, box6=None, g_dev=None, g=1.0):
        args = self._dice_prob_args(probs, target, mask, mode, box6)
        d = torch.empty_strided(probs.shape, probs.stride(), dtype=torch.float32, device=probs.device)
        self.b.call("bcp_dice_prob_bwd", *args, _p(ws), _p(g_dev), float(g), _p(d), self.stream(probs))
        return d

    # ------------------------------------------------------------------ norm
    def norm_fwd(self, y, G, gamma, beta, rmean, rvar, act, out=None, chan_scale=None, elem_mask=None, elem_scale=1.0,
                 residual=None, momentum=0.1, eps=1e-5, partial=None, nb=0, stats_only=False):
        """y [N,D,H,W,C] -> (a, stats[5,G,C]).  G = 1: BatchNorm; G = N (no affine): InstanceNorm; G > 1 with affine:
        G consecutive BatchNorm calls in one launch.  stats_only: statistics (and the running-statistics update) without the
        apply pass -- the consumer normalises on its way in (pw16_fwd_norm); returns (None, stats)."""
        elem_mask, em_seed, em_keep = self._mask_split(elem_mask)
        self._chk(y, gamma, beta, rmean, rvar, chan_scale, elem_mask, residual)
        N = y.shape[0]
        Cc = y.shape[-1]
        rows = y.numel() // Cc
        rpg = rows // G
        rps = rows // N
        nbytes = self._ws_bytes("bcp_norm_workspace_bytes", G, rpg, Cc)
        ws = self.workspace("norm", nbytes, y)
        stats = torch.empty((5, G, Cc), dtype=torch.float32, device=y.device)
        out_ld = 0
        if stats_only:
            assert out is None and residual is None and elem_mask is None and em_seed is None
        elif out is None:
            out = torch.empty_like(y)
        else:
            out_ld = self._chk_rows(out)        # out may be the leading channels of a wider buffer (channel_slab: the U-Net's concat)
        amax = self._amax_slot(out)
        self.b.call("bcp_norm_fwd", _p(y), G, rpg, Cc, _p(gamma), _p(beta), _p(rmean), _p(rvar), float(momentum), float(eps), act,
                    _p(chan_scale), rps, _p(elem_mask), float(elem_scale), em_seed, em_keep, _p(residual), _p(stats), _p(ws), _p(partial), int(nb),
                    _p(out), out_ld, _p(amax), self.stream(y))
        return out, stats

    def norm_eval(self, y, gamma, beta, rmean, rvar, act, residual=None, eps=1e-5, out=None):
        """eval-mode BatchNorm (running statistics, no update) + activation [+ residual]"""
        self._chk(y, gamma, beta, rmean, rvar, residual)
        Cc = y.shape[-1]
        if out is None:
            out = torch.empty_like(y)
        self.b.call("bcp_norm_eval", _p(y), y.numel() // Cc, Cc, _p(gamma), _p(beta), _p(rmean), _p(rvar), float(eps), act, _p(residual),
                    _p(out), self.stream(y))
        return out

    def sw_accumulate(self, logits_patch, score, cnt, origin, cls=1):
        """logits_patch [px,py,pz,C] (one patch), score / cnt [X,Y,Z] float32: score[origin + ijk] += softmax(...)[cls], cnt += 1"""
        self._chk(logits_patch, score, cnt)
        px, py, pz, Cc = logits_patch.shape
        X, Y, Z = score.shape
        self.b.call("bcp_sw_accumulate", _p(logits_patch), _p(score), _p(cnt), X, Y, Z, px, py, pz, int(origin[0]), int(origin[1]),
                    int(origin[2]), Cc, int(cls), self.stream(score))

    def sw_finish(self, score, cnt, thres=0.5):
        """score /= cnt in place; -> uint8 label map (score > thres)"""
        self._chk(score, cnt)
        label = torch.empty(score.shape, dtype=torch.uint8, device=score.device)
        self.b.call("bcp_sw_finish", _p(score), _p(cnt), _p(label), score.numel(), float(thres), self.stream(score))
        return label

    def overlap_counts(self, pred, gt, cls=0):
        """-> int64[3] device tensor {|A & B|, |A|, |B|} of two uint8 maps (A = pred != 0 / == cls, B = gt likewise)"""
        self._chk(pred, gt)
        counts = torch.empty(3, dtype=torch.int64, device=pred.device)
        self.b.call("bcp_overlap_counts", _p(pred), _p(gt), pred.numel(), int(cls), _p(counts), self.stream(pred))
        return counts

    def crop_rotflip(self, src, patch, k, flip_axis, pads, origin):
        """src [n0,n1,n2] float32 or uint8 -> [P0,P1,P2]: RandomRotFlip + RandomCrop (dataloaders/dataset.py) as one gather"""
        self._chk(src)
        assert src.dtype in (torch.float32, torch.uint8)
        dst = torch.empty(tuple(patch), dtype=src.dtype, device=src.device)
        n0, n1, n2 = src.shape
        self.b.call("bcp_crop_rotflip", _p(src), _p(dst), src.element_size(), n0, n1, n2, int(k), int(flip_axis), int(pads[0]), int(pads[1]),
                    int(pads[2]), int(origin[0]), int(origin[1]), int(origin[2]), int(patch[0]), int(patch[1]), int(patch[2]), self.stream(src))
        return dst

    def acdc_augment(self, src, out_hw, mode, k=0, flip_axis=0, affine6=None):
        """RandomGenerator's data movement (dataset.py:69-88) for one 2-D tensor: src [H,W] float32 / uint8 -> [OH,OW]"""
        self._chk(src)
        Hh, Ww = src.shape
        dst = torch.empty((int(out_hw[0]), int(out_hw[1])), dtype=src.dtype, device=src.device)
        aff = (C.c_double * 6)(*[float(v) for v in affine6]) if affine6 is not None else None
        self.b.call("bcp_acdc_augment", _p(src), _p(dst), src.element_size(), Hh, Ww, int(mode), int(k), int(flip_axis), aff,
                    int(out_hw[0]), int(out_hw[1]), self.stream(src))
        return dst

    def norm_bwd(self, y, da, G, stats, act, dgamma=None, dbeta=None, accumulate=False, chan_scale=None, elem_mask=None,
                 elem_scale=1.0, out=None, partial=None, nb=0):
        elem_mask, em_seed, em_keep = self._mask_split(elem_mask)
        self._chk(y, da, stats, dgamma, dbeta, chan_scale, elem_mask)
        N = y.shape[0]
        Cc = y.shape[-1]
        rows = y.numel() // Cc
        rpg = rows // G
        rps = rows // N
        nbytes = self._ws_bytes("bcp_norm_workspace_bytes", G, rpg, Cc)
        ws = self.workspace("norm", nbytes, y)
        if out is None:
            out = torch.empty_like(y)
        self.b.call("bcp_norm_bwd", _p(y), _p(da), G, rpg, Cc, _p(stats), act, _p(chan_scale), rps, _p(elem_mask), float(elem_scale),
                    em_seed, em_keep, _p(dgamma), _p(dbeta), int(bool(accumulate)), _p(ws), _p(partial), int(nb), _p(out), _p(self._amax_slot(out, backward=True)), self.stream(y))
        return out


    # ------------------------------------------------------------------ deep levels: the norm takes the conv's raw split-K slabs
    def norm_slabs_ok(self, G, rows_per_group, Cc):
        """True when bcp_norm_fwd_slabs / bcp_norm_bwd_slabs serve this shape (rows per group <= 4096, option norm_slabs)"""
        return bool(self._ws_bytes("bcp_norm_slabs_ok", int(G), int(rows_per_group), int(Cc)))

    def conv3_nslabs(self, xshape, Cout, KD):
        """split-K slabs bcp_conv3_fwd_raw writes for this shape; 0: not served in raw mode"""
        N, D, H, W, Cin = xshape
        return self._ws_bytes("bcp_conv3_fwd_nslabs", int(N), int(D), int(H), int(W), int(Cin), int(Cout), int(KD))

    def _note_pack_section(self, wp):
        """after an eager / recorded forward or dgrad launch: tell the network that owns the packed weight `wp` which of its sections the
        launch read (bcp_conv3_last_section) -- what lets it pack partially in front of replays (networks/_hipnet.py PACK_PARTIAL)"""
        info = getattr(wp, "_bcp_pack", None)
        if info is not None:
            net = info[0]()
            if net is not None:
                net.pack_section_used(info[1], info[2], int(self.b.call("bcp_conv3_last_section")))

    def conv3_fwd_raw(self, x, wp, Cout, KD, nslab):
        """conv (forward or dgrad) whose split-K partial slabs ARE the result: float32 [nslab, N, D, H, W, Cout], no bias; the
        consumer (norm_fwd_slabs / norm_bwd_slabs) sums them on its way in"""
        self._chk(x, wp)
        N, D, H, W, Cin = x.shape
        slabs = torch.empty((nslab, N, D, H, W, Cout), dtype=torch.float32, device=x.device)
        self.b.call("bcp_conv3_fwd_raw", _p(x), _p(wp), _p(slabs), int(nslab), N, D, H, W, Cin, Cout, KD, _p(self._amax_of(x)), self.stream(x))
        self._note_pack_section(wp)
        return slabs

    def norm_fwd_slabs(self, src, nslab, bias, G, gamma, beta, rmean, rvar, act, chan_scale=None, elem_mask=None, elem_scale=1.0,
                       residual=None, momentum=0.1, eps=1e-5, stats_only=False):
        """src: raw conv slabs [nslab,N,D,H,W,C] (+ the conv bias) -> (a, stats, y): the statistics pass sums the slabs (+ bias) on its way
        in and writes y once; finalize and apply as bcp_norm_fwd"""
        elem_mask, em_seed, em_keep = self._mask_split(elem_mask)
        self._chk(src, bias, gamma, beta, rmean, rvar, chan_scale, elem_mask, residual)
        shape = tuple(src.shape[1:]) if src.dim() == 6 else tuple(src.shape)
        N, Cc = shape[0], shape[-1]
        n = 1
        for d in shape:
            n *= d
        rows = n // Cc
        rpg, rps = rows // G, rows // N
        y = torch.empty(shape, dtype=torch.float32, device=src.device)
        stats = torch.empty((5, G, Cc), dtype=torch.float32, device=src.device)
        out = None if stats_only else torch.empty(shape, dtype=torch.float32, device=src.device)
        ws = self.workspace("norm", self._ws_bytes("bcp_norm_workspace_bytes", G, rpg, Cc), src)
        amax = self._amax_slot(out)
        self.b.call("bcp_norm_fwd_slabs", _p(src), int(nslab), n, _p(bias), _p(y), G, rpg, Cc, _p(gamma), _p(beta), _p(rmean), _p(rvar),
                    float(momentum), float(eps), act, _p(chan_scale), rps, _p(elem_mask), float(elem_scale), em_seed, em_keep, _p(residual),
                    _p(stats), _p(ws), _p(out), _p(amax), self.stream(src))
        return out, stats, y

    def norm_bwd_slabs(self, y, da_src, nslab, G, stats, act, dgamma=None, dbeta=None, accumulate=False, chan_scale=None, elem_mask=None,
                       elem_scale=1.0):
        """da_src: the dgrad's raw slabs [nslab,N,D,H,W,C] -> (dy, da): the backward-statistics pass sums the slabs and writes da once"""
        elem_mask, em_seed, em_keep = self._mask_split(elem_mask)
        self._chk(y, da_src, stats, dgamma, dbeta, chan_scale, elem_mask)
        N, Cc = y.shape[0], y.shape[-1]
        rows = y.numel() // Cc
        rpg, rps = rows // G, rows // N
        dy = torch.empty_like(y)
        da = torch.empty_like(y)
        ws = self.workspace("norm", self._ws_bytes("bcp_norm_workspace_bytes", G, rpg, Cc), y)
        self.b.call("bcp_norm_bwd_slabs", _p(y), _p(da_src), int(nslab), y.numel(), _p(da), G, rpg, Cc, _p(stats), act,
                    _p(chan_scale), rps, _p(elem_mask), float(elem_scale), em_seed, em_keep, _p(dgamma), _p(dbeta), int(bool(accumulate)), _p(ws),
                    _p(dy), _p(self._amax_slot(dy, backward=True)), self.stream(y))
        return dy, da

    # ------------------------------------------------------------------ 3x3(x3) conv
    def conv3_packed_floats(self, Cin, Cout, KD):
        """floats per packed weight buffer: the fp32 pack + the three-piece bf16 pack behind it (csrc/conv3b.hip)"""
        return int(self.b.call("bcp_conv3_packed_weight_floats", int(Cin), int(Cout), int(KD)))

    def conv3_pack(self, w, KD):
        """torch weight [Cout,Cin,(3,)3,3] -> (wp_fwd, wp_dgrad) packed for the MFMA kernels"""
        self._chk(w)
        Cout, Cin = w.shape[0], w.shape[1]
        n = self.b.call("bcp_conv3_packed_weight_floats", Cin, Cout, KD)
        wf = torch.empty(int(n), dtype=torch.float32, device=w.device)
        wd = torch.empty(int(n), dtype=torch.float32, device=w.device)
        self.b.call("bcp_conv3_pack_weight", _p(w), _p(wf), _p(wd), Cin, Cout, KD, self.stream(w))
        return wf, wd

    def conv3_pack_many(self, descs, n):
        """descs: uint8 device tensor holding n 40-byte descriptors (networks/_hipnet.py builds it)"""
        self._chk(descs)
        self.b.call("bcp_conv3_pack_many", _p(descs), int(n), self.stream(descs))

    def conv3_fwd(self, x, wp, bias, Cout, KD, out=None, accumulate=False):
        self._chk(x, wp, bias)
        N, D, H, W, Cin = x.shape
        if out is None:
            out = torch.empty((N, D, H, W, Cout), dtype=torch.float32, device=x.device)
        nbytes = self._ws_bytes("bcp_conv3_fwd_workspace_bytes", N, D, H, W, Cin, Cout, KD)
        ws = self.workspace("conv3", nbytes, x) if nbytes else None
        self.b.call("bcp_conv3_fwd", _p(x), _p(wp), _p(bias), _p(out), N, D, H, W, Cin, Cout, KD, int(bool(accumulate)), _p(ws), _p(self._amax_of(x)),
                    self.stream(x))
        self._note_pack_section(wp)
        return self._no_amax(out)

    def conv3_fwd_stats(self, x, wp, bias, Cout, KD, groups):
        """conv + fused norm statistics -> (y, partial, rows); rows == 0: statistics not fused for this shape (partial None)"""
        self._chk(x, wp, bias)
        N, D, H, W, Cin = x.shape
        nbytes = self._ws_bytes("bcp_conv3_fwd_workspace_bytes", N, D, H, W, Cin, Cout, KD)
        rows = self._ws_bytes("bcp_conv3_stat_rows", N, D, H, W, Cin, Cout, KD, groups, 1 if nbytes else 0)
        if rows == 0:
            return self.conv3_fwd(x, wp, bias, Cout, KD), None, 0
        ws = self.workspace("conv3", nbytes, x) if nbytes else None
        out = torch.empty((N, D, H, W, Cout), dtype=torch.float32, device=x.device)
        part = self.workspace(("statpart", rows), groups * rows * Cout * 16, x)
        self.b.call("bcp_conv3_fwd_stats", _p(x), _p(wp), _p(bias), _p(out), N, D, H, W, Cin, Cout, KD, _p(ws), _p(part), groups, _p(self._amax_of(x)),
                    self.stream(x))
        self._note_pack_section(wp)
        return out, part, rows

    def conv3_dgrad_bwdstats(self, dy, wd, Cin, KD, y_prev, stats_prev, act, groups):
        """dgrad whose epilogue also accumulates the backward statistics of the norm layer that consumes it (pre-norm tensor y_prev,
        statistics stats_prev): -> (da, partial, rows) for norm_bwd(partial=, nb=); rows == 0: not fused for this shape (plain dgrad)"""
        self._chk(dy, wd, y_prev, stats_prev)
        N, D, H, W, Cout = dy.shape
        rows = self._ws_bytes("bcp_conv3_bwdstat_rows", N, D, H, W, Cout, Cin, KD, groups)
        if rows == 0:
            return self.conv3_fwd(dy, wd, None, Cin, KD), None, 0
        nbytes = self._ws_bytes("bcp_conv3_fwd_workspace_bytes", N, D, H, W, Cout, Cin, KD)
        ws = self.workspace("conv3", nbytes, dy) if nbytes else None
        da = torch.empty((N, D, H, W, Cin), dtype=torch.float32, device=dy.device)
        part = self.workspace(("bstatpart", rows), groups * rows * Cin * 16, dy)
        self.b.call("bcp_conv3_dgrad_bwdstats", _p(dy), _p(wd), _p(da), N, D, H, W, Cout, Cin, KD, _p(y_prev), _p(stats_prev), act, _p(ws),
                    _p(part), groups, _p(self._amax_of(dy)), self.stream(dy))
        self._note_pack_section(wd)
        return da, part, rows

    def conv3_wgrad(self, x, dy, dw, KD, accumulate=False):
        """dw: torch-layout gradient tensor [Cout,Cin,(3,)3,3], written (or += when accumulate)"""
        self._chk(x, dy, dw)
        N, D, H, W, Cin = x.shape
        Cout = dy.shape[-1]
        nbytes = self._ws_bytes("bcp_conv3_wgrad_workspace_bytes", N, D, H, W, Cin, Cout, KD)
        ws = self.workspace("wgrad", nbytes, x)
        self.b.call("bcp_conv3_wgrad", _p(x), _p(dy), _p(dw), N, D, H, W, Cin, Cout, KD, int(bool(accumulate)), _p(ws), _p(self._amax_of(x)),
                    _p(self._amax_of(dy)), self.stream(x))
        return dw

    def conv3_c1_fwd(self, x, w, bias, KD, out=None):
        self._chk(x, w, bias)
        N, D, H, W, Cin = x.shape
        assert Cin == 1 and w.shape[0] == 16
        if out is None:
            out = torch.empty((N, D, H, W, 16), dtype=torch.float32, device=x.device)
        self.b.call("bcp_conv3_c1_fwd", _p(x), _p(w), _p(bias), _p(out), N, D, H, W, KD, self.stream(x))
        return out

    def conv3_c1_fwd_stats(self, x, w, bias, KD, groups):
        """first layer + fused norm statistics -> (y, partial, rows), as conv3_fwd_stats"""
        self._chk(x, w, bias)
        N, D, H, W, Cin = x.shape
        assert Cin == 1 and w.shape[0] == 16
        rows = self._ws_bytes("bcp_conv3_c1_stat_rows", N, D, H, W, KD, groups)
        if rows == 0:
            return self.conv3_c1_fwd(x, w, bias, KD), None, 0
        out = torch.empty((N, D, H, W, 16), dtype=torch.float32, device=x.device)
        part = self.workspace(("statpart", rows), groups * rows * 16 * 16, x)
        self.b.call("bcp_conv3_c1_fwd_stats", _p(x), _p(w), _p(bias), _p(out), N, D, H, W, KD, _p(part), groups, self.stream(x))
        return out, part, rows

    def conv3_c1_norm_ok(self, xshape, KD, groups):
        N, D, H, W, _ = xshape
        return self._ws_bytes("bcp_conv3_c1_norm_workspace_bytes", int(N), int(D), int(H), int(W), int(KD), int(groups)) > 0

    def conv3_c1_norm_fwd(self, x, w, bias, KD, G, gamma, beta, rmean, rvar, act, elem_mask=None, elem_scale=1.0, momentum=0.1, eps=1e-5):
        """first layer + norm + activation with recompute (bcp_conv3_c1_norm_fwd): -> (a, stats); y is never materialised"""
        elem_mask, em_seed, em_keep = self._mask_split(elem_mask)
        self._chk(x, w, bias, gamma, beta, rmean, rvar, elem_mask)
        N, D, H, W, Cin = x.shape
        assert Cin == 1 and w.shape[0] == 16
        nbytes = self._ws_bytes("bcp_conv3_c1_norm_workspace_bytes", N, D, H, W, KD, G)
        ws = self.workspace("c1norm", nbytes, x)
        stats = torch.empty((5, G, 16), dtype=torch.float32, device=x.device)
        out = torch.empty((N, D, H, W, 16), dtype=torch.float32, device=x.device)
        self.b.call("bcp_conv3_c1_norm_fwd", _p(x), _p(w), _p(bias), N, D, H, W, KD, G, _p(gamma), _p(beta), _p(rmean), _p(rvar), float(momentum),
                    float(eps), act, _p(elem_mask), float(elem_scale), em_seed, em_keep, _p(stats), _p(ws), _p(out), _p(self._amax_slot(out)),
                    self.stream(x))
        return out, stats

    def conv3_c1_norm_bwd(self, x, w, bias, KD, G, stats, da, act, dgamma=None, dbeta=None, accumulate=False, elem_mask=None, elem_scale=1.0):
        """backward of conv3_c1_norm_fwd's norm: y recomputed from x, -> dy (the input of conv3_c1_wgrad)"""
        elem_mask, em_seed, em_keep = self._mask_split(elem_mask)
        self._chk(x, w, bias, stats, da, dgamma, dbeta, elem_mask)
        N, D, H, W, _ = x.shape
        nbytes = self._ws_bytes("bcp_conv3_c1_norm_workspace_bytes", N, D, H, W, KD, G)
        ws = self.workspace("c1norm", nbytes, x)
        dy = torch.empty((N, D, H, W, 16), dtype=torch.float32, device=x.device)
        self.b.call("bcp_conv3_c1_norm_bwd", _p(x), _p(w), _p(bias), _p(da), N, D, H, W, KD, G, _p(stats), act, _p(elem_mask), float(elem_scale),
                    em_seed, em_keep, _p(dgamma), _p(dbeta), int(bool(accumulate)), _p(ws), _p(dy), self.stream(x))
        return dy

    # networks: the first layer's backward leaves its weight gradient itself (bcp_conv3_c1_norm_bwd_wgrad: dy is never written); False
    # (BCP_C1_BWD_FUSED=0, a measurement switch): conv3_c1_norm_bwd + conv3_c1_wgrad on the weight-gradient stream, the round-3 pair
    C1_BWD_FUSED = os.environ.get("BCP_C1_BWD_FUSED", "1") != "0"

    def conv3_c1_norm_bwd_wgrad(self, x, w, bias, KD, G, stats, da, act, dw, dgamma=None, dbeta=None, accumulate=False, dw_accumulate=False,
                                elem_mask=None, elem_scale=1.0):
        """conv3_c1_norm_bwd + conv3_c1_wgrad in one call: dgamma / dbeta of the norm and dw of the conv; the gradient w.r.t. the conv
        output in between exists tile by tile in LDS only"""
        elem_mask, em_seed, em_keep = self._mask_split(elem_mask)
        self._chk(x, w, bias, stats, da, dgamma, dbeta, elem_mask, dw)
        N, D, H, W, _ = x.shape
        nbytes = self._ws_bytes("bcp_conv3_c1_norm_bwd_wgrad_workspace_bytes", N, D, H, W, KD, G)
        ws = self.workspace("c1normw", nbytes, x)
        self.b.call("bcp_conv3_c1_norm_bwd_wgrad", _p(x), _p(w), _p(bias), _p(da), N, D, H, W, KD, G, _p(stats), act, _p(elem_mask), float(elem_scale),
                    em_seed, em_keep, _p(dgamma), _p(dbeta), int(bool(accumulate)), _p(ws), _p(dw), int(bool(dw_accumulate)), self.stream(x))
        return dw

    def conv3_c1_wgrad(self, x, dy, dw, KD, accumulate=False):
        self._chk(x, dy, dw)
        N, D, H, W, _ = x.shape
        nbytes = self._ws_bytes("bcp_conv3_wgrad_workspace_bytes", N, D, H, W, 1, 16, KD)
        ws = self.workspace("wgrad", nbytes, x)
        self.b.call("bcp_conv3_c1_wgrad", _p(x), _p(dy), _p(dw), N, D, H, W, KD, int(bool(accumulate)), _p(ws), self.stream(x))
        return dw

    # ------------------------------------------------------------------ k2s2 / 1x1 GEMM convs
    def k2_pack(self, w, Cin, Cout, kind):
        self._chk(w)
        bp = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
        self.b.call("bcp_k2_pack_weight", _p(w), _p(bp), Cin, Cout, kind, self.stream(w))
        return bp

    def k2_pack_desc(self, w, bp, Cin, Cout, kind):
        """64-byte descriptor (bytes) of one (weight, packed matrix, kind) for k2_pack_many"""
        import ctypes
        buf = ctypes.create_string_buffer(64)
        self.b.call("bcp_k2_pack_desc", _p(w), _p(bp), Cin, Cout, kind, ctypes.addressof(buf))
        return buf.raw

    def k2_pack_many(self, descs, n):
        self._chk(descs)
        self.b.call("bcp_k2_pack_many", _p(descs), int(n), self.stream(descs))

    def down_fwd(self, x, bp, bias, Cout, out=None):
        self._chk(x, bp, bias)
        N, D, H, W, Cin = x.shape
        if out is None:
            out = torch.empty((N, D // 2, H // 2, W // 2, Cout), dtype=torch.float32, device=x.device)
        self.b.call("bcp_down_fwd", _p(x), _p(bp), _p(bias), _p(out), N, D, H, W, Cin, Cout, self.stream(x))
        return self._no_amax(out)

    def k2_stat_rows(self, kind, xshape, Cout, groups):
        """partial rows per normalisation group k2_fwd_stats leaves for this shape (kind 0: down conv, 1: transposed conv); 0: not fused"""
        N, D, H, W, Cin = xshape
        fine = (D, H, W) if kind == 0 else (2 * D, 2 * H, 2 * W)
        return self._ws_bytes("bcp_k2_stat_rows", int(kind), N, fine[0], fine[1], fine[2], Cin, Cout, int(groups))

    def k2_fwd_stats(self, kind, x, bp, bias, Cout, groups):
        """down (kind 0) / transposed (kind 1) conv forward + the norm statistics of its output in the GEMM's epilogue (round 6) ->
        (y, partial, rows) for norm_fwd(partial=, nb=); only where k2_stat_rows(...) > 0"""
        self._chk(x, bp, bias)
        N, D, H, W, Cin = x.shape
        fine = (D, H, W) if kind == 0 else (2 * D, 2 * H, 2 * W)
        rows = self.k2_stat_rows(kind, x.shape, Cout, groups)
        assert rows > 0, "k2_fwd_stats: statistics not fused for this shape (check k2_stat_rows)"
        osp = (D // 2, H // 2, W // 2) if kind == 0 else fine
        out = torch.empty((N,) + osp + (Cout,), dtype=torch.float32, device=x.device)
        part = self.workspace(("statpart", rows), groups * rows * Cout * 16, x)
        self.b.call("bcp_down_fwd_stats" if kind == 0 else "bcp_up_fwd_stats", _p(x), _p(bp), _p(bias), _p(out), N, fine[0], fine[1], fine[2], Cin, Cout,
                    _p(part), int(groups), self.stream(x))
        return self._no_amax(out), part, rows

    def k2_bwdstat_rows(self, kind, dyshape, Cin, groups):
        """partial rows per group k2_dgrad_bwdstats leaves (kind 0: dgrad of the down conv, dy coarse; 1: of the transposed conv, dy fine); 0: not fused"""
        N, D, H, W, Cout = dyshape
        fine = (2 * D, 2 * H, 2 * W) if kind == 0 else (D, H, W)
        return self._ws_bytes("bcp_k2_bwdstat_rows", int(kind), N, fine[0], fine[1], fine[2], int(Cin), Cout, int(groups))

    def k2_dgrad_bwdstats(self, kind, dy, bp, Cin, y_prev, stats_prev, act, groups, out=None, accumulate=False):
        """dgrad of the down (kind 0) / transposed (kind 1) conv whose epilogue also accumulates the backward statistics of the norm layer in
        front of it (pre-norm tensor y_prev -- laid out like the result -- and statistics stats_prev) -> (dx, partial, rows) for
        norm_bwd(partial=, nb=); only where k2_bwdstat_rows(...) > 0.  out + accumulate: dx = out + dgrad (the skip gradient joined in place)"""
        self._chk(dy, bp, out, y_prev, stats_prev)
        N, D, H, W, Cout = dy.shape
        fine = (2 * D, 2 * H, 2 * W) if kind == 0 else (D, H, W)
        rows = self.k2_bwdstat_rows(kind, dy.shape, Cin, groups)
        assert rows > 0, "k2_dgrad_bwdstats: statistics not fused for this shape (check k2_bwdstat_rows)"
        osp = fine if kind == 0 else (D // 2, H // 2, W // 2)
        if out is None:
            assert not accumulate
            out = torch.empty((N,) + osp + (Cin,), dtype=torch.float32, device=dy.device)
        assert tuple(y_prev.shape) == tuple(out.shape), "k2_dgrad_bwdstats: y_prev must be laid out like the result"
        part = self.workspace(("bstatpart", rows), groups * rows * Cin * 16, dy)
        self.b.call("bcp_down_dgrad_bwdstats" if kind == 0 else "bcp_up_dgrad_bwdstats", _p(dy), _p(bp), _p(out), N, fine[0], fine[1], fine[2], int(Cin), Cout,
                    int(bool(accumulate)), _p(y_prev), _p(stats_prev), int(act), _p(part), int(groups), self.stream(dy))
        return self._no_amax(out), part, rows

    # ---- round 6: transposed conv + norm + activation (+ skip add) with the conv output recomputed instead of stored
    def up_norm_rows(self, xshape, Cout, groups):
        """> 0: bcp_up_fwd_norm / bcp_up_norm_bwd serve a transposed conv of this (coarse) input shape (option up_recompute)"""
        N, D, H, W, Cin = xshape
        return self._ws_bytes("bcp_up_norm_rows", N, 2 * D, 2 * H, 2 * W, Cin, int(Cout), int(groups))

    def up_fwd_norm(self, x, bp, bias, Cout, G, gamma, beta, rmean, rvar, act, residual=None, momentum=0.1, eps=1e-5):
        """-> (a, stats): a = act(norm(up(x))) + residual; y = up(x) is never materialised (both passes of the norm recompute it)"""
        self._chk(x, bp, bias, gamma, beta, rmean, rvar, residual)
        N, D, H, W, Cin = x.shape
        fine = (2 * D, 2 * H, 2 * W)
        ws = self.workspace("upnorm", self._ws_bytes("bcp_up_norm_workspace_bytes", N, fine[0], fine[1], fine[2], Cin, int(Cout), int(G)), x)
        stats = torch.empty((5, G, Cout), dtype=torch.float32, device=x.device)
        out = torch.empty((N,) + fine + (Cout,), dtype=torch.float32, device=x.device)
        self.b.call("bcp_up_fwd_norm", _p(x), _p(bp), _p(bias), N, fine[0], fine[1], fine[2], Cin, int(Cout), int(G), _p(gamma), _p(beta), _p(rmean),
                    _p(rvar), float(momentum), float(eps), int(act), _p(residual), _p(stats), _p(ws), _p(out), _p(self._amax_slot(out)), self.stream(x))
        return out, stats

    def up_norm_bwd(self, x, bp, bias, Cout, G, stats, da, act, dgamma=None, dbeta=None, accumulate=False):
        """backward of up_fwd_norm's norm: y recomputed from x -> dy (the input of the transposed conv's dgrad and weight gradient)"""
        self._chk(x, bp, bias, stats, da, dgamma, dbeta)
        N, D, H, W, Cin = x.shape
        fine = (2 * D, 2 * H, 2 * W)
        ws = self.workspace("upnorm", self._ws_bytes("bcp_up_norm_workspace_bytes", N, fine[0], fine[1], fine[2], Cin, int(Cout), int(G)), x)
        dy = torch.empty((N,) + fine + (Cout,), dtype=torch.float32, device=x.device)
        self.b.call("bcp_up_norm_bwd", _p(x), _p(bp), _p(bias), _p(da), N, fine[0], fine[1], fine[2], Cin, int(Cout), int(G), _p(stats), int(act),
                    _p(dgamma), _p(dbeta), int(bool(accumulate)), _p(ws), _p(dy), self.stream(x))
        return self._no_amax(dy)

    def down_dgrad(self, dy, bp, Cin, out=None, accumulate=False):
        self._chk(dy, bp, out)
        N, Dc, Hc, Wc, Cout = dy.shape
        if out is None:
            assert not accumulate
            out = torch.empty((N, 2 * Dc, 2 * Hc, 2 * Wc, Cin), dtype=torch.float32, device=dy.device)
        self.b.call("bcp_down_dgrad", _p(dy), _p(bp), _p(out), N, 2 * Dc, 2 * Hc, 2 * Wc, Cin, Cout, int(bool(accumulate)), self.stream(dy))
        return self._no_amax(out)

    def up_fwd(self, x, bp, bias, Cout, out=None):
        self._chk(x, bp, bias)
        N, Dc, Hc, Wc, Cin = x.shape
        if out is None:
            out = torch.empty((N, 2 * Dc, 2 * Hc, 2 * Wc, Cout), dtype=torch.float32, device=x.device)
        self.b.call("bcp_up_fwd", _p(x), _p(bp), _p(bias), _p(out), N, 2 * Dc, 2 * Hc, 2 * Wc, Cin, Cout, self.stream(x))
        return self._no_amax(out)

    def up_dgrad(self, dy, bp, Cin, out=None, accumulate=False):
        self._chk(dy, bp, out)
        N, D, H, W, Cout = dy.shape
        if out is None:
            out = torch.empty((N, D // 2, H // 2, W // 2, Cin), dtype=torch.float32, device=dy.device)
        self.b.call("bcp_up_dgrad", _p(dy), _p(bp), _p(out), N, D, H, W, Cin, Cout, int(bool(accumulate)), self.stream(dy))
        return self._no_amax(out)

    def pw_fwd(self, x, bp, bias, Cout, out=None):
        self._chk(x, bp, bias)
        Cin = x.shape[-1]
        rows = x.numel() // Cin
        if out is None:
            out = torch.empty(tuple(x.shape[:-1]) + (Cout,), dtype=torch.float32, device=x.device)
        self.b.call("bcp_pw_fwd", _p(x), _p(bp), _p(bias), _p(out), rows, Cin, Cout, self.stream(x))
        return self._no_amax(out)

    def k2_wgrad(self, x, dy, dw, kind, accumulate=False):
        """kind WG_DOWN: x fine, dy coarse; WG_UP: x coarse, dy fine; WG_PW: same grid."""
        self._chk(x, dy, dw)
        Cin, Cout = x.shape[-1], dy.shape[-1]
        fine = x if kind != WG_UP else dy
        N, D, H, W = fine.shape[:4]
        if kind == WG_DOWN:
            M, K, Nn = N * (D // 2) * (H // 2) * (W // 2), 8 * Cin, Cout
        elif kind == WG_UP:
            M, K, Nn = N * (D // 2) * (H // 2) * (W // 2), Cin, 8 * Cout
        else:
            M, K, Nn = N * D * H * W, Cin, Cout
        nbytes = self._ws_bytes("bcp_tn_workspace_bytes", M, K, Nn)
        ws = self.workspace("wgrad", nbytes, x)
        self.b.call("bcp_k2_wgrad", _p(x), _p(dy), _p(dw), N, D, H, W, Cin, Cout, kind, int(bool(accumulate)), _p(ws), self.stream(x))
        return dw

    def pw16_fwd(self, x, w, bias, Cout, out=None):
        self._chk(x, w, bias)
        assert x.shape[-1] == 16
        nvox = x.numel() // 16
        if out is None:
            out = torch.empty(tuple(x.shape[:-1]) + (Cout,), dtype=torch.float32, device=x.device)
        self.b.call("bcp_pw16_fwd", _p(x), _p(w), _p(bias), _p(out), nvox, Cout, self.stream(x))
        return out

    def pw16_bwd(self, x, dy, w, dw, db, accumulate=False, dx=None):
        self._chk(x, dy, w, dw, db)
        Cout = dy.shape[-1]
        nvox = x.numel() // 16
        if dx is None:
            dx = torch.empty_like(x)
        ws = self.workspace("pw16", Cout * 17 * 8, x)
        self.b.call("bcp_pw16_bwd", _p(x), _p(dy), _p(w), _p(dx), _p(dw), _p(db), nvox, Cout, int(bool(accumulate)), _p(ws), self.stream(x))
        return dx

    def pw16_fwd_norm(self, y_raw, stats, chan_scale, G, act, w, bias, Cout, out=None):
        """the 1x1x1 head on act(norm(y_raw)) * chan_scale without materialising that activation (bcp_pw16_fwd_norm)"""
        self._chk(y_raw, stats, chan_scale, w, bias)
        assert y_raw.shape[-1] == 16
        N = y_raw.shape[0]
        nvox = y_raw.numel() // 16
        if out is None:
            out = torch.empty(tuple(y_raw.shape[:-1]) + (Cout,), dtype=torch.float32, device=y_raw.device)
        self.b.call("bcp_pw16_fwd_norm", _p(y_raw), _p(stats), _p(chan_scale), N, G, act, _p(w), _p(bias), _p(out), nvox, Cout,
                    self.stream(y_raw))
        return out

    def pw16_bwd_norm(self, y_raw, stats, chan_scale, G, act, dy, w, dw, db, accumulate=False, dx=None):
        """backward of pw16_fwd_norm: dw / db from the recomputed activation; returns its gradient (input of norm_bwd)"""
        self._chk(y_raw, stats, chan_scale, dy, w, dw, db)
        Cout = dy.shape[-1]
        N = y_raw.shape[0]
        nvox = y_raw.numel() // 16
        if dx is None:
            dx = torch.empty_like(y_raw)
        ws = self.workspace("pw16", Cout * 17 * 8, y_raw)
        self.b.call("bcp_pw16_bwd_norm", _p(y_raw), _p(stats), _p(chan_scale), N, G, act, _p(dy), _p(w), _p(dx), _p(dw), _p(db), nvox, Cout,
                    int(bool(accumulate)), _p(ws), self.stream(y_raw))
        return dx

    # networks/VNet.py: the fused head's backward runs THROUGH the last conv's norm (bcp_pw16_bwd_norm_bwd); False (BCP_HEAD_BWD_FUSED=0, a
    # measurement switch): pw16_bwd_norm + norm_bwd, the round-4 chain
    HEAD_BWD_FUSED = os.environ.get("BCP_HEAD_BWD_FUSED", "1") != "0"

    def pw16_bwd_norm_bwd(self, y_raw, stats, chan_scale, G, act, dy, w, dw, db, dgamma=None, dbeta=None, norm_accumulate=False, accumulate=False):
        """pw16_bwd_norm followed by norm_bwd of the same layer in one call: dw / db of the head (+= when accumulate), dgamma / dbeta of the
        norm (+= when norm_accumulate), returns the gradient w.r.t. y_raw; the activation gradient in between is never written"""
        self._chk(y_raw, stats, chan_scale, dy, w, dw, db, dgamma, dbeta)
        Cout = dy.shape[-1]
        N = y_raw.shape[0]
        nvox = y_raw.numel() // 16
        out = torch.empty_like(y_raw)
        ws = self.workspace("pw16nb", self._ws_bytes("bcp_pw16_bwd_norm_bwd_workspace_bytes", N, int(G), nvox), y_raw)
        self.b.call("bcp_pw16_bwd_norm_bwd", _p(y_raw), _p(stats), _p(chan_scale), N, int(G), act, _p(dy), _p(w), _p(out), _p(dw), _p(db),
                    _p(dgamma), _p(dbeta), int(bool(norm_accumulate)), nvox, Cout, int(bool(accumulate)), _p(ws),
                    _p(self._amax_slot(out, backward=True)), self.stream(y_raw))
        return out

    def colsum(self, x, out, accumulate=False):
        self._chk(x, out)
        Cc = x.shape[-1]
        ws = self.workspace("colsum", Cc * 8, x)
        self.b.call("bcp_colsum", _p(x), x.numel() // Cc, Cc, _p(out), int(bool(accumulate)), _p(ws), self.stream(x))
        return out

    # ------------------------------------------------------------------ 2-D U-Net plumbing
    def maxpool2d_fwd(self, x, concat=None):
        """x: contiguous or a channel slab (channel_slab).  The pooled tensor carries x's |max| slots (max |pool(x)| <= max |x|: an upper
        bound is all the fp16 pre-scale of the next conv needs).  concat: the concat buffer whose leading channels x is -- it gets |max|
        slots of its own, started as a copy of x's by this launch (bilinear2x_fwd max-reduces the upsampled half into them)"""
        ldx = self._chk_rows(x)
        N, D, H, W, Cc = x.shape
        assert D == 1
        y = torch.empty((N, 1, H // 2, W // 2, Cc), dtype=torch.float32, device=x.device)
        a_src = self._amax_of(x)
        a_dst = self._amax_slot(concat) if (concat is not None and a_src is not None) else None
        self.b.call("bcp_maxpool2d_fwd", _p(x), ldx, _p(y), N, H, W, Cc, _p(a_src if a_dst is not None else None), _p(a_dst), self.stream(x))
        if a_src is not None:
            y._bcp_amax = a_src
        return y

    def maxpool3d_k3s2_fwd(self, x):
        """nn.MaxPool3d(3, stride=2) of a channels-last volume (the V-Net's pooled x5 features; forward only)"""
        self._chk(x)
        N, D, H, W, Cc = x.shape
        y = torch.empty((N, (D - 3) // 2 + 1, (H - 3) // 2 + 1, (W - 3) // 2 + 1, Cc), dtype=torch.float32, device=x.device)
        self.b.call("bcp_maxpool3d_k3s2_fwd", _p(x), _p(y), N, D, H, W, Cc, self.stream(x))
        return y

    def maxpool2d_bwd(self, x, dy, dx, accumulate=False, add=None):
        """add: a second gradient of x joined on the way out (contiguous or a channel slab, e.g. the skip half of the concat buffer's
        gradient): dx = scatter(dy) + add"""
        self._chk(dy, dx)
        ldx = self._chk_rows(x)
        ld_add = self._chk_rows(add) if add is not None else 0
        N, D, H, W, Cc = x.shape
        self.b.call("bcp_maxpool2d_bwd", _p(x), ldx, _p(dy), _p(dx), N, H, W, Cc, int(bool(accumulate)), _p(add), ld_add, self.stream(x))
        return dx

    def bilinear2x_fwd(self, x, y, y_off):
        """writes the 2x upsample of x [N,1,H,W,C] into channels [y_off, y_off+C) of y [N,1,2H,2W,ld]"""
        self._chk(x, y)
        N, D, H, W, Cc = x.shape
        self.b.call("bcp_bilinear2x_fwd", _p(x), _p(y), N, H, W, Cc, y.shape[-1], y_off, _p(getattr(y, "_bcp_amax", None)), self.stream(x))
        return y

    def bilinear2x_bwd(self, dy, dy_off, Cc):
        self._chk(dy)
        N, D, Ho, Wo, ld = dy.shape
        dx = torch.empty((N, 1, Ho // 2, Wo // 2, Cc), dtype=torch.float32, device=dy.device)
        self.b.call("bcp_bilinear2x_bwd", _p(dy), _p(dx), N, Ho // 2, Wo // 2, Cc, ld, dy_off, self.stream(dy))
        return dx

    def copy_channels(self, src, dst, Cc, src_off=0, dst_off=0, accumulate=False, carry_amax=False):
        """carry_amax: dst gets an |max| slot initialised with src's (the skip half of a concat buffer; bilinear2x_fwd max-reduces the
        other half into it) -- only when src carries one"""
        self._chk(src, dst)
        rows = src.numel() // src.shape[-1]
        a_src = self._amax_of(src) if carry_amax else None
        a_dst = self._amax_slot(dst) if a_src is not None else None
        self.b.call("bcp_copy_channels", _p(src), _p(dst), rows, Cc, src.shape[-1], src_off, dst.shape[-1], dst_off,
                    int(bool(accumulate)), _p(a_src), _p(a_dst), self.stream(src))
        return dst

    # ------------------------------------------------------------------ optimiser / EMA / misc
    def ema(self, dst, src, alpha):
        self._chk(dst, src)
        self.b.call("bcp_ema", _p(dst), _p(src), dst.numel(), float(alpha), self.stream(dst))

    def sgd(self, p, g, buf, lr, momentum, wd, first_step, grad_scale=1.0, ema=None, ema_alpha=0.99):
        self._chk(p, g, buf, ema)
        self.b.call("bcp_sgd", _p(p), _p(g), _p(buf), _p(ema), p.numel(), float(lr), float(momentum), float(wd), float(grad_scale),
                    int(bool(first_step)), float(ema_alpha), self.stream(p))

    def adam(self, p, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0):
        self._chk(p, g, m, v)
        self.b.call("bcp_adam", _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(b1), float(b2), float(eps), int(step),
                    float(grad_scale), self.stream(p))

    def cast(self, x, kind):
        self._chk(x)
        dt = {CAST_I64_U8: torch.uint8, CAST_F32_U8: torch.uint8, CAST_U8_F32: torch.float32, CAST_U8_I64: torch.int64}[kind]
        out = torch.empty(x.shape, dtype=dt, device=x.device)
        self.b.call("bcp_cast", _p(x), _p(out), x.numel(), kind, self.stream(x))
        return out

    def to_u8(self, x):
        """labels of any reference dtype (int64 / float32 / uint8) -> uint8 on device"""
        if x.dtype == torch.uint8:
            return x if x.is_contiguous() else x.contiguous()
        x = x if x.is_contiguous() else x.contiguous()
        if x.dtype == torch.int64:
            return self.cast(x, CAST_I64_U8)
        if x.dtype == torch.float32:
            return self.cast(x, CAST_F32_U8)
        raise _lib.BcpError(f"unsupported label dtype {x.dtype}")

    def axpy(self, y, x, a=1.0):
        self._chk(y, x)
        self.b.call("bcp_axpy", _p(y), _p(x), y.numel(), float(a), self.stream(y))
        return y

    def store_u64(self, dst, values, like):
        """dst[i] = values[i] (<= 16 host integers as kernel arguments): the dropout seed table of a launch plan"""
        n = len(values)
        arr = (C.c_ulonglong * n)(*[int(v) & 0xFFFFFFFFFFFFFFFF for v in values])
        fn, _ = self.b._fns["bcp_store_u64"]            # never recorded: it runs ahead of every replay
        rc = fn(dst.data_ptr(), n, arr, self.stream(like))
        if rc:
            raise _lib.BcpError(f"bcp_store_u64 failed ({rc}): {self.b.last_error()}")

    def bernoulli(self, out, p_keep, keep_value, seed):
        self._chk(out)
        pl = self._rec_plan
        if pl is not None:          # inside a recorded pass the seed is read from the plan's device table (bcp_amd/plan.py)
            slot = pl.seed_slot(out.device)
            n = pl.n_seeds - 1
            fn, _ = self.b._fns["bcp_store_u64"]
            arr = (C.c_ulonglong * 1)(int(seed) & 0xFFFFFFFFFFFFFFFF)
            rc = fn(slot, 1, arr, self.stream(out))
            if rc:
                raise _lib.BcpError(f"bcp_store_u64 failed ({rc}): {self.b.last_error()}")
            self.b.call("bcp_bernoulli_dev", _p(out), out.numel(), float(p_keep), float(keep_value), int(out.dtype == torch.uint8), slot,
                        self.stream(out))
            return out
        self.b.call("bcp_bernoulli", _p(out), out.numel(), float(p_keep), float(keep_value), int(out.dtype == torch.uint8), int(seed),
                    self.stream(out))
        return out

    # ------------------------------------------------------------------ timing (bench)
    def event(self):
        e = C.c_void_p()
        self.b.call("bcp_event_create", C.byref(e))
        return e

    def event_record(self, e, like):
        self.b.call("bcp_event_record", e, self.stream(like))

    def event_elapsed_ms(self, e0, e1):
        ms = C.c_float()
        self.b.call("bcp_event_elapsed_ms", e0, e1, C.byref(ms))
        return ms.value


# ---------------------------------------------------------------------------------------------- measurement hooks
# bench.py's per-op table: HIP events on the launch stream around every call of the ops below while a profile is open
# (Ops.profile_begin / profile_end).  Closed (the default) the wrappers cost one attribute test.
_PROFILED = ("mix_box", "plabel_bin", "plabel_argmax4", "cc_largest", "plabel_cc_largest", "mixloss_fwd", "mixloss_bwd", "mixloss_pair_fwd", "mixloss_pair_bwd", "norm_fwd", "norm_bwd", "norm_fwd_slabs", "norm_bwd_slabs", "conv3_fwd_raw", "conv3_dgrad_bwdstats", "pw16_bwd_norm_bwd", "conv3_c1_norm_bwd_wgrad", "conv3_pack_many",
             "conv3_fwd", "conv3_fwd_stats", "conv3_wgrad", "conv3_c1_fwd", "conv3_c1_fwd_stats", "conv3_c1_norm_fwd", "conv3_c1_norm_bwd", "conv3_c1_wgrad", "k2_pack_many", "down_fwd", "down_dgrad", "up_fwd", "k2_fwd_stats", "k2_dgrad_bwdstats", "up_fwd_norm", "up_norm_bwd",
             "up_dgrad", "pw_fwd", "k2_wgrad", "pw16_fwd", "pw16_bwd", "pw16_fwd_norm", "pw16_bwd_norm", "maxpool2d_fwd", "maxpool2d_bwd", "bilinear2x_fwd", "bilinear2x_bwd",
             "copy_channels", "ema", "sgd", "adam")


def _profiled(name, fn):
    def wrapper(self, *a, **k):
        prof = self._prof
        rec = self.b._rec
        if rec is not None and not k.get("stats_only"):
            # a pass is being recorded (bcp_amd/plan.py): note which of its launches belong to this op, so that a profiled REPLAY can put
            # its events around the same launches (LaunchPlan.run_entries_timed) -- the op's shape key is taken here, once
            ts = [t for t in a if isinstance(t, torch.Tensor)]
            n0 = len(rec.entries)
            r = fn(self, *a, **k)
            if ts:
                rec.spans.append((name, tuple(tuple(t.shape) for t in ts[:3]), tuple(x for x in a if isinstance(x, int))[:3],
                                  sum(1 for t in ts[:2] if getattr(t, "_bcp_amax", None) is not None), n0, len(rec.entries)))
            return r
        if prof is None or k.get("stats_only"):      # (statistics-only norm_fwd behind a fused conv epilogue = one finalize launch: not an op row)
            return fn(self, *a, **k)
        ts = [t for t in a if isinstance(t, torch.Tensor)]
        like = ts[0]
        from . import plan as _plan
        if _plan.PROFILE_ONLY is not None and (name, tuple(like.shape)) not in _plan.PROFILE_ONLY:
            return fn(self, *a, **k)
        e0, e1 = self._prof_event(), self._prof_event()
        self.event_record(e0, like)
        r = fn(self, *a, **k)
        self.event_record(e1, like)
        extra = tuple(x for x in a if isinstance(x, int))[:3]
        namax = sum(1 for t in ts[:2] if getattr(t, "_bcp_amax", None) is not None)      # operands that carried their |max| (fp16 instances)
        prof.append((name, tuple(tuple(t.shape) for t in ts[:3]), extra, e0, e1, namax))
        return r
    wrapper.__name__ = name
    wrapper.__doc__ = fn.__doc__
    return wrapper


def _install_profile_hooks():
    Ops._prof = None
    Ops._prof_pool = None

    def _prof_event(self):
        if self._prof_pool:
            return self._prof_pool.pop()
        return self.event()

    def profile_begin(self):
        """start recording (op, shapes, HIP events) of every profiled op"""
        self._prof = []
        if self._prof_pool is None:
            self._prof_pool = []

    def profile_end(self):
        """-> [(op, shapes, ints, milliseconds)] in call order; the caller must have synchronised the device"""
        rec, self._prof = self._prof, None
        out = []
        for name, shapes, extra, e0, e1, namax in rec:
            out.append((name, shapes, extra, self.event_elapsed_ms(e0, e1), namax))
            self._prof_pool.append(e0); self._prof_pool.append(e1)
        return out

    Ops._prof_event, Ops.profile_begin, Ops.profile_end = _prof_event, profile_begin, profile_end
    for n in _PROFILED:
        setattr(Ops, n, _profiled(n, getattr(Ops, n)))


_install_profile_hooks()
