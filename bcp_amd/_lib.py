"""ctypes binding of libbcp_hip.so (include/bcp_hip.h).

The product path loads ONLY bcp_amd/csrc/libbcp_hip.so (built for gfx950 by
__graft_entry__.build()).  If it is missing this module raises: there is no CPU fallback and
the oracle is never imported from here.  `Binding` is parametrised by a CDLL handle only so that
tests can drive the very same wrappers against the host kernel-logic simulator
(tests/_emu/libbcp_emu.so, see tools/emu) with CPU tensors.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbcp_hip.so")

ABI_VERSION = 511      # include/bcp_hip.h BCP_ABI_VERSION: the revision these signatures were written against

P = C.c_void_p
I = C.c_int
L = C.c_longlong
F = C.c_float
U64 = C.c_ulonglong
SZ = C.c_size_t

# name -> (restype, argtypes)
_SIGS = {
    "bcp_version": (I, []),
    "bcp_set_option": (I, [C.c_char_p, C.c_char_p]),
    "bcp_comm_available": (I, []),
    "bcp_comm_unique_id": (I, [P]),
    "bcp_comm_init_rank": (I, [C.POINTER(P), I, I, P]),
    "bcp_comm_count": (I, [P, C.POINTER(I)]),
    "bcp_allreduce_f32": (I, [P, P, C.c_longlong, P]),
    "bcp_comm_destroy": (I, [P]),
    "bcp_last_error": (C.c_char_p, []),
    "bcp_device_arch": (I, [C.c_char_p, I]),
    "bcp_event_create": (I, [C.POINTER(P)]),
    "bcp_event_record": (I, [P, P]),
    "bcp_event_elapsed_ms": (I, [P, P, C.POINTER(F)]),
    "bcp_event_destroy": (I, [P]),
    "bcp_mix_box": (I, [P, P, P, I, I, I, I, I, P, P]),
    "bcp_plabel_bin": (I, [P, P, L, F, P]),
    "bcp_plabel_argmax4": (I, [P, P, L, P]),
    "bcp_cc_workspace_bytes": (SZ, [I, I, I, I, I]),
    "bcp_cc_largest": (I, [P, P, P, I, I, I, I, I, I, P, P]),
    "bcp_plabel_cc_largest": (I, [P, I, F, P, P, P, I, I, I, I, I, I, P, P]),
    "bcp_mixloss_workspace_bytes": (SZ, [I, I]),
    "bcp_mixloss_fwd": (I, [P, P, P, P, P, I, I, I, I, I, I, F, F, P, P, P, P, P]),
    "bcp_mixloss_bwd": (I, [P, P, P, P, P, I, I, I, I, I, I, P, F, F, P, I, P, P]),
    "bcp_mixloss_pair_workspace_bytes": (SZ, [I, I]),
    "bcp_mixloss_pair_fwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, F, F, F, F, P, P, P, P]),
    "bcp_mixloss_pair_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, P, F, F, P, I, P, P]),
    "bcp_dice_prob_workspace_bytes": (SZ, [I]),
    "bcp_dice_prob_fwd": (I, [P, L, L, L, P, P, I, P, I, I, I, I, I, P, P, P, P]),
    "bcp_dice_prob_bwd": (I, [P, L, L, L, P, P, I, P, I, I, I, I, I, P, P, F, P, P]),
    "bcp_norm_workspace_bytes": (SZ, [I, L, I]),
    "bcp_norm_fwd": (I, [P, I, L, I, P, P, P, P, F, F, I, P, L, P, F, P, F, P, P, P, P, I, P, L, P, P]),
    "bcp_norm_bwd": (I, [P, P, I, L, I, P, I, P, L, P, F, P, F, P, P, I, P, P, I, P, P, P]),
    "bcp_norm_slabs_ok": (I, [I, L, I]),
    "bcp_norm_fwd_slabs": (I, [P, I, L, P, P, I, L, I, P, P, P, P, F, F, I, P, L, P, F, P, F, P, P, P, P, P, P]),
    "bcp_norm_bwd_slabs": (I, [P, P, I, L, P, I, L, I, P, I, P, L, P, F, P, F, P, P, I, P, P, P, P]),
    "bcp_conv3_fwd_nslabs": (I, [I, I, I, I, I, I, I]),
    "bcp_conv3_bwdstat_rows": (I, [I, I, I, I, I, I, I, I]),
    "bcp_conv3_dgrad_bwdstats": (I, [P, P, P, I, I, I, I, I, I, I, P, P, I, P, P, I, P, P]),
    "bcp_conv3_fwd_raw": (I, [P, P, P, I, I, I, I, I, I, I, I, P, P]),
    "bcp_conv3_packed_weight_floats": (SZ, [I, I, I]),
    "bcp_conv3_fwd_path": (SZ, [I, I, I, I, I, I, I]),
    "bcp_conv3_planes": (I, [I, I, I, I, I, I, I, I]),
    "bcp_conv3_wgrad_path": (SZ, [I, I, I, I, I, I, I]),
    "bcp_conv3_pack_weight": (I, [P, P, P, I, I, I, P]),
    "bcp_conv3_pack_many": (I, [P, I, P]),
    "bcp_conv3_last_section": (I, []),
    "bcp_conv3_fwd_workspace_bytes": (SZ, [I, I, I, I, I, I, I]),
    "bcp_conv3_fwd": (I, [P, P, P, P, I, I, I, I, I, I, I, I, P, P, P]),
    "bcp_conv3_stat_rows": (I, [I, I, I, I, I, I, I, I, I]),
    "bcp_conv3_fwd_stats": (I, [P, P, P, P, I, I, I, I, I, I, I, P, P, I, P, P]),
    "bcp_conv3_wgrad_workspace_bytes": (SZ, [I, I, I, I, I, I, I]),
    "bcp_conv3_wgrad": (I, [P, P, P, I, I, I, I, I, I, I, I, P, P, P, P]),
    "bcp_conv3_c1_fwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "bcp_conv3_c1_stat_rows": (I, [I, I, I, I, I, I]),
    "bcp_conv3_c1_fwd_stats": (I, [P, P, P, P, I, I, I, I, I, P, I, P]),
    "bcp_conv3_c1_norm_workspace_bytes": (SZ, [I, I, I, I, I, I]),
    "bcp_conv3_c1_norm_fwd": (I, [P, P, P, I, I, I, I, I, I, P, P, P, P, F, F, I, P, F, P, F, P, P, P, P, P]),
    "bcp_conv3_c1_norm_bwd": (I, [P, P, P, P, I, I, I, I, I, I, P, I, P, F, P, F, P, P, I, P, P, P]),
    "bcp_conv3_c1_norm_bwd_wgrad_workspace_bytes": (SZ, [I, I, I, I, I, I]),
    "bcp_conv3_c1_norm_bwd_wgrad": (I, [P, P, P, P, I, I, I, I, I, I, P, I, P, F, P, F, P, P, I, P, P, I, P]),
    "bcp_conv3_c1_wgrad": (I, [P, P, P, I, I, I, I, I, I, P, P]),
    "bcp_k2_pack_weight": (I, [P, P, I, I, I, P]),
    "bcp_k2_pack_desc": (I, [P, P, I, I, I, P]),
    "bcp_k2_pack_many": (I, [P, I, P]),
    "bcp_down_fwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "bcp_down_dgrad": (I, [P, P, P, I, I, I, I, I, I, I, P]),
    "bcp_up_fwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "bcp_up_dgrad": (I, [P, P, P, I, I, I, I, I, I, I, P]),
    "bcp_k2_stat_rows": (I, [I, I, I, I, I, I, I, I]),
    "bcp_down_fwd_stats": (I, [P, P, P, P, I, I, I, I, I, I, P, I, P]),
    "bcp_up_fwd_stats": (I, [P, P, P, P, I, I, I, I, I, I, P, I, P]),
    "bcp_k2_bwdstat_rows": (I, [I, I, I, I, I, I, I, I]),
    "bcp_down_dgrad_bwdstats": (I, [P, P, P, I, I, I, I, I, I, I, P, P, I, P, I, P]),
    "bcp_up_dgrad_bwdstats": (I, [P, P, P, I, I, I, I, I, I, I, P, P, I, P, I, P]),
    "bcp_up_norm_rows": (I, [I, I, I, I, I, I, I]),
    "bcp_up_norm_workspace_bytes": (SZ, [I, I, I, I, I, I, I]),
    "bcp_up_fwd_norm": (I, [P, P, P, I, I, I, I, I, I, I, P, P, P, P, F, F, I, P, P, P, P, P, P]),
    "bcp_up_norm_bwd": (I, [P, P, P, P, I, I, I, I, I, I, I, P, I, P, P, I, P, P, P]),
    "bcp_pw_fwd": (I, [P, P, P, P, L, I, I, P]),
    "bcp_tn_workspace_bytes": (SZ, [L, I, I]),
    "bcp_k2_wgrad": (I, [P, P, P, I, I, I, I, I, I, I, I, P, P]),
    "bcp_pw16_fwd": (I, [P, P, P, P, L, I, P]),
    "bcp_pw16_bwd": (I, [P, P, P, P, P, P, L, I, I, P, P]),
    "bcp_pw16_fwd_norm": (I, [P, P, P, I, I, I, P, P, P, L, I, P]),
    "bcp_pw16_bwd_norm": (I, [P, P, P, I, I, I, P, P, P, P, P, L, I, I, P, P]),
    "bcp_pw16_bwd_norm_bwd_workspace_bytes": (SZ, [I, I, L]),
    "bcp_pw16_bwd_norm_bwd": (I, [P, P, P, I, I, I, P, P, P, P, P, P, P, I, L, I, I, P, P, P]),
    "bcp_colsum": (I, [P, L, I, P, I, P, P]),
    "bcp_maxpool2d_fwd": (I, [P, I, P, I, I, I, I, P, P, P]),
    "bcp_maxpool3d_k3s2_fwd": (I, [P, P, I, I, I, I, I, P]),
    "bcp_maxpool2d_bwd": (I, [P, I, P, P, I, I, I, I, I, P, I, P]),
    "bcp_bilinear2x_fwd": (I, [P, P, I, I, I, I, I, I, P, P]),
    "bcp_bilinear2x_bwd": (I, [P, P, I, I, I, I, I, I, P]),
    "bcp_copy_channels": (I, [P, P, L, I, I, I, I, I, I, P, P, P]),
    "bcp_ema": (I, [P, P, L, C.c_double, P]),
    "bcp_sgd": (I, [P, P, P, P, L, F, F, F, F, I, C.c_double, P]),
    "bcp_adam": (I, [P, P, P, P, L, F, F, F, F, I, F, P]),
    "bcp_norm_eval": (I, [P, L, I, P, P, P, P, F, I, P, P, P]),
    "bcp_sw_accumulate": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "bcp_sw_finish": (I, [P, P, P, L, F, P]),
    "bcp_overlap_counts": (I, [P, P, L, I, P, P]),
    "bcp_crop_rotflip": (I, [P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "bcp_acdc_augment": (I, [P, P, I, I, I, I, I, I, P, I, I, P]),
    "bcp_cast": (I, [P, P, L, I, P]),
    "bcp_axpy": (I, [P, P, L, F, P]),
    "bcp_bernoulli": (I, [P, L, F, F, I, U64, P]),
    "bcp_bernoulli_dev": (I, [P, L, F, F, I, P, P]),
    "bcp_store_u64": (I, [P, I, P, P]),
    "bcp_graph_begin_capture": (I, [P]),
    "bcp_graph_end_capture": (I, [P, C.POINTER(P)]),
    "bcp_graph_launch": (I, [P, P]),
    "bcp_graph_destroy": (I, [P]),
    "bcp_replay_create": (I, [C.POINTER(P)]),
    "bcp_replay_add": (I, [P, P, C.c_char_p, P, I]),
    "bcp_replay_run": (I, [P]),
    "bcp_replay_run_timed": (I, [P, P, P, P]),
    "bcp_replay_count": (I, [P]),
    "bcp_replay_destroy": (I, [P]),
    "bcp_stream_wait_stream": (I, [P, P]),
}

EXPORTED_SYMBOLS = tuple(_SIGS.keys())


def _arg_class(t):
    if t is I:
        return "i"
    if t is L:
        return "l"
    if t is F:
        return "f"
    if t is C.c_double:
        return "d"
    if t is U64:
        return "u"
    if t is SZ:
        return "z"
    return "p"          # void*, char*, POINTER(...)


def shape_of(name):
    """argument-class string of an entry point ("ppplf..."): how bcp_replay_add (csrc/replay.hip) is told to call it"""
    return "".join(_arg_class(t) for t in _SIGS[name][1])


_PACK = {"i": "<q", "l": "<q", "p": "<Q", "u": "<Q", "z": "<Q", "d": "<d"}


def pack_slots(shape, args):
    """the 8-byte argument images bcp_replay_add copies (ints sign-extended, floats in the low four bytes)"""
    import struct
    out = bytearray()
    for c, v in zip(shape, args):
        if hasattr(v, "value"):          # a ctypes scalar / pointer object
            v = v.value
        if c == "f":
            out += struct.pack("<fI", float(v), 0)
        elif c == "d":
            out += struct.pack("<d", float(v))
        elif c in ("p", "u", "z"):
            out += struct.pack("<Q", int(v or 0) & 0xFFFFFFFFFFFFFFFF)
        else:
            out += struct.pack("<q", int(v))
    return bytes(out)


class BcpError(RuntimeError):
    pass


class Binding:
    """A loaded library with typed entry points; `call` raises BcpError on a non-zero status."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise BcpError(
                f"{path} is missing: the HIP extension has not been built.  Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the product path.")
        self.path = path
        self.cdll = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(self.cdll, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        self._status_fns = {n for n, (r, _) in _SIGS.items() if r is I and n not in ("bcp_version", "bcp_conv3_stat_rows", "bcp_comm_available", "bcp_replay_count", "bcp_norm_slabs_ok", "bcp_conv3_planes", "bcp_conv3_fwd_nslabs", "bcp_conv3_bwdstat_rows", "bcp_conv3_c1_stat_rows", "bcp_k2_stat_rows", "bcp_k2_bwdstat_rows", "bcp_conv3_last_section", "bcp_up_norm_rows")}
        self._fns = {n: (getattr(self.cdll, n), n in self._status_fns) for n in _SIGS}
        got = int(self.cdll.bcp_version())
        if got != ABI_VERSION:
            raise BcpError(f"{path}: ABI revision {got}, this binding was written against {ABI_VERSION} (include/bcp_hip.h) -- rebuild: "
                           "python -c 'import __graft_entry__ as g; g.build()'")
        self._rec = None          # a bcp_amd.plan.LaunchPlan while a network pass is being recorded
        self.options_epoch = 0    # bumped by set_option: cached shape queries (hip_ops.Ops._ws_bytes) are keyed on it

    def last_error(self) -> str:
        return self.cdll.bcp_last_error().decode("utf-8", "replace")

    def set_option(self, name: str, value="") -> None:
        """tuning / test switch of the library (bcp_set_option); value: int, sequence of ints, or "" = default"""
        if isinstance(value, (tuple, list)):
            value = ",".join(str(int(v)) for v in value)
        self.call("bcp_set_option", name.encode(), str(value).encode())
        self.options_epoch += 1
        from . import plan
        plan.invalidate_all()      # recorded launch plans carry the kernel choices and workspace sizes of the old options

    def call(self, name: str, *args):
        fn, is_status = self._fns[name]
        rc = fn(*args)
        if rc and is_status:
            raise BcpError(f"{name} failed ({rc}): {self.last_error()}")
        if is_status and self._rec is not None:       # launches only: size / shape queries are pure
            self._rec.add_call(name, fn, args)
        return rc

    def check_replayed(self, rc):
        raise BcpError(f"a replayed launch failed ({rc}): {self.last_error()}")


_product = None


def product() -> Binding:
    """The gfx950 library.  Loud failure when absent."""
    global _product
    if _product is None:
        _product = Binding(LIB_PATH)
    return _product
