"""Kernel logic on the HOST simulator (tools/emu): the very kernel sources of bcp_amd/csrc compiled for
x86 and driven through the same C ABI + Python wrappers, CPU tensors.  This checks indexing, tiling,
MFMA fragment layouts (as documented for gfx950) and the reductions WITHOUT a GPU; the -m gpu tests
repeat the same checks on the real library.  Not a product path: nothing under bcp_amd/ loads the
simulator."""
import os
import subprocess

import pytest
import torch

import kernel_checks as K
from bcp_amd import _lib
from bcp_amd.hip_ops import Ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "_emu", "libbcp_emu.so")
EMU_OVERRIDE = os.environ.get("BCP_EMU_LIB")      # tools/emu/run_asan.sh: the AddressSanitizer build of the simulator


@pytest.fixture(scope="session")
def emu_ops():
    if EMU_OVERRIDE:
        return Ops(_lib.Binding(EMU_OVERRIDE), allow_cpu=True)
    srcs = [os.path.join(ROOT, "bcp_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "bcp_amd", "csrc")) if f.endswith((".hip", ".h"))]
    srcs += [os.path.join(ROOT, "tools", "emu", "emu_runtime.cpp"), os.path.join(ROOT, "tools", "emu", "hip", "hip_runtime.h")]
    if not os.path.exists(EMU) or any(os.path.getmtime(s) > os.path.getmtime(EMU) for s in srcs):
        subprocess.check_call([os.path.join(ROOT, "tools", "emu", "build_emu.sh")])
    return Ops(_lib.Binding(EMU), allow_cpu=True)


# The heaviest simulator runs are twins of checks `pytest -m gpu` executes on the device every round: skipped in the default CPU run
# (which must stay at a few minutes), BCP_EXTENDED=1 runs them.
EXTENDED_EMU = {"norm_slabs", "norm_own", "conv3_b6", "conv3_res", "dgrad_bwdstats"}


@pytest.mark.parametrize("name", [pytest.param(n, marks=pytest.mark.extended) if n in EXTENDED_EMU else n for n in K.ALL_CHECKS])
def test_emu(emu_ops, golden_dir, name):
    fn = getattr(K, "check_" + name)
    if name in ("diceloss_class", "plabel", "cc", "mixloss", "augment", "augment_acdc", "augment_pancreas"):
        fn(emu_ops, torch.device("cpu"), golden_dir)
    else:
        fn(emu_ops, torch.device("cpu"))


def test_abi_symbols_exported(emu_ops):
    """every symbol include/bcp_hip.h declares is exported (the product .so is checked in test_abi.py)"""
    import re
    hdr = open(os.path.join(ROOT, "include", "bcp_hip.h")).read()
    declared = set(re.findall(r"\b(bcp_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)


def test_pancreas_loader_streams(emu_ops):
    """the four loader streams of pancreas/dataloaders.py:185-195 over device-resident cases: forwards / reversed lists, RandomCrop for
    the labeled pair, CenterCrop for the unlabeled pair, labels stay integer"""
    import numpy as np
    from bcp_amd.utils import BCP_utils as BU
    from bcp_amd.pancreas.train_pancreas import _LoaderStreams
    from bcp_amd.pancreas.dataloaders import SyntheticPancreas
    BU.set_test_ops(emu_ops)
    np.random.seed(3)
    ls = _LoaderStreams(torch.device("cpu"), 1, n_cases=2)
    st = ls()
    assert tuple(st.vols.shape) == (4, 1, 96, 96, 96) and tuple(st.labs.shape) == (4, 96, 96, 96) and st.labs.dtype == torch.int64
    ds = SyntheticPancreas("train_unlab", "cpu", 2)
    a, _ = ds[0]
    assert torch.equal(st[2][0][0], a)                                  # unlab_a: case 0, centre crop (no random draw)
    rv = SyntheticPancreas("train_unlab", "cpu", 2, reverse=True)
    assert torch.equal(st[3][0][0], rv[0][0]) and torch.equal(rv[0][0], ds[1][0])   # unlab_b walks the list backwards
    assert len(SyntheticPancreas("train_lab", "cpu", 2)) == 20 and len(SyntheticPancreas("train_lab", "cpu", 2, labelp=20)) == 10


def test_replay_shapes_table_is_current():
    """bcp_amd/csrc/replay_shapes.inc is generated from the binding table: a changed signature must regenerate it"""
    import subprocess, sys
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_replay_shapes.py"), "--check"]) == 0


def test_replay_list_calls_with_typed_arguments(emu_ops):
    """bcp_replay_add / bcp_replay_run (csrc/replay.hip): pointer, long long, int, float and double arguments arrive intact, calls run
    in order, an unknown shape and a failing entry are reported"""
    import ctypes as C
    b = emu_ops.b
    y, x = torch.arange(8, dtype=torch.float32), torch.ones(8)
    p, q = torch.zeros(8), torch.full((8,), 2.0)
    h = C.c_void_p()
    assert b.cdll.bcp_replay_create(C.byref(h)) == 0
    for name, args in (("bcp_axpy", (y.data_ptr(), x.data_ptr(), 8, 0.5, None)),          # y += 0.5 x        (p p l f p)
                       ("bcp_axpy", (y.data_ptr(), x.data_ptr(), 8, -0.25, None)),        # y -= 0.25 x
                       ("bcp_ema", (p.data_ptr(), q.data_ptr(), 8, 0.75, None))):         # p = .75 p + .25 q  (p p l d p)
        shape = _lib.shape_of(name)
        fn = b._fns[name][0]
        assert b.cdll.bcp_replay_add(h, C.cast(fn, C.c_void_p), shape.encode(), _lib.pack_slots(shape, args), len(args)) == 0
    assert b.cdll.bcp_replay_count(h) == 3
    assert b.cdll.bcp_replay_run(h) == 0
    assert torch.equal(y, torch.arange(8, dtype=torch.float32) + 0.25) and torch.equal(p, torch.full((8,), 0.5))
    assert b.cdll.bcp_replay_run(h) == 0
    assert torch.equal(y, torch.arange(8, dtype=torch.float32) + 0.5) and torch.equal(p, torch.full((8,), 0.875))
    assert b.cdll.bcp_replay_add(h, C.cast(b._fns["bcp_axpy"][0], C.c_void_p), b"pq", b"\0" * 16, 2) != 0 and "shape" in b.last_error()
    bad = _lib.pack_slots("pplfp", (0, 0, 8, 1.0, None))                                 # null tensors: bcp_axpy refuses
    assert b.cdll.bcp_replay_add(h, C.cast(b._fns["bcp_axpy"][0], C.c_void_p), b"pplfp", bad, 5) == 0
    assert b.cdll.bcp_replay_run(h) != 0
    b.cdll.bcp_replay_destroy(h)
