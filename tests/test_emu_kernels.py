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


@pytest.fixture(scope="session")
def emu_ops():
    srcs = [os.path.join(ROOT, "bcp_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "bcp_amd", "csrc")) if f.endswith((".hip", ".h"))]
    srcs += [os.path.join(ROOT, "tools", "emu", "emu_runtime.cpp"), os.path.join(ROOT, "tools", "emu", "hip", "hip_runtime.h")]
    if not os.path.exists(EMU) or any(os.path.getmtime(s) > os.path.getmtime(EMU) for s in srcs):
        subprocess.check_call([os.path.join(ROOT, "tools", "emu", "build_emu.sh")])
    return Ops(_lib.Binding(EMU), allow_cpu=True)


@pytest.mark.parametrize("name", K.ALL_CHECKS)
def test_emu(emu_ops, golden_dir, name):
    fn = getattr(K, "check_" + name)
    if name in ("plabel", "cc", "mixloss", "augment", "augment_acdc"):
        fn(emu_ops, torch.device("cpu"), golden_dir)
    else:
        fn(emu_ops, torch.device("cpu"))


def test_abi_symbols_exported(emu_ops):
    """every symbol include/bcp_hip.h declares is exported (the product .so is checked in test_abi.py)"""
    import re
    hdr = open(os.path.join(ROOT, "include", "bcp_hip.h")).read()
    declared = set(re.findall(r"\b(bcp_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
