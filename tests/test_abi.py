"""CPU-only: the product library libbcp_hip.so loads (dlopen needs no GPU) and exports every symbol that
include/bcp_hip.h declares; argument validation works without launching anything; the product loader
refuses to run without the library (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "bcp_amd", "csrc", "libbcp_hip.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import sys
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build()
    return ctypes.CDLL(LIB)


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "bcp_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(bcp_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 45
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    from bcp_amd import _lib
    assert set(declared) == set(_lib.EXPORTED_SYMBOLS)


def test_version_and_argument_validation(lib):
    from bcp_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "bcp_hip.h")).read()
    assert lib.bcp_version() == int(re.search(r"#define BCP_ABI_VERSION (\d+)", hdr).group(1)) == _lib.ABI_VERSION
    lib.bcp_last_error.restype = ctypes.c_char_p
    # null pointers / bad shapes are rejected before any launch (no GPU needed)
    rc = lib.bcp_mix_box(None, None, None, 1, 1, 1, 1, 1, None, None)
    assert rc == -1 and b"null" in lib.bcp_last_error()
    lib.bcp_norm_workspace_bytes.restype = ctypes.c_size_t
    lib.bcp_norm_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_int]
    assert lib.bcp_norm_workspace_bytes(1, 1003520, 16) > 0
    lib.bcp_conv3_packed_weight_floats.restype = ctypes.c_size_t
    assert lib.bcp_conv3_packed_weight_floats(16, 16, 3) == (27 + 3 * 14 + 2 * 14) * 16 * 16 + 32    # fp32 pack + three bf16 / two fp16 planes of 14 tap pairs + header


def test_product_ops_refuse_cpu_tensors():
    from bcp_amd import _lib
    from bcp_amd.hip_ops import Ops
    ops = Ops(_lib.Binding(LIB), allow_cpu=False)
    with pytest.raises(_lib.BcpError, match="no CPU fallback"):
        ops.mix_box(torch.zeros(1, 1, 4, 4, 1), torch.zeros(1, 1, 4, 4, 1), (0, 0, 0, 1, 1, 1))


def test_missing_library_is_loud(tmp_path):
    from bcp_amd import _lib
    with pytest.raises(_lib.BcpError, match="has not been built"):
        _lib.Binding(str(tmp_path / "libbcp_hip.so"))
