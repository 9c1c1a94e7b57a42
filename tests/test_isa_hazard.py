"""CPU-only (hipcc cross-compiles): the instruction-level gate of DESIGN.md section 4 -- the memory-bound plumbing kernels that run beside the
bf16-pipe convs contain no swizzled packed-fp32 instruction (v_pk_mov_b32, v_pk_*_f32 with op_sel:[..]): the round-4 build of
k_bilinear2x_fwd had them and returned wrong values under load (tools/probe/bilinear_race_probe.py)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="needs hipcc")
def test_plumbing_kernels_have_no_swizzled_packed_fp32(tmp_path):
    import isa_scan
    os.environ["BCP_ISA_DIR"] = str(tmp_path)
    try:
        bad = isa_scan.gate()
    finally:
        os.environ.pop("BCP_ISA_DIR", None)
    assert not bad, bad[:5]


def test_built_library_has_no_packed_multiply_with_a_crossed_multiplier_input():
    """gate 1: the instruction tools/probe/pkmul_mfma_repro.hip shows returning +-0 beside 16-bit MFMAs is nowhere in libbcp_hip.so"""
    import isa_scan
    if not os.path.exists(isa_scan.LIB):
        pytest.skip("libbcp_hip.so not built")
    watch = []
    assert isa_scan.lib_gate(report=watch) == []
    # the watch list: every packed fp32 multiply / fma with an operand selection left in the library takes its LOW result
    # from low halves (op_sel all zero) -- the forms tests/test_gpu_kernels.py::test_packed_fp32_hazard_canary keeps measuring on the device
    forms = {" ".join(t for t in ins.split() if t.startswith("op_sel")) for _, ins in watch}
    print(f"{len(watch)} packed fp32 multiplies / fmas with an operand selection: {sorted(forms)}")
    assert all("op_sel:" not in f for f in forms), forms
    assert forms <= {"op_sel_hi:[0,1]", "op_sel_hi:[1,0]", "op_sel_hi:[0,1,1]", "op_sel_hi:[1,0,1]"}, \
        f"a packed form the canary does not measure yet: {forms} (add it to tools/probe/pkmul_mfma_repro.hip)"
    assert isa_scan.CROSSED.match("\tv_pk_mul_f32 v[24:25], v[20:21], v[22:23] op_sel:[0,1] op_sel_hi:[1,0]").group(3) == "1"
    assert isa_scan.CROSSED.match("\tv_pk_fma_f32 v[24:25], v[10:11], v[34:35], v[24:25] op_sel:[0,0,1] op_sel_hi:[1,1,0]").groups()[1:] == ("0", "0")


def test_scanner_recognises_the_round4_forms(tmp_path):
    import isa_scan
    p = tmp_path / "k.s"
    p.write_text("_Zk:\n\tv_pk_mov_b32 v[16:17], v[24:25], v[16:17] op_sel:[1,0]\n\tv_pk_mul_f32 v[18:19], v[18:19], v[14:15] op_sel:[0,1] op_sel_hi:[1,0]\n"
                 "\tv_pk_fma_f32 v[16:17], v[16:17], v[32:33], v[24:25] op_sel_hi:[1,1,0]\n\tv_pk_add_f32 v[0:1], v[0:1], v[2:3]\n\ts_endpgm\n")
    got = [ins.split()[0] for _, _, ins in isa_scan.swizzled_sites(str(p))]
    assert got == ["v_pk_mov_b32", "v_pk_mul_f32"]
