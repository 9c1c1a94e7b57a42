"""tools/isa_scan.py -- instruction-level gates behind round 4's red test.

The cause, down to one instruction (round 5; tools/probe/pkmul_mfma_repro.hip is the stand-alone reproducer, tools/probe/isa_bisect.py the
bisection that found it inside the round-4 build of k_bilinear2x_fwd; MI355X gfx950 / ROCm 7.2):

    v_pk_mul_f32 D, S0, S1 op_sel:[0,1] op_sel_hi:[1,0]        (packed fp32 multiply, the halves of the SECOND source crossed)

returns D.lo = +-0 in lanes 48..63 whenever a wave of ANOTHER kernel on the same CU is issuing dense 16-bit MFMAs
(v_mfma_f32_16x16x32_bf16 / _f16): 1908 of 1920 launches beside a pure MFMA loop, 0 of 1920 with the operands in plain order, 0 beside fp32
MFMAs / LDS-DMA / LDS / VALU neighbours.  The condition (tools/probe/pkmov_hazard.py variants 8-17): a packed fp32 multiply or fma whose LOW
result takes its first multiplier input from a low half and its second from a HIGH half (op_sel:[0,1,..]); S1.lo broadcast (op_sel_hi only),
the first source crossed, both crossed, packed adds: unaffected.  hipcc's SLP vectoriser emits the crossed form for float2 shuffles.  Two gates
(the first forbids every op_sel bit on a multiplier input, a superset of the measured condition):

  1. THE BUILT LIBRARY (what ships): no v_pk_mul_f32 / v_pk_fma_f32 anywhere whose LOW result takes the HIGH half of a multiplier input
     (op_sel bit of source 0 or 1 set) -- disassembly of bcp_amd/csrc/libbcp_hip.so, two seconds; __graft_entry__.build() runs it.
  2. the plumbing kernels (pool2d.hip, elementwise.hip, eval.hip, gemm.hip: HBM-bound streams and latency-bound small GEMMs that run beside the
     convs and gain nothing from packed arithmetic) are compiled with -fno-slp-vectorize and must contain NO swizzled packed-fp32 instruction
     at all (v_pk_mov_b32, v_pk_*_f32 with op_sel:[..]) -- checked on their assembly.

  python tools/isa_scan.py            # both gates: exit 1 on a violation
  python tools/isa_scan.py --report   # + per-kernel census of every swizzled packed form over the whole library (assembly of every source)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bcp_amd", "csrc")
SWIZZLED = re.compile(r"^\s*(v_pk_mov_b32\b.*|v_pk_\w+_f32\b.*\bop_sel:\[)")
LABEL = re.compile(r"^([A-Za-z_][\w$]*):")            # a function label (not a .L local label, not a "; %bb.N:" comment)
PLUMBING = ("pool2d", "elementwise", "eval", "gemm")


def _flags(name):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    return list(g.EXTRA_FLAGS.get(name, []))


def assembly(names, outdir=None):
    outdir = outdir or os.environ.get("BCP_ISA_DIR", "/tmp/bcp_isa")
    os.makedirs(outdir, exist_ok=True)
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + [os.path.join(ROOT, "__graft_entry__.py")]
    procs, outs = [], {}
    for n in names:
        src, o = os.path.join(CSRC, n + ".hip"), os.path.join(outdir, n + ".s")
        outs[n] = o
        if not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(p) for p in [src] + deps):
            procs.append(subprocess.Popen([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + _flags(n) +
                                          ["--cuda-device-only", "-S", "-o", o, src], stderr=subprocess.DEVNULL))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc -S failed")
    return outs


def swizzled_sites(path):
    """[(kernel symbol, line number, instruction)] of the swizzled packed-fp32 instructions in one assembly file"""
    out, kernel = [], "?"
    for i, ln in enumerate(open(path).read().splitlines()):
        m = LABEL.match(ln)
        if m:
            kernel = m.group(1)
        s = ln.split(";")[0]
        if SWIZZLED.match(s):
            out.append((kernel, i + 1, s.strip()))
    return out


def gate():
    bad = []
    for n, path in assembly(PLUMBING).items():
        bad += [(n,) + s for s in swizzled_sites(path)]
    return bad


CROSSED = re.compile(r"^\s*(v_pk_(?:mul|fma)_f32)\b.*\bop_sel:\[([01]),([01])")       # op_sel bits of source 0 and source 1
LIB = os.path.join(CSRC, "libbcp_hip.so")


class ToolUnavailable(RuntimeError):
    """llvm-objdump is missing or cannot unpack offload bundles: the gate could not RUN (callers decide; build() wants BCP_SKIP_ISA_GATE=1)"""


ANY_SEL = re.compile(r"^\s*v_pk_(?:mul|fma)_f32\b.*\bop_sel")       # a packed fp32 multiply / fma with ANY operand selection (op_sel / op_sel_hi)


def lib_gate(lib=LIB, report=None):
    """[(kernel, instruction)] of the packed multiplies / fmas of the BUILT library whose low result takes the high half of a multiplier
    input (the form that fails beside 16-bit MFMAs); the library's code objects are extracted and disassembled in a scratch directory.
    report: a list that receives (kernel, instruction) of EVERY packed fp32 multiply / fma carrying an operand selection of any kind --
    not failures (the forms found wrong are the gate's), the watch list VERDICT r05 item 9 asked for: what a stack update could turn"""
    import shutil
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        objdump = shutil.which("llvm-objdump") or objdump
    out = []
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(lib, os.path.join(d, "lib.so"))
        try:
            subprocess.run([objdump, "--offloading", "lib.so"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        except (FileNotFoundError, subprocess.CalledProcessError) as e:
            raise ToolUnavailable(f"{objdump} --offloading: {e}")          # NOT the same thing as "the instruction was found" 
        cos = [f for f in os.listdir(d) if "amdgcn" in f]
        if not cos:
            raise RuntimeError("no gfx950 code object found inside " + lib)
        for f in sorted(cos):
            kernel = "?"
            dis = subprocess.run([objdump, "-d", f], cwd=d, capture_output=True, text=True, check=True).stdout
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
                if m:
                    kernel = m.group(1)
                    continue
                s = ln.split("//")[0]
                c = CROSSED.match(s)
                if c and (c.group(2) == "1" or c.group(3) == "1"):
                    out.append((kernel, s.strip()))
                if report is not None and ANY_SEL.match(s):
                    report.append((kernel, s.strip()))
    return out


def main(argv):
    watch = []
    crossed = lib_gate(report=watch) if os.path.exists(LIB) else []
    for kernel, ins in crossed:
        print(f"libbcp_hip.so: {kernel[:90]}: {ins}")
    print(f"gate 1: {len(crossed)} packed fp32 multiply / fma with a crossed multiplier input in the built library" +
          ("" if os.path.exists(LIB) else " (library not built: skipped)"))
    by_kernel = {}
    for kernel, ins in watch:
        by_kernel[kernel] = by_kernel.get(kernel, 0) + 1
    print(f"watch list (reported, not failed): {len(watch)} packed fp32 multiplies / fmas with an operand selection of any kind in {len(by_kernel)} kernels")
    forms = {}
    for _, ins in watch:
        key = ins.split()[0] + " " + " ".join(t for t in ins.split() if t.startswith("op_sel"))
        forms[key] = forms.get(key, 0) + 1
    for k, n in sorted(forms.items(), key=lambda kv: -kv[1]):
        print(f"    {n:5d}  {k}      (low result from low halves: the form measured 0-wrong beside 16-bit MFMAs, DESIGN.md section 4.0)")
    if "--report" in argv:
        for k, n in sorted(by_kernel.items(), key=lambda kv: -kv[1])[:40]:
            print(f"    {n:5d}  {k[:120]}")
    bad = gate()
    for n, kernel, line, ins in bad:
        print(f"{n}.hip: {kernel[:80]} line {line}: {ins}")
    print(f"gate 2: {len(bad)} swizzled packed-fp32 instruction(s) in the plumbing kernels ({', '.join(p + '.hip' for p in PLUMBING)})")
    if "--report" in argv:
        names = sorted(f[:-4] for f in os.listdir(CSRC) if f.endswith(".hip"))
        for n, path in assembly(names).items():
            per = {}
            for kernel, _, ins in swizzled_sites(path):
                per.setdefault(kernel, []).append(ins.split()[0])
            for kernel, ops in sorted(per.items(), key=lambda e: -len(e[1])):
                print(f"  {n}.hip {kernel[:90]}: {len(ops)} ({', '.join(sorted(set(ops)))})")
    return 1 if (bad or crossed) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
