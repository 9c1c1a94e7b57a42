"""tools/isa_scan.py -- instruction-level gate behind round 4's red test.

What was measured (round 5, tools/probe/bilinear_race_probe.py, MI355X / ROCm 7.2): the round-4 build of k_bilinear2x_fwd returned wrong
values for 8-16 consecutive lanes (one input corner's dwords 1 and 3 counted as zero) in 4-7 % of its launches whenever the bf16-pipe convs
(MFMA + LDS-DMA) shared its CUs under a copy / GEMM load, and in none otherwise.  The same source compiled without the SLP vectoriser (no
packed-fp32 instruction in the kernel) or with this round's |max| expression (packed fma / mul in their plain forms only): 0 of 7680.  A full
s_waitcnt vmcnt(0) in front of the consumers did not help; what the failing build had and no passing build has are the SWIZZLED packed forms --
v_pk_mov_b32 ... op_sel:[1,0] and v_pk_mul_f32 / v_pk_fma_f32 with op_sel:[..] -- so those are what this gate keeps out of the kernels that
run beside the convs and do not need them (pool2d.hip, elementwise.hip, eval.hip, gemm.hip: HBM-bound streams and latency-bound small GEMMs,
compiled with -fno-slp-vectorize).

  python tools/isa_scan.py            # gate: exit 1 if a plumbing kernel contains a swizzled packed-fp32 instruction
  python tools/isa_scan.py --report   # + per-kernel census of those forms over the whole library (information: conv / GEMM epilogues have some)
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


def main(argv):
    bad = gate()
    for n, kernel, line, ins in bad:
        print(f"{n}.hip: {kernel[:80]} line {line}: {ins}")
    print(f"gate: {len(bad)} swizzled packed-fp32 instruction(s) in the plumbing kernels ({', '.join(p + '.hip' for p in PLUMBING)})")
    if "--report" in argv:
        names = sorted(f[:-4] for f in os.listdir(CSRC) if f.endswith(".hip"))
        for n, path in assembly(names).items():
            per = {}
            for kernel, _, ins in swizzled_sites(path):
                per.setdefault(kernel, []).append(ins.split()[0])
            for kernel, ops in sorted(per.items(), key=lambda e: -len(e[1])):
                print(f"  {n}.hip {kernel[:90]}: {len(ops)} ({', '.join(sorted(set(ops)))})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
