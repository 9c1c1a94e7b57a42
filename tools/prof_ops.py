"""Launch every op family of the LA step at its in-step shape a few times, in a fixed order, for rocprofv3 --pmc passes
(tools/collect_pmc_ops.sh).  Between two ops a MARKER launch (k_ema on 64 elements) lets tools/pmc_ops_summary.py cut the
dispatch list into per-op segments, so counters are summed over ALL kernels an op launches (e.g. norm_fwd = statistics +
finalize + apply) and divided by the repetitions.  Writes the op list to <out>/ops.json.

  python tools/prof_ops.py <out_dir> [reps]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bcp_amd import hip_ops as H  # noqa: E402
from bcp_amd.hip_ops import Ops  # noqa: E402

out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_ops"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
os.makedirs(out_dir, exist_ok=True)
ops = Ops.product()
dev = torch.device("cuda:0")
mk_a, mk_b = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
N = 2
LEVELS = [(16, (112, 112, 80)), (32, (56, 56, 40)), (64, (28, 28, 20)), (128, (14, 14, 10)), (256, (7, 7, 5))]
plan = []


def run(key, fn, work):
    ops.ema(mk_a, mk_b, 0.5)   # marker: the warm-up segment starts (first-call allocations, weight packs -- not counted)
    fn()
    torch.cuda.synchronize()
    ops.ema(mk_a, mk_b, 0.5)   # marker: the measured segment starts
    for _ in range(reps):
        fn()
    ops.ema(mk_a, mk_b, 0.5)   # marker: the measured segment ENDS (round 5: what the script launches before the next op's first marker --
                               # torch.randn / fill / weight-pack launches of the next level's set-up -- used to be counted into this op:
                               # rounds 2-4 reported the last op of every level, and mix_box, with the next set-up's time and traffic)
    plan.append({"op": key, "reps": reps, **work})


for C, sp in LEVELS:
    x = torch.randn(N, *sp, C, device=dev)
    dy = torch.randn(N, *sp, C, device=dev)
    if os.environ.get("BCP_PROF_NO_AMAX") != "1":      # as in the step: the tensors carry their |max| (two-plane fp16 conv instances, round 4)
        x._bcp_amax = H.amax_slots(float(x.abs().max()), dev)
        dy._bcp_amax = H.amax_slots(float(dy.abs().max()), dev)
    w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
    b = torch.zeros(C, device=dev)
    wf, wd = ops.conv3_pack(w, 3)
    y = torch.empty(N, *sp, C, device=dev)
    dw = torch.empty_like(w)
    vox = N * sp[0] * sp[1] * sp[2]
    fl = 2.0 * vox * 27 * C * C
    tag = "x".join(str(v) for v in (N, *sp, C))
    run(f"conv3_fwd_stats[{tag}]", lambda: ops.conv3_fwd_stats(x, wf, b, C, 3, N), {"flop": fl, "bytes": 8.0 * vox * C})
    run(f"conv3_fwd[{tag}]", lambda: ops.conv3_fwd(dy, wd, None, C, 3, out=y), {"flop": fl, "bytes": 8.0 * vox * C})
    # (round 6: the gradient tensor itself counts -- 4 * 27 * C * C bytes, 7.1 MB at 256 channels against 1.0 MB of operands: rounds 2-5 priced the
    #  deep weight gradients against their operands only)
    run(f"conv3_wgrad[{tag}]", lambda: ops.conv3_wgrad(x, dy, dw, 3), {"flop": fl, "bytes": 8.0 * vox * C + 4.0 * 27 * C * C})
    g, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    a = torch.empty_like(y)
    st = [None]

    def nf():
        st[0] = ops.norm_fwd(x, N, g, be, rm, rv, H.ACT_RELU, out=a)[1]
    run(f"norm_fwd[{tag}]", nf, {"flop": 0, "bytes": 12.0 * vox * C})
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    run(f"norm_bwd[{tag}]", lambda: ops.norm_bwd(x, dy, N, st[0], H.ACT_RELU, dg, db, True, out=a), {"flop": 0, "bytes": 20.0 * vox * C})
    # round 3: the layer as the networks launch it now
    if ops.norm_slabs_ok(N, vox // N, C):
        sk = ops.conv3_nslabs(x.shape, C, 3)
        if sk > 1:
            # deep levels: conv leaves its split-K slabs, ONE norm kernel sums / finalises / applies
            def chain_f():
                st[0] = ops.norm_fwd_slabs(ops.conv3_fwd_raw(x, wf, C, 3, sk), sk, b, N, g, be, rm, rv, H.ACT_RELU)[1]
            run(f"conv3_raw+norm_slabs_fwd[{tag}]", chain_f, {"flop": fl, "bytes": 16.0 * vox * C})
            run(f"dgrad_raw+norm_slabs_bwd[{tag}]", lambda: ops.norm_bwd_slabs(x, ops.conv3_fwd_raw(dy, wd, C, 3, sk), sk, N, st[0], H.ACT_RELU, dg, db, True),
                {"flop": fl, "bytes": 16.0 * vox * C})
    else:
        def chain_b():
            da, part, rows = ops.conv3_dgrad_bwdstats(dy, wd, C, 3, x, st[0], H.ACT_RELU, N)
            ops.norm_bwd(x, da, N, st[0], H.ACT_RELU, dg, db, True, out=a, partial=part, nb=rows)
        run(f"dgrad_bwdstats+norm_bwd[{tag}]", chain_b, {"flop": fl, "bytes": 24.0 * vox * C})
        run(f"conv3_dgrad_bwdstats[{tag}]", lambda: ops.conv3_dgrad_bwdstats(dy, wd, C, 3, x, st[0], H.ACT_RELU, N), {"flop": fl, "bytes": 12.0 * vox * C})

sp = (112, 112, 80)
vox = N * sp[0] * sp[1] * sp[2]
x1 = torch.randn(N, *sp, 1, device=dev)
w1 = torch.randn(16, 1, 3, 3, 3, device=dev)
y16 = torch.empty(N, *sp, 16, device=dev)
run("conv3_c1_fwd[2x112x112x80x1]", lambda: ops.conv3_c1_fwd(x1, w1, None, 3, out=y16), {"flop": 2.0 * vox * 27 * 16, "bytes": 4.0 * vox * 17})
wdn = torch.randn(32, 16, 2, 2, 2, device=dev)
bp = ops.k2_pack(wdn, 16, 32, H.PACK_DOWN_FWD)
yd = torch.empty(N, 56, 56, 40, 32, device=dev)
run("down_fwd[2x112x112x80x16]", lambda: ops.down_fwd(y16, bp, None, 32, out=yd), {"flop": 2.0 * yd.numel() * 128, "bytes": 4.0 * (y16.numel() + yd.numel())})
wup = torch.randn(32, 16, 2, 2, 2, device=dev)
bpu = ops.k2_pack(wup, 32, 16, H.PACK_UP_FWD)
run("up_fwd[2x56x56x40x32]", lambda: ops.up_fwd(yd, bpu, None, 16, out=y16), {"flop": 2.0 * yd.numel() * 128, "bytes": 4.0 * (y16.numel() + yd.numel())})
# round 6: the transposed conv leaving its output's norm statistics (the statistics pass of the norm behind it is skipped) + that norm, and the
# plain chain it replaces
gu, beu, rmu, rvu = torch.ones(16, device=dev), torch.zeros(16, device=dev), torch.zeros(16, device=dev), torch.ones(16, device=dev)
au = torch.empty_like(y16)
if ops.k2_stat_rows(1, yd.shape, 16, N) > 0:
    def up_chain():
        yy, part, nb = ops.k2_fwd_stats(1, yd, bpu, None, 16, N)
        ops.norm_fwd(yy, N, gu, beu, rmu, rvu, H.ACT_RELU, out=au, partial=part, nb=nb)
    run("up_fwd_stats+norm_fwd[2x56x56x40x32]", up_chain, {"flop": 2.0 * yd.numel() * 128, "bytes": 4.0 * (yd.numel() + 3 * y16.numel())})
run("up_fwd+norm_fwd[2x56x56x40x32]", lambda: ops.norm_fwd(ops.up_fwd(yd, bpu, None, 16, out=y16), N, gu, beu, rmu, rvu, H.ACT_RELU, out=au),
    {"flop": 2.0 * yd.numel() * 128, "bytes": 4.0 * (yd.numel() + 4 * y16.numel())})
dwd = torch.empty_like(wdn)
run("k2_wgrad[2x112x112x80x16]", lambda: ops.k2_wgrad(y16, yd, dwd, H.WG_DOWN), {"flop": 2.0 * yd.numel() * 128, "bytes": 4.0 * (y16.numel() + yd.numel())})
wo = torch.randn(2, 16, 1, 1, 1, device=dev)
lo = torch.empty(N, *sp, 2, device=dev)
run("pw16_fwd[2x112x112x80x16]", lambda: ops.pw16_fwd(y16, wo, None, 2, out=lo), {"flop": 0, "bytes": 4.0 * (y16.numel() + lo.numel())})
st16 = ops.norm_fwd(y16, N, None, None, None, None, H.ACT_RELU, stats_only=True)[1]
run("pw16_fwd_norm[2x112x112x80x16]", lambda: ops.pw16_fwd_norm(y16, st16, None, N, H.ACT_RELU, wo, None, 2, out=lo), {"flop": 0, "bytes": 4.0 * (y16.numel() + lo.numel())})
dwo, dbo, dlo = torch.zeros_like(wo), torch.zeros(2, device=dev), torch.randn(N, *sp, 2, device=dev) * 1e-3
# round 5: the head's backward through the norm (stats pass + finalize kernels + apply pass; the activation gradient is recomputed) and the
# round-4 chain it replaces
run("pw16_bwd_norm_bwd[2x112x112x80x16]", lambda: ops.pw16_bwd_norm_bwd(y16, st16, None, N, H.ACT_RELU, dlo, wo, dwo, dbo), {"flop": 0, "bytes": 4.0 * (3 * y16.numel() + 2 * dlo.numel())})
run("pw16_bwd_norm+norm_bwd[2x112x112x80x16]", lambda: ops.norm_bwd(y16, ops.pw16_bwd_norm(y16, st16, None, N, H.ACT_RELU, dlo, wo, dwo, dbo), N, st16, H.ACT_RELU),
    {"flop": 0, "bytes": 4.0 * (3 * y16.numel() + 2 * dlo.numel())})
la = (torch.rand(N, *sp, device=dev) > 0.9).to(torch.uint8)
box = (10, 20, 5, 74, 74, 53)
ws3 = [None]


def mf():
    ws3[0] = ops.mixloss_fwd(lo, la, la, box, H.LOSS_LA, 1.0, 0.5)[1]
run("mixloss_fwd[2x112x112x80x2]", mf, {"flop": 0, "bytes": 10.0 * vox})
run("mixloss_bwd[2x112x112x80x2]", lambda: ops.mixloss_bwd(lo, la, la, box, H.LOSS_LA, ws3[0], 0.5, 0.5), {"flop": 0, "bytes": 18.0 * vox})
# (round 5: the optimiser tensors are made BEFORE the mix segment -- their four torch.randn launches used to fall between the mix op's
#  markers, and rounds 2-4 reported them as the mix kernel's time and traffic: "58.5 MB written for an 8 MB output" was 4 x 37.8 MB of randn / 3)
n = 9457318
p, gq, bu, em = (torch.randn(n, device=dev) for _ in range(4))
x1b, mixo = torch.randn_like(x1), torch.empty_like(x1)
torch.cuda.synchronize()
run("mix_box[2x112x112x80x1]", lambda: ops.mix_box(x1, x1b, box, out=mixo), {"flop": 0, "bytes": 4.0 * vox * (2 + 0.25)})      # a, out, b inside the box (~ a quarter of the volume)
run("sgd[9457318]", lambda: ops.sgd(p, gq, bu, 0.01, 0.9, 1e-4, False), {"flop": 0, "bytes": 20.0 * n})
run("ema[9457318]", lambda: ops.ema(em, p, 0.99), {"flop": 0, "bytes": 12.0 * n})
ops.ema(mk_a, mk_b, 0.5)   # closing marker
torch.cuda.synchronize()
json.dump(plan, open(os.path.join(out_dir, "ops.json"), "w"), indent=1)
print("ops:", len(plan))
