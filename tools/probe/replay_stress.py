"""Stress harness (GPU) for the round-4 red test: is a self-training step ONE function of (weights, inputs, masks, seeds) whatever runs
beside it?  Runs the small ACDC / LA / pancreas step of tests/net_checks.py:check_launch_plans many times from identical seeds in one of
the host modes (eager Python path | recorded passes replayed from C | HIP graphs), optionally with a LOAD GENERATOR on a third stream,
and compares every run with a serial reference (eager, teacher on the student's stream, weight gradients on the main stream).

  python tools/probe/replay_stress.py --what acdc --mode replay --runs 100 --load 1 --deep 1

--deep 1 (replay only): after every step the bits of EVERY tensor the plans keep (all intermediates of the teacher forward, the student
forward and the student backward: a plan never frees) are hashed; a deviating run is diffed against a clean one and the first divergent
tensor is printed with the launch that produced it.  The plan tensors are zeroed after the recording step so that never-written padding
compares equal across runs.
Switches for the bisection: --amax 0 (no |max| slots: three-plane bf16 kernels everywhere), --overlap 0 (teacher on the main stream),
--wgrad 0 (no weight-gradient side stream), --opt conv3_f16=0,conv3_b6=0 (library options), --graphs 1|2, --pregraph 1 (capture and run a
graph-mode step sequence first, as the test file's order does), --serialize 1 (hipDeviceSynchronize after every step function).
"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np
import torch

import net_checks as NC
import bcp_oracle as O
from bcp_amd.hip_ops import Ops
from bcp_amd import plan, train_step
from bcp_amd.networks._hipnet import HipNet

ap = argparse.ArgumentParser()
ap.add_argument("--what", default="acdc")
ap.add_argument("--mode", default="replay", choices=("eager", "replay"))
ap.add_argument("--main", default="real", choices=("real", "null"))
ap.add_argument("--runs", type=int, default=50)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--load", type=int, default=0, help="0 none, 1 a burst enqueued on a third stream before every step, 2 a background thread launching all the time")
ap.add_argument("--deep", type=int, default=0)
ap.add_argument("--amax", type=int, default=1)
ap.add_argument("--overlap", type=int, default=1)
ap.add_argument("--wgrad", type=int, default=1)
ap.add_argument("--graphs", type=int, default=0)
ap.add_argument("--pregraph", type=int, default=0)
ap.add_argument("--opt", default="")
ap.add_argument("--zero", type=int, default=1, help="deep mode: zero the plan tensors after the recording step")
ap.add_argument("--tag", default="")
ap.add_argument("--dump", default="", help="deep mode: file that receives the GEMM / upsample / concat tensors (bad and clean run) of the first deviating step")
ap.add_argument("--attr", default="", help="network switches set on student and teacher, e.g. fuse_c1=0,skip_in_concat=0,inline_dropout=0")
ap.add_argument("--selftest", type=int, default=0)
ap.add_argument("--show", type=int, default=6, help="deviating runs to print in detail")
ap.add_argument("--showt", type=int, default=8, help="divergent tensors to print per deviating run")
ap.add_argument("--loadkind", default="both", choices=("both", "copy", "mm"))
ap.add_argument("--emu", type=int, default=0, help="dry run of the harness itself on the host simulator (CPU tensors, no streams)")
A = ap.parse_args()

if A.emu:
    from bcp_amd import _lib
    from bcp_amd.utils import BCP_utils as BU
    ops = Ops(_lib.Binding(os.path.join(os.getcwd(), "tests", "_emu", "libbcp_emu.so")), allow_cpu=True); dev = torch.device("cpu")
    A.main, A.load = "null", 0
    torch.cuda.synchronize = lambda *a, **k: None
else:
    ops = Ops.product(); dev = torch.device("cuda:0")
for kv in [s for s in A.opt.split(",") if s]:
    k, v = kv.split("=")
    ops.set_option(k, v)
Ops.AMAX = bool(A.amax)


# ---------------------------------------------------------------------------------------------------------------- load generator
class Load:
    def __init__(self):
        self.s = torch.cuda.Stream(device=dev)
        self.a = torch.randn(32 << 20, device=dev)           # 128 MB
        self.b = torch.empty_like(self.a)
        self.m = torch.randn(2048, 2048, device=dev)
        self.o = torch.empty_like(self.m)
        self.stop = False
        self.th = None

    def burst(self, n=8):
        with torch.cuda.stream(self.s):
            for i in range(n):
                if A.loadkind != "mm":
                    self.b.copy_(self.a)                          # HBM / L2 pressure
                if A.loadkind != "copy":
                    torch.mm(self.m, self.m, out=self.o)           # CU / matrix-pipe pressure
                if A.loadkind == "both":
                    self.a[: 1 << 20].add_(1.0)

    def start(self):
        def loop():
            while not self.stop:
                self.burst(2)
                time.sleep(0.0005)
        self.th = threading.Thread(target=loop, daemon=True); self.th.start()

    def finish(self):
        self.stop = True
        if self.th is not None:
            self.th.join()
        self.s.synchronize()


LOAD = Load() if A.load else None


# ---------------------------------------------------------------------------------------------------------------- one run
def bits(t):
    """exact 64-bit hash of a tensor's bytes (sum of its 32-bit words weighted by position parity; cheap and order-sensitive enough)"""
    v = t.detach().reshape(-1)
    if v.numel() == 0:
        return 0
    if getattr(t, "_is_amax", False) or (v.dtype == torch.float32 and v.numel() == 1024):
        v = v[::32].contiguous()                               # |max| slot buffers: 32 live floats, the rest never written
    u = v.view(torch.uint8)
    n4 = u.numel() // 4 * 4
    w = u[:n4].view(torch.int32).to(torch.int64)
    h = int(w.sum()) + 3 * int(w[::2].sum())
    if n4 < u.numel():
        h += int(u[n4:].to(torch.int64).sum())
    return h & 0xFFFFFFFFFFFFFFFF


def plan_tensors(pl):
    """[(first launch index, launch name, tensor)] of a plan's kept tensors, in production order"""
    first = {}
    for i, (fn, args, name) in enumerate(pl.entries):
        if name is None:
            continue
        for a in args:
            if isinstance(a, int) and a > (1 << 32) and a not in first:
                first[a] = (i, name)
    out = []
    for t in pl.keep:
        if not isinstance(t, torch.Tensor):
            continue
        i, name = first.get(t.data_ptr(), (1 << 30, "?"))
        out.append((i, name, t))
    out.sort(key=lambda e: e[0])
    return out


def all_plans(model, ema):
    r = []
    for who, m in (("teacher", ema), ("student", model)):
        st = m.__dict__.get("_plan_state")
        if st is None:
            continue
        for key, pl in st[1].items():
            r.append((who + ":" + key[0], pl))
    return r


def run_(mode, overlap, wgrad, deep, graphs=0):
    plan.ENABLED = (mode == "replay")
    g0, w0 = plan.GRAPHS, HipNet.overlap_wgrad
    plan.GRAPHS = graphs
    HipNet.overlap_wgrad = bool(wgrad)
    try:
        torch.manual_seed(5); np.random.seed(5)
        what = A.what
        if what == "acdc":
            P = O.init_params(O.unet_param_shapes(), seed=51, random_affine=True)
            model, ema = NC.make_unet(P, dev, ops), NC.make_unet(P, dev, ops)
            vol, lab = O.synth_acdc_batch(8, shape=(64, 64), seed=78)
        else:
            shape = (32, 32, 16) if what == "la" else (32, 32, 32)
            P = O.init_params(O.vnet_param_shapes(variant=what), seed=41, random_affine=True)
            model, ema = NC.make_vnet(P, dev, ops, what), NC.make_vnet(P, dev, ops, what)
            vol, lab = O.synth_la_batch(4, shape=shape, seed=77)
        for kv in [s for s in A.attr.split(",") if s]:
            k_, v_ = kv.split("=")
            setattr(model, k_, bool(int(v_))); setattr(ema, k_, bool(int(v_)))
        model.seed_dropout(11); ema.seed_dropout(12)
        for p in ema.parameters():
            p.detach_()
        vol, lab = vol.to(dev), lab.to(dev)
        opt = train_step.FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
        tl = []
        orig = ema.forward

        def fwd(*a, **k):
            r = orig(*a, **k)
            tl.append((r[0] if isinstance(r, (tuple, list)) else r).double().sum())      # teacher logits checksum, on the teacher's stream
            return r
        ema.forward = fwd
        out, deepout = [], []
        for it in range(A.steps):
            if LOAD is not None and A.load == 1:
                LOAD.burst()
            if what == "acdc":
                r = train_step.acdc_self_train_step(model, ema, opt, vol, lab, 4, box=(9, 13, 42, 42), overlap=bool(overlap))
                o1, o2 = r["out_unl"], r["out_l"]
            else:
                r = train_step.la_self_train_step(model, ema, opt, vol, lab, 2, box=(3, 5, 2, 21, 21, 10), variant=what,
                                                  connect_mode=2 if what != "la" else None, grouped=True, overlap=bool(overlap))
                o1, o2 = r["outputs_l"], r["outputs_u"]
            out.append((float(r["loss"]), int(r["plab_a"].sum()), int(r["plab_b"].sum()), float(o1.double().sum()), float(o2.double().sum()),
                        float(tl[-1]), bits(model.flat_state()), bits(ema.flat_state())))
            if deep:
                torch.cuda.synchronize()
                step = []
                for pname, pl in all_plans(model, ema):
                    for i, name, t in plan_tensors(pl):
                        v = t.detach().reshape(-1)
                        if v.dtype == torch.float32 and v.numel() == 1024:
                            v = v[::32]                                  # |max| slots: 32 live floats
                        step.append((pname, i, name, tuple(t.shape), str(t.dtype), v.cpu().clone()))
                deepout.append(step)
                if it == 0 and A.zero:
                    for pname, pl in all_plans(model, ema):
                        for i, name, t in plan_tensors(pl):
                            if t.data_ptr() not in (pl.static_in.data_ptr(), 0) and (pl.seed_dev is None or t.data_ptr() != pl.seed_dev.data_ptr()):
                                t.zero_()
                    torch.cuda.synchronize()
        return out, deepout
    finally:
        plan.ENABLED = True
        plan.GRAPHS = g0
        HipNet.overlap_wgrad = w0


def run(mode, overlap, wgrad, deep, main, graphs=0):
    if main == "real":
        s = torch.cuda.Stream(device=dev); s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            r = run_(mode, overlap, wgrad, deep, graphs)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize()
        return r
    r = run_(mode, overlap, wgrad, deep, graphs)
    torch.cuda.synchronize()
    return r


FIELDS = ("loss", "plab_a", "plab_b", "sum(out1)", "sum(out2)", "sum(teacher logits)", "bits(student state)", "bits(teacher state)")
t0 = time.time()
ref, _ = run("eager", 0, 0, 0, "null")
ref2, _ = run("eager", 0, 0, 0, "null")
print("reference (serial eager) reproducible:", ref == ref2, flush=True)
if A.pregraph:
    for g in (1, 2):
        r, _ = run("replay", 0, 1, 0, "real", graphs=g)
        print(f"pregraph graphs={g}: equals reference: {r == ref}", flush=True)
if LOAD is not None and A.load == 2:
    LOAD.start()
bad, first_clean_deep, shown = 0, None, 0
hist = {}
for k in range(A.runs):
    r, d = run(A.mode, A.overlap, A.wgrad, A.deep and A.mode == "replay", A.main, graphs=A.graphs)
    if A.selftest and first_clean_deep is not None:      # exercise the diff printer: pretend a few elements of two tensors went stale
        r = list(r); r[1] = (r[1][0] + 1.0,) + tuple(r[1][1:])
        for j in (3, len(d[1]) // 2):
            d[1][j][5][2:5] = d[0][j][5][2:5] if d[0][j][5].shape == d[1][j][5].shape else 0
    if r == ref:
        if d and first_clean_deep is None:
            first_clean_deep = d
        continue
    bad += 1
    for i, (x, y) in enumerate(zip(r, ref)):
        if x != y:
            f = [FIELDS[j] for j in range(len(x)) if x[j] != y[j]]
            hist[(i, f[0])] = hist.get((i, f[0]), 0) + 1
            if shown < 6:
                print(f"run {k}: first deviation at step {i}: {f}\n     got {x}\n     ref {y}", flush=True)
            break
    if d and first_clean_deep is not None and shown < A.show:
        done = False
        for si in range(1, len(d)):          # (step 0 = the recording step, before the plan tensors were zeroed: padding differs run to run)
            sa, sb = d[si], first_clean_deep[si]
            nd = [j for j, (ea, eb) in enumerate(zip(sa, sb)) if not torch.equal(ea[5], eb[5])]
            if not nd:
                continue
            print(f"     deep: step {si}: {len(nd)} of {len(sa)} plan tensors differ from the clean run; in launch order:", flush=True)
            for j in nd[:A.showt]:
                ea, eb = sa[j], sb[j]
                x, y = ea[5], eb[5]
                if x.dtype == torch.uint8 and x.numel() % 4 == 0 and ea[2] != "bcp_bernoulli_dev":
                    x, y = x.view(torch.float32), y.view(torch.float32)      # workspaces: mostly floats / doubles
                ne = (x != y) & ~((x != x) & (y != y))
                idx = ne.nonzero().reshape(-1)
                prev_same = d[si - 1][j][5] if x is ea[5] else d[si - 1][j][5].view(torch.float32)
                prev_clean = first_clean_deep[si - 1][j][5] if x is ea[5] else first_clean_deep[si - 1][j][5].view(torch.float32)
                xs, ys = x[idx], y[idx]
                stale = int((xs == prev_same[idx]).sum()) if prev_same.shape == x.shape else -1
                stale_c = int((xs == prev_clean[idx]).sum()) if prev_clean.shape == x.shape else -1
                dd = (xs.double() - ys.double()).abs()
                C_ = ea[3][-1] if len(ea[3]) > 1 else 1
                rows = (idx // C_).unique() if len(ea[3]) > 1 else idx
                print(f"       {ea[0]} launch #{ea[1]} {ea[2]} shape {ea[3]} {ea[4]}: {idx.numel()} of {x.numel()} elements differ, index range "
                      f"[{int(idx.min())}, {int(idx.max())}], {rows.numel()} rows (first {rows[:6].tolist()}), max |d| {float(dd.max()):.3e} (max |clean| {float(ys.double().abs().max()):.3e}); "
                      f"equal to the SAME run's previous-step value: {stale}, to the clean run's previous-step value: {stale_c}; zeros: {int((xs == 0).sum())}", flush=True)
                if idx.numel() <= 16:
                    print(f"         idx {idx.tolist()} got {xs.tolist()} clean {ys.tolist()}", flush=True)
            if A.dump and not os.path.exists(A.dump):
                which = sa[nd[0]][0]
                keep = {}
                for j, (ea, eb) in enumerate(zip(sa, sb)):
                    if ea[0] == which and ea[2] in ("bcp_pw_fwd", "bcp_norm_fwd", "bcp_copy_channels", "bcp_bilinear2x_fwd") and ea[4] == "torch.float32" and ea[5].numel() <= 600000:
                        keep[f"{j}|{ea[1]}|{ea[2]}|{ea[3]}"] = (ea[5].clone(), eb[5].clone())
                torch.save({"step": si, "plan": which, "tensors": keep}, A.dump)
                print(f"     deep: dumped {len(keep)} tensor pairs of plan {which} to {A.dump}", flush=True)
            done = True
            break
        if not done:
            print("     deep: no plan tensor differs in steps >= 1 (the deviation is outside the recorded passes)", flush=True)
    shown += 1
if LOAD is not None:
    LOAD.finish()
cfg = {k: v for k, v in vars(A).items()}
print(f"RESULT tag={A.tag} what={A.what} mode={A.mode} main={A.main} load={A.load} overlap={A.overlap} wgrad={A.wgrad} amax={A.amax} graphs={A.graphs} "
      f"pregraph={A.pregraph} opt={A.opt!r} deep={A.deep}: {bad} of {A.runs} runs deviate from the serial reference; first-deviation histogram {hist}; "
      f"{time.time() - t0:.0f} s", flush=True)
