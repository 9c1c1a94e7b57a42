#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric: training volumes/sec of the BCP self-training step on the LA 3-D V-Net
(112x112x80 patches), per-GPU batch 4 (2 labeled + 2 unlabeled; BASELINE.json configs[1]), synthetic data,
fp32, random-init weights, everything inside the timed region that the reference's loop does per iteration
(LA_BCP_train.py:235-270): teacher forward x2, pseudo-label + largest-CC x2, box draw, copy-paste mix x2,
student forward x2, masked Dice+CE x2, backward, [gradient all-reduce when N>1], SGD, EMA.

  python bench.py [--gpus N --steps K --warmup W] [--workload la|acdc|pancreas]
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

stdout's LAST line (rank 0) is ONE compact JSON record (< 4 KB, asserted): the contract keys only --
  metric value unit n_gpus steps warmup ms_per_step higher_is_better scaling vs_baseline dtype data config{workload, global_batch, parallelism}
  roofline{bound, kernel, achieved, peak, unit, frac, traffic, ...}   the MFMA-bound op with the LARGEST share of the step (not a hand-picked
                layer): `frac` from the mean duration of its recorded launches between two HIP events on its stream (agrees with the
                rocprofv3 kernel trace of the same command, profiles/); `frac_in_step_bracket` = the same from an event pair around the op
                inside the busy step (includes 10-20 us of marker latency); `traffic` = HBM bytes per launch from the committed --pmc passes
  cpu_baseline{value, unit, cores, kind, sample}   the oracle (CPU restatement of the reference, oracle/bcp_oracle.py) on this box's host
                cores: median of >= 5 timed steps after 2 warm-ups (rank 0, N = 1 only)
  extra_workloads{acdc, pancreas, la_b8}{value, unit, ms_per_step, roofline_frac, cpu_baseline_value}   the north_star's other single-GPU lines
  ranks_seen / exposed_allreduce_ms_per_step   N > 1 only
Everything else -- the per-op table (`kernels`), `roofline_hbm`, host enqueue times, provenance prose -- is written to bench_detail.json
(repo root, and gpurun_out/ when that directory exists) and summarised on stderr.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0      # same guide: dense BF16 MFMA peak
# csrc/conv3b.hip computes the fp32 convolution with SIX bf16 MFMAs per K = 32 block (three-piece operands, fp32 accumulate):
# its roofline in fp32-equivalent TFLOP/s is the bf16 peak / 6
PEAK_BF16X3_F32EQ_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
PEAK_F16X2_F32EQ_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0      # round 4: two fp16 planes per operand, three MFMAs per product block (same MFMA rate as bf16)
PEAK_HBM_GBS = 8000.0               # same guide: HBM3E spec (6290 GB/s measured with a float4 copy)
# the arithmetic type of the path: fp32 tensors and fp32 accumulation; the 3x3(x3) convolutions take each fp32 operand as two fp16 planes
# (~22 mantissa bits, per-tensor power-of-two pre-scales) on the matrix cores -- said here, not only in the detail file
DTYPE = "f32 (fp16x2 split operands)"
STEP_GFLOP_PER_VOLUME = 160.0       # 1/2 teacher fwd + 1/2 student fwd+bwd per input volume


# as the training scripts run the networks (bcp_amd/LA_BCP_train.py, ACDC_BCP_train.py, pancreas/train_pancreas.py): results of a replayed pass
# are consumed before the network's next pass, so they are handed out as aliases and the producers write into the plans' own input tensors
# (networks/_hipnet.py volatile_io).  BCP_VOLATILE_IO=0: defensive copies in and out of every pass (measurement switch)
_VOLATILE_IO = os.environ.get("BCP_VOLATILE_IO", "1") != "0"


def build_models(dev, seed):
    from bcp_amd.networks.net_factory import net_factory
    torch.manual_seed(seed)
    model = net_factory(net_type="VNet", in_chns=1, class_num=2, mode="train")
    ema_model = net_factory(net_type="VNet", in_chns=1, class_num=2, mode="train")
    for p in ema_model.parameters():
        p.detach_()
    ema_model.load_state_dict(model.state_dict())   # the reference loads both from the same checkpoint (LA:220-222)
    model.train()
    ema_model.train()
    return model, ema_model


# ------------------------------------------------------------------------------------------------ per-op table
def _numel(s):
    n = 1
    for v in s:
        n *= v
    return n


def _work(name, shapes, ints):
    """algorithmic work of one call: (bound, flops, bytes).  SURVEY.md 8d / DESIGN.md section 3: every tensor read / written once"""
    s0 = shapes[0] if shapes else ()
    if name in CONV3_FWD_OPS:                              # (x [N,D,H,W,Cin], wp, [bias | y_prev]); ints = (Cout, KD, ...)
        cout, kd = ints[0], ints[1]
        vox = _numel(s0[:-1])
        extra = vox * cout if name == "conv3_dgrad_bwdstats" else 0      # the epilogue also reads the consumer norm layer's y
        return "mfma", 2.0 * vox * kd * 9 * s0[-1] * cout, 4.0 * (_numel(s0) + vox * cout + extra)
    if name == "conv3_wgrad":                              # (x, dy, dw); ints = (KD,)
        vox = _numel(s0[:-1])
        return "mfma", 2.0 * vox * ints[0] * 9 * s0[-1] * shapes[1][-1], 4.0 * (_numel(s0) + _numel(shapes[1]) + ints[0] * 9 * s0[-1] * shapes[1][-1])      # x, dy and dW
    if name in ("conv3_c1_fwd", "conv3_c1_fwd_stats", "conv3_c1_wgrad"):         # Cin = 1 -> 16: HBM-bound (AI 12.7)
        vox = _numel(s0[:-1])
        return "hbm", 2.0 * vox * ints[0] * 9 * 16, 4.0 * vox * 17
    if name == "conv3_c1_norm_fwd":                        # first layer + norm + activation with recompute: x [N,D,H,W,1] is read by the statistics
        vox = _numel(s0[:-1])                              # pass and again by the apply pass, the 16-channel activation is written once
        return "hbm", 2.0 * 2.0 * vox * ints[0] * 9 * 16, 4.0 * vox * (2 + 16)
    if name == "conv3_c1_norm_bwd":                        # (x, ..., da [.,16]): statistics pass + apply pass both read x and da, dy [.,16] written once
        vox = _numel(s0[:-1])
        return "hbm", 2.0 * 2.0 * vox * ints[0] * 9 * 16, 4.0 * vox * (2 + 2 * 16 + 16)
    if name == "conv3_c1_norm_bwd_wgrad":                  # the same two passes, dy stays in LDS and feeds the layer's weight gradient (a third product)
        vox = _numel(s0[:-1])
        return "hbm", 3.0 * 2.0 * vox * ints[0] * 9 * 16, 4.0 * vox * (2 + 2 * 16)
    if name in ("down_fwd", "up_fwd", "down_dgrad", "up_dgrad", "pw_fwd"):   # k2s2 / 1x1 GEMMs: (x, packed B, [bias]); ints = (Cout,)
        cout = ints[0]
        vin = _numel(s0[:-1])
        k = 1 if name == "pw_fwd" else 8
        vout = vin * 8 if name in ("up_fwd", "down_dgrad") else (vin // 8 if name in ("down_fwd", "up_dgrad") else vin)
        fl = 2.0 * min(vin, vout) * k * s0[-1] * cout
        by = 4.0 * (_numel(s0) + vout * cout)
        return ("hbm" if fl / by < 20 else "mfma"), fl, by
    if name == "k2_fwd_stats":                             # (kind, x, packed B, bias, Cout, groups): the k2s2 / transposed conv + its output's norm statistics
        kind, cout = ints[0], ints[1]
        vin = _numel(s0[:-1])
        vout = vin * 8 if kind == 1 else vin // 8
        fl = 2.0 * min(vin, vout) * 8 * s0[-1] * cout
        by = 4.0 * (_numel(s0) + vout * cout)
        return ("hbm" if fl / by < 20 else "mfma"), fl, by
    if name == "k2_dgrad_bwdstats":                        # (kind, dy, packed B, Cin, y_prev, ...): the k2s2 / transposed conv dgrad + the previous norm's backward statistics
        kind, cin = ints[0], ints[1]
        vin = _numel(s0[:-1])
        vout = vin * 8 if kind == 0 else vin // 8
        fl = 2.0 * min(vin, vout) * 8 * s0[-1] * cin
        by = 4.0 * (_numel(s0) + 2 * vout * cin)
        return ("hbm" if fl / by < 20 else "mfma"), fl, by
    if name == "k2_wgrad":
        fl = 2.0 * min(_numel(s0[:-1]), _numel(shapes[1][:-1])) * 8 * s0[-1] * shapes[1][-1]
        by = 4.0 * (_numel(s0) + _numel(shapes[1]))
        return ("hbm" if fl / by < 20 else "mfma"), fl, by
    if name == "norm_fwd_slabs":                           # (slabs [nslab, ...] | y, ...); ints = (nslab, G, act): read the slabs, write y and a
        nsl = ints[0] if ints else 1
        n = _numel(s0) // (nsl if len(s0) == 6 else 1)
        return "hbm", 0.0, 4.0 * n * ((nsl if len(s0) == 6 else 1) + 2)
    if name == "norm_bwd_slabs":                           # (y, da slabs | da, ...): read y and the slabs, write dy
        s1 = shapes[1] if len(shapes) > 1 else s0
        return "hbm", 0.0, 4.0 * (2 * _numel(s0) + _numel(s1))
    if name == "norm_fwd":                                 # statistics (fused into the conv when possible) + apply: read y twice, write a
        return "hbm", 0.0, 12.0 * _numel(s0)
    if name == "norm_bwd":                                 # statistics pass over (y, da) + apply pass reading both, writing dy
        return "hbm", 0.0, 20.0 * _numel(s0)
    if name == "pw16_fwd":
        return "hbm", 0.0, 4.0 * (_numel(s0) + _numel(s0[:-1]) * (ints[0] if ints else 2))
    if name == "pw16_bwd":
        return "hbm", 0.0, 4.0 * (2 * _numel(s0) + _numel(shapes[1]))
    if name == "pw16_fwd_norm":                            # (y_raw, stats, [chan_scale]); ints = (G, act, Cout): read y, write logits
        return "hbm", 0.0, 4.0 * (_numel(s0) + _numel(s0[:-1]) * (ints[2] if len(ints) > 2 else 2))
    if name == "pw16_bwd_norm":                            # read y and dlogits, write d(activation)
        return "hbm", 0.0, 4.0 * (2 * _numel(s0) + _numel(s0[:-1]) * 2)
    if name == "pw16_bwd_norm_bwd":                        # two passes reading y and dlogits, one write of dy (the activation gradient is recomputed)
        return "hbm", 0.0, 4.0 * (3 * _numel(s0) + _numel(s0[:-1]) * 4)
    if name == "mixloss_fwd":
        return "hbm", 0.0, _numel(s0) * 4.0 + 2.0 * _numel(s0[:-1])
    if name == "mixloss_bwd":
        return "hbm", 0.0, _numel(s0) * 8.0 + 2.0 * _numel(s0[:-1])
    if name == "mix_box":
        return "hbm", 0.0, 12.0 * _numel(s0)
    if name == "plabel_bin":
        return "hbm", 0.0, 4.0 * _numel(s0) + _numel(s0[:-1])
    if name == "ema":
        return "hbm", 0.0, 12.0 * _numel(s0)
    if name == "sgd":
        return "hbm", 0.0, 20.0 * _numel(s0)
    if name == "adam":
        return "hbm", 0.0, 28.0 * _numel(s0)
    return None, 0.0, 0.0


CONV3_FWD_OPS = ("conv3_fwd", "conv3_fwd_stats", "conv3_fwd_raw", "conv3_dgrad_bwdstats")


def _conv3_path(xshape, ints, wgrad_cout=None, namax=0):
    """operand planes of the kernel the library serves this conv launch with: 0 = fp32 MFMA, 3 = three bf16 planes (six MFMAs per product
    block), 2 = two fp16 planes (three MFMAs) -- the latter only when the launch carried its operands' maxima (namax: forward / dgrad
    need 1, the weight gradient 2) and the shape has such an instance (bcp_conv3_planes; no launch)"""
    try:
        from bcp_amd.hip_ops import Ops
        N, D, H, W, Cin = xshape
        b = Ops.product().b
        if wgrad_cout is not None:
            on16 = int(b.call("bcp_conv3_wgrad_path", int(N), int(D), int(H), int(W), int(Cin), int(wgrad_cout), int(ints[0]))) == 1
            if not on16:
                return 0
            return 2 if (namax >= 2 and int(b.call("bcp_conv3_planes", int(N), int(D), int(H), int(W), int(Cin), int(wgrad_cout), int(ints[0]), 1)) == 2) else 3
        on16 = int(b.call("bcp_conv3_fwd_path", int(N), int(D), int(H), int(W), int(Cin), int(ints[0]), int(ints[1]))) == 1
        if not on16:
            return 0
        return 2 if (namax >= 1 and int(b.call("bcp_conv3_planes", int(N), int(D), int(H), int(W), int(Cin), int(ints[0]), int(ints[1]), 0)) == 2) else 3
    except Exception:
        return 0


def op_table(records, steps, step_ms, top=14):
    agg = {}
    for rec in records:
        name, shapes, ints, ms = rec[:4]
        namax = rec[4] if len(rec) > 4 else 0
        a = agg.setdefault((name, shapes[:2], ints[:2], namax), [0, 0.0, shapes, ints])
        a[0] += 1
        a[1] += ms
    rows = []
    for (name, _, _, namax), (calls, tot, shapes, ints) in agg.items():
        bound, fl, by = _work(name, shapes, ints)
        avg = tot / calls
        row = {"op": name, "shape": "x".join(str(v) for v in shapes[0]) if shapes else "", "launches_per_step": round(calls / steps, 2),
               "avg_us": round(avg * 1e3, 1), "ms_per_step": round(tot / steps, 4), "share_of_step": round(tot / steps / step_ms, 4), "bound": bound}
        if bound == "mfma":
            tf = fl / (avg * 1e-3) / 1e12
            pipe, peak = "f32", PEAK_F32_MFMA_TFLOPS
            planes = 0
            if name in CONV3_FWD_OPS and shapes and len(shapes[0]) == 5:
                planes = _conv3_path(shapes[0], ints, namax=namax)
            elif name == "conv3_wgrad" and len(shapes) > 1 and len(shapes[0]) == 5:
                planes = _conv3_path(shapes[0], ints, wgrad_cout=shapes[1][-1], namax=namax)
            if planes == 3:
                pipe, peak = "bf16x3 (fp32-equivalent, 6 bf16 MFMAs per product)", PEAK_BF16X3_F32EQ_TFLOPS
            elif planes == 2:
                pipe, peak = "f16x2 (fp32-equivalent, 3 fp16 MFMAs per product, per-tensor power-of-two pre-scales)", PEAK_F16X2_F32EQ_TFLOPS
            row.update({"flop_per_launch": fl, "achieved_tflops": round(tf, 2), "pipe": pipe, "peak_tflops": round(peak, 1), "frac": round(tf / peak, 4)})
        elif bound == "hbm":
            gbs = by / (avg * 1e-3) / 1e9
            row.update({"bytes_per_launch": by, "achieved_gbs": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4)})
        rows.append(row)
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows[:top], rows


def pmc_traffic(kernel_key):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes (tools/collect_pmc_ops.sh: FETCH_SIZE and WRITE_SIZE in
    separate runs; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note on 16-B/lane streams), next to the op's algorithmic
    bytes.  bench.py cannot run the profiler itself: the numbers are those of the commit the file's `_meta` names (the newest
    round's file first); null when profiles/ holds no entry for this op."""
    for fn in ("r06_pmc_ops.json", "r05_pmc_ops.json", "r04_pmc_ops.json", "r03_pmc_ops.json", "r02_pmc_ops.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", fn)))
        except Exception:
            continue
        e = d.get(kernel_key)
        if e and "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            by = int((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
            out = {"bytes_per_launch": by, "fetch_kb_raw": e["FETCH_SIZE"], "write_kb": e["WRITE_SIZE"], "source": "profiles/" + fn}
            if e.get("algorithmic_mb"):
                out["algorithmic_bytes_per_launch"] = int(e["algorithmic_mb"] * 1e6)
                out["traffic_over_algorithmic"] = round(by / (e["algorithmic_mb"] * 1e6), 3)
            meta = d.get("_meta") or {}
            out["measured_at_commit"] = meta.get("commit")
            out["evidence_set"] = meta.get("evidence_set")
            return out
    return None


def roofline_from(rows, bound="mfma"):
    cand = [r for r in rows if r.get("bound") == bound]
    if not cand:
        return None
    timed = [q for q in cand if q.get("timed")]          # the dominant op as ranked with every op bracketed, re-timed alone (measure())
    r = timed[0] if timed else max(cand, key=lambda q: q["ms_per_step"])
    key = f"{r['op']}[{r['shape']}]"
    if bound == "mfma":
        return {"bound": "mfma", "kernel": key, "achieved": r["achieved_tflops"], "peak": r["peak_tflops"], "unit": "TFLOP/s", "frac": r["frac"],
                "pipe": r["pipe"],
                "flop_per_launch": r["flop_per_launch"], "avg_launch_ms": round(r["avg_us"] / 1e3, 4), "launches_per_step": r["launches_per_step"],
                "share_of_step": r["share_of_step"], "traffic": pmc_traffic(key),
                "avg_launch_ms_in_step_event_bracket": round(r.get("avg_us_in_step_bracket", r["avg_us"]) / 1e3, 4), "timed": r.get("timed"),
                "how": "ranked by HIP-event brackets around every recorded op inside replayed steps (bcp_replay_run_timed) run right after the timed region; the dominant op is then timed by back-to-back launches between two HIP events on its stream (`timed`)"}
    return {"bound": "hbm", "kernel": key, "achieved": r["achieved_gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": r["frac"],
            "bytes_per_launch": r["bytes_per_launch"], "avg_launch_ms": round(r["avg_us"] / 1e3, 4), "traffic": pmc_traffic(key),
            "avg_launch_ms_in_step_event_bracket": round(r.get("avg_us_in_step_bracket", r["avg_us"]) / 1e3, 4), "timed": r.get("timed")}


def time_op_back_to_back(models, key, reps=20):
    """average GPU duration of ONE recorded op -- the launches a plan holds for (op name, first tensor's shape) -- each of `reps` issues
    bracketed by a pair of HIP events on the op's stream with the GPU idle around it.  (A pair of events around the same launches INSIDE the
    busy step reads 10-20 us more: marker packets wait for the command processor, which is serving three queues there; a rocprofv3
    --kernel-trace of the step -- profiles/*_kernel_stats.csv -- reports the kernels' own dispatch times, which this number stays within a
    few us above.)  None: no plan holds the op"""
    from bcp_amd.hip_ops import Ops
    ops = Ops.product()
    for m in models:
        st = m.__dict__.get("_plan_state")
        for pl in (st[1].values() if st else ()):
            for name, shapes, ints, namax, n0, n1 in pl.spans:
                if (name, shapes[0] if shapes else ()) != key or n1 <= n0:
                    continue
                ent = [(fn, args) for fn, args, nm in pl.entries[n0:n1] if nm is not None and nm != "bcp_stream_wait_stream"]
                if len(ent) != n1 - n0:
                    continue
                stream = ent[0][1][-1]
                torch.cuda.synchronize()
                for fn, args in ent:                      # warm-up
                    fn(*args)
                e0, e1 = ops.event(), ops.event()
                tot = 0.0
                for _ in range(reps):                     # one bracket per launch of the op, the GPU idle in front of it: the event pair costs
                    ops.b.call("bcp_event_record", e0, stream)      # 3-5 us here (inside the busy step 10-20), and nothing is cache-warm from a
                    for fn, args in ent:                            # back-to-back repetition of the same launch that the step would not have
                        rc = fn(*args)
                        assert rc == 0, ops.b.last_error()
                    ops.b.call("bcp_event_record", e1, stream)
                    torch.cuda.synchronize()
                    tot += ops.event_elapsed_ms(e0, e1)
                return tot / reps
    return None


def profile_steps(step, n, only=None):
    from bcp_amd.hip_ops import Ops
    ops = Ops.product()
    from bcp_amd import plan
    torch.cuda.synchronize()
    # Round 5: the profiled steps are REPLAYS, as the timed steps are -- launch by launch from C (bcp_replay_run_timed; forward graphs bypassed,
    # same kernels) with one HIP event in front of and one behind every recorded op on the op's own stream, the host as far ahead of the GPU
    # as in the timed region: the brackets hold the ops as they run inside the step, beside the other streams' work.  (Rounds 1-4 profiled
    # the EAGER path: its host is the bottleneck, every bracket included the Python between the record and the launch -- 8 % on the dominant
    # 55 us conv, 3 ms per call where the allocator stalled -- and no two streams ever overlapped.)  The few ops outside the recorded passes
    # (mix, pseudo-label, largest-CC, loss, optimiser, weight packs) are bracketed by their Python wrappers as before.
    plan.PROFILE, plan.PROFILE_ONLY = ops, only
    try:
        ops.profile_begin()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        return ops.profile_end(), ms
    finally:
        plan.PROFILE, plan.PROFILE_ONLY = None, None


# ------------------------------------------------------------------------------------------------ CPU baseline
def usable_cores():
    """threads this process may really use: scheduler affinity and the cgroup CPU quota, not os.cpu_count()
    (a container that reports 256 CPUs but owns 16 would otherwise be oversubscribed 16x)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = min(n, max(1, int(float(q[0]) / float(q[1]))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return max(1, min(n, 64))   # torch-CPU convs stop scaling long before 64 threads


def cpu_baseline(workload, batch, labeled_bs, budget_s=45.0):
    """the oracle's self-training step (forward x2 teacher, CC, mix, student forward / backward, optimiser, EMA) on the host cores:
    2 warm-ups, then the median of >= 5 timed steps (fewer only if the time budget runs out -- said in `sample`)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bcp_oracle as O  # checker / baseline only
    cores = usable_cores()
    torch.set_num_threads(cores)
    rng = np.random.default_rng(0)
    if workload == "acdc":
        shapes = O.unet_param_shapes()
        vol, lab = O.synth_acdc_batch(batch, seed=1337)
        box = O.box_acdc(lambda lo, hi: int(rng.integers(lo, hi)))
        unit, what = "slices/s", f"batch {batch}, 256x256"
    elif workload == "pancreas":
        shapes = O.vnet_param_shapes(variant="pancreas")
        vol, lab = O.synth_la_batch(batch, shape=(96, 96, 96), seed=1337)
        box = O.box_pancreas(lambda lo, hi: int(rng.integers(lo, hi)))
        unit, what = "volumes/s", f"batch {batch}, 96^3"
    else:
        shapes = O.vnet_param_shapes()
        vol, lab = O.synth_la_batch(batch, seed=1337)
        box = O.box_la(lambda lo, hi: int(rng.integers(lo, hi)))
        unit, what = "volumes/s", f"batch {batch}, 112x112x80"
    Ps = O.init_params(shapes, seed=1337)
    Pt = {k: v.clone() for k, v in Ps.items()}
    tkeys = O.trainable_keys(shapes)
    bufs, times = {}, []
    sub = labeled_bs // 2
    t_start = time.time()
    warm, want = 2, 5
    for it in range(warm + 9):
        t0 = time.time()
        if workload == "acdc":
            r = O.acdc_self_train_step(Ps, Pt, vol, lab, box, {}, sub, (batch - labeled_bs) // 2)
            O.sgd_step(Ps, r["grads"], bufs, tkeys, lr=0.01)
            O.ema_state_dict(Ps, Pt, 0.99)
        elif workload == "pancreas":
            r = O.la_self_train_step(Ps, Pt, vol, lab, box, {}, sub, variant="pancreas", connectivity=2)
            O.adam_step(Ps, r["grads"], bufs, tkeys)
            O.ema_params(Ps, Pt, tkeys, 0.99)
        else:
            drops = {k: {"x5": torch.from_numpy((rng.random((sub, 256)) < 0.5).astype(np.float32)),
                         "x9": torch.from_numpy((rng.random((sub, 16)) < 0.5).astype(np.float32))} for k in ("t_a", "t_b", "s_l", "s_u")}
            r = O.la_self_train_step(Ps, Pt, vol, lab, box, drops, sub)
            O.sgd_step(Ps, r["grads"], bufs, tkeys, lr=0.01)
            O.ema_params(Ps, Pt, tkeys, 0.99)
        times.append(time.time() - t0)
        print(f"[bench] cpu_baseline step {it}: {times[-1]:.2f} s ({cores} threads)", file=sys.stderr, flush=True)
        timed = len(times) - warm
        if timed >= want and (timed >= 7 or time.time() - t_start > budget_s * 0.6):
            break
        if time.time() - t_start > budget_s and timed >= 1:
            break
    steady = times[warm:] if len(times) > warm else times[-1:]
    sec = float(np.median(steady))
    return {"value": round(batch / sec, 3), "unit": unit, "cores": cores, "kind": "port",
            "sample": f"median of {len(steady)} timed self-train steps ({what}) after {min(warm, len(times) - len(steady))} warm-ups, torch-CPU oracle "
                      f"(oracle/bcp_oracle.py), {cores} threads", "sec_per_step": round(sec, 3),
            "sec_per_step_all": [round(t, 3) for t in steady]}


# ------------------------------------------------------------------------------------------------ workloads
def make_workload(args, dp, dev):
    from bcp_amd import synth, train_step
    seed = 1337 + dp.rank
    np.random.seed(seed)            # context_mask draws from the global numpy RNG as the reference does
    if args.workload == "la":
        model, ema_model = build_models(dev, 1337)
        dp.broadcast_params(model); dp.broadcast_params(ema_model)
        opt = train_step.FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
        vol, lab = synth.la_batch(args.batch_size, seed=seed)
        vol, lab = vol.to(dev), lab.to(dev)

        def step():
            return train_step.la_self_train_step(model, ema_model, opt, vol, lab, args.labeled_bs, dp=dp if dp.enabled else None)
        step.models = (model, ema_model)
        model.volatile_io = ema_model.volatile_io = _VOLATILE_IO
        return step, {"metric": "training volumes/sec (LA 112x112x80 V-Net, BCP self-training step)", "unit": "volumes/s",
                      "what": f"LA 3D V-Net BCP self-train step, per-GPU batch {args.batch_size} ({args.labeled_bs} labeled), 112x112x80 patches, "
                              "SGD m0.9 wd1e-4, EMA 0.99 (BASELINE.json configs[1])", "gflop_per_item": STEP_GFLOP_PER_VOLUME}
    torch.manual_seed(1337)
    if args.workload == "acdc":
        from bcp_amd.networks.net_factory import BCP_net
        model, ema_model = BCP_net(in_chns=1, class_num=4), BCP_net(in_chns=1, class_num=4, ema=True)
        ema_model.load_state_dict(model.state_dict())
        model.train(); ema_model.train()
        dp.broadcast_params(model); dp.broadcast_params(ema_model)
        opt = train_step.FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
        vol, lab = synth.acdc_batch(args.batch_size, seed=seed)
        vol, lab = vol.to(dev), lab.to(dev)

        def step():
            return train_step.acdc_self_train_step(model, ema_model, opt, vol, lab, args.labeled_bs, dp=dp if dp.enabled else None)
        step.models = (model, ema_model)
        model.volatile_io = ema_model.volatile_io = _VOLATILE_IO
        return step, {"metric": "training slices/sec (ACDC 256x256 U-Net, BCP self-training step)", "unit": "slices/s",
                      "what": f"ACDC 2D U-Net BCP self-train step, per-GPU batch {args.batch_size} ({args.labeled_bs} labeled), 256x256 slices, SGD, "
                              "state-dict EMA (BASELINE.json configs[3])",
                      "gflop_per_item": 5.90 * 2}   # 5.90 GFLOP fwd / slice; step = 1/2 teacher fwd + 1/2 (fwd + 2x bwd) per input slice
    from bcp_amd.pancreas import train_pancreas as TP
    from bcp_amd.pancreas.Vnet import create_Vnet
    model, ema_model = create_Vnet(), create_Vnet(ema=True)
    ema_model.load_state_dict(model.state_dict())
    dp.broadcast_params(model); dp.broadcast_params(ema_model)
    opt = train_step.FlatAdam(model, lr=1e-3)
    assert args.batch_size % 4 == 0, "pancreas: four equal streams (lab_a, lab_b, unlab_a, unlab_b)"
    streams = TP._streams(dev, 4, args.batch_size // 4, seed=seed)

    def step():
        return {"loss": TP.ema_cutmix(model, ema_model, opt, streams, 1, dp=dp if dp.enabled else None)}
    step.models = (model, ema_model)
    model.volatile_io = ema_model.volatile_io = _VOLATILE_IO
    return step, {"metric": "training volumes/sec (Pancreas 96^3 IN-V-Net, BCP self-training step)", "unit": "volumes/s",
                  "what": f"Pancreas IN-V-Net BCP self-train step, per-GPU 4 streams x {args.batch_size // 4}, 96^3 patches, Adam 1e-3 "
                          "(BASELINE.json configs[4])", "gflop_per_item": 70.72 * 2}


def measure(args, dp, dev, cpu_budget_s=45.0, trim=False):
    """one workload under the timing contract: W warm-ups, K timed steps between barrier + synchronize, max over ranks; then the
    per-op table from profiled steps and (rank 0, N = 1) the CPU baseline.  -> the JSON object of the line (rank 0), None elsewhere"""
    from bcp_amd import plan
    step, info = make_workload(args, dp, dev)
    ranks_seen = dp.ranks_seen()

    # one-time host-side setup, whatever W is: the first pass of a network records its launch plan, the second is captured into a HIP graph
    # (bcp_amd/plan.py) -- the equivalent of a compile step, never part of the W warm-ups or the K timed steps
    SETUP_STEPS = 2
    for i in range(SETUP_STEPS):
        step()
        if i == 0:      # (measurement only: what-if switches that leave buffers unwritten go on AFTER one real pass has filled them)
            for kv in getattr(args, "opt_late", None) or []:
                k, _, v = kv.partition("=")
                from bcp_amd.hip_ops import Ops as _Ops
                _Ops.product().set_option(k, v)
    for _ in range(args.warmup):
        step()
    dp.barrier()
    torch.cuda.synchronize()
    dp.reset_exposed()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = step()
    t_enq = time.perf_counter() - t0          # host time to enqueue the K steps (host-bound if ~= the total)
    torch.cuda.synchronize()
    dp.barrier()
    dt = time.perf_counter() - t0
    dt = dp.max_over_ranks(dt)
    loss = float(r["loss"])
    assert np.isfinite(loss), "non-finite loss in the timed region"
    exposed = dp.exposed_ms_per_step(args.steps)
    # host cost of ONE step with an empty queue (t_enq above includes back-pressure: once the host runs ahead, hipLaunchKernel
    # blocks on the full queue and "enqueue time" just tracks the GPU)
    host_one = []
    for _ in range(5):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        step()
        host_one.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    host_ms = sorted(host_one)[len(host_one) // 2] * 1e3

    rows_top, rows_all, prof_ms = [], [], None
    if args.profile_steps > 0:
        recs, prof_ms = profile_steps(step, args.profile_steps)
        rows_top, rows_all = op_table(recs, args.profile_steps, dt / args.steps * 1e3, top=8 if trim else 14)
        # second pass: the dominant op of each kind bracketed ALONE (every event is a marker packet between two kernels; with all ~300 ops
        # of a step bracketed the step itself is 20 % slower and every bracket a few us longer) -- these are the numbers `roofline` carries
        dom = []
        for b_ in ("mfma", "hbm"):
            cand = [r for r in rows_all if r.get("bound") == b_]
            if cand:
                r = max(cand, key=lambda q: q["ms_per_step"])
                dom.append((r["op"], tuple(int(v) for v in r["shape"].split("x")) if r["shape"] else ()))
        if dom:
            recs2, prof2_ms = profile_steps(step, args.profile_steps, only=set(dom))
            _, rows2 = op_table(recs2, args.profile_steps, dt / args.steps * 1e3, top=4)
            alone = {(r["op"], r["shape"]): r for r in rows2}
            for r in rows_all:
                a_ = alone.get((r["op"], r["shape"]))
                if a_ is not None and (r["op"], tuple(int(v) for v in r["shape"].split("x")) if r["shape"] else ()) in dom:
                    r["avg_us_all_ops_bracketed"] = r["avg_us"]
                    for k_ in ("avg_us", "ms_per_step", "share_of_step", "achieved_tflops", "achieved_gbs", "frac"):
                        if k_ in a_:
                            r[k_] = a_[k_]
                    r["timed"] = f"bracketed alone in {args.profile_steps} replayed steps of {prof2_ms:.2f} ms"
                    # third number, the one the roofline fraction is taken from: the op's recorded launches issued one at a time on an idle
                    # GPU, a pair of events around each -- an event in front of a kernel inside the busy step costs the bracket 10-20 us of
                    # marker latency (LA: 70-73 us in the step's bracket against 54 us in a rocprofv3 kernel trace of the same step), which
                    # is not kernel time
                    b2b = time_op_back_to_back(getattr(step, "models", ()), (r["op"], tuple(int(v) for v in r["shape"].split("x")) if r["shape"] else ()))
                    if b2b is not None:
                        r["avg_us_in_step_bracket"] = r["avg_us"]
                        r["avg_us"] = round(b2b * 1e3, 1)
                        if r["bound"] == "mfma":
                            r["achieved_tflops"] = round(r["flop_per_launch"] / (b2b * 1e-3) / 1e12, 2)
                            r["frac"] = round(r["achieved_tflops"] / r["peak_tflops"], 4)
                        else:
                            r["achieved_gbs"] = round(r["bytes_per_launch"] / (b2b * 1e-3) / 1e9, 1)
                            r["frac"] = round(r["achieved_gbs"] / PEAK_HBM_GBS, 4)
                        r["timed"] = (f"mean of 20 issues of the op's recorded launches, each between two HIP events on its stream with the GPU idle around it; "
                                      f"inside the busy step a pair of events around the same launches reads {r['avg_us_in_step_bracket']} us (marker latency of a "
                                      "command processor serving three queues), a rocprofv3 kernel trace of the step gives the kernels' own times (profiles/)")
    dp.barrier()
    del step
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    if dp.rank != 0:
        return None
    ms = dt / args.steps * 1e3
    global_batch = args.batch_size * dp.world
    value = global_batch * args.steps / dt
    step_tflops = value * info["gflop_per_item"] / 1e3 / dp.world
    out = {
        "metric": info["metric"], "value": round(value, 3), "unit": info["unit"], "n_gpus": dp.world, "steps": args.steps, "warmup": args.warmup, "setup_steps": SETUP_STEPS,
        "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3), "host_ms_per_step_empty_queue": round(host_ms, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": info["what"], "global_batch": global_batch, "parallelism": f"dp{dp.world}", "last_loss": round(loss, 6),
                   "host_path": (f"recorded network passes (bcp_amd/plan.py): forward = {'one HIP graph launch' if plan.GRAPHS >= 1 else 'per-launch replay from C'}, "
                                 f"backward = {'HIP graph' if plan.GRAPHS >= 2 else 'per-launch replay from C (bcp_replay_run)'}; teacher forward on its own stream "
                                 "under the student's, weight gradients on a third stream.  This is the configuration the parity suite runs bit for bit "
                                 "against the eager path under a load generator (tests/test_gpu_vnet.py::test_product_configuration_under_load_equals_eager_path)"),
                   "arithmetic": "fp32 tensors, fp32 accumulation; the 16- to 256-channel 3x3x3 / 3x3 convolutions (forward, dgrad, weight gradient) take their fp32 "
                                 "operands as 16-bit pieces on the matrix cores (csrc/conv3b.hip, conv3bw.hip): two fp16 pieces pre-scaled by powers of two from "
                                 "each tensor's own |max| (three MFMAs per product block; round 4) wherever the producing norm pass left that |max|, three bf16 "
                                 "pieces (six MFMAs) elsewhere -- fp32-equivalent results: error vs fp64 within 3x of the fp32-MFMA kernels' "
                                 "(tests/kernel_checks.py check_conv3_f16 / check_conv3_b6), same network parity tolerances (tests/test_gpu_vnet.py); the kernels "
                                 "table names the pipe and the peak each op is priced against"},
        "ranks_seen": ranks_seen,
        "step_flops": {"gflop_per_item": info["gflop_per_item"], "achieved_tflops_per_gpu": round(step_tflops, 2),
                       "frac_of_bf16x3_peak": round(step_tflops / PEAK_BF16X3_F32EQ_TFLOPS, 4),
                       "frac_of_f16x2_peak": round(step_tflops / PEAK_F16X2_F32EQ_TFLOPS, 4)},
    }
    if dp.world > 1:
        # how much of the gradient exchange the backward pass did NOT hide: wall time the optimiser's stream waited for the last
        # buckets (HIP events around allreduce_grads' final wait), and the bucket plan the run used
        out["exposed_allreduce_ms_per_step"] = exposed
        out["allreduce_exposed_ms"] = exposed          # (the key of rounds 3-4)
        if getattr(args, "share_gpu", False):
            out["share_gpu"] = "all ranks on cuda:0, exchange over gloo: exercises the bucket / overlap logic, NOT a scaling measurement"
        out["allreduce_buckets"] = dp.bucket_report()
    if rows_all:
        out["roofline"] = roofline_from(rows_all, "mfma")
        out["roofline_hbm"] = roofline_from(rows_all, "hbm")
        out["kernels"] = rows_top
        out["kernels_note"] = (f"{args.profile_steps} profiled steps of {prof_ms:.2f} ms after the timed region; ms_per_step sums over streams "
                               "(teacher / student / weight-gradient streams overlap), so shares add up to more than 1")
    else:
        out["roofline"] = {"bound": "mfma", "achieved": None, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None}
    r0 = out["roofline"] or {}
    print(f"[bench] {args.workload} gpu: {value:.2f} {info['unit']}, {ms:.2f} ms/step (host enqueue {t_enq / args.steps * 1e3:.2f} ms/step), "
          f"dominant MFMA op {r0.get('kernel')} {r0.get('achieved')} TFLOP/s", file=sys.stderr, flush=True)
    if dp.world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload, args.batch_size, args.labeled_bs, budget_s=cpu_budget_s)
    return out


# ------------------------------------------------------------------------------------------------ the record
MAX_LINE_BYTES = 4096               # the LAST stdout line: a record, not a report


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def compact_roofline(r):
    """contract keys of a `roofline` object: bound, kernel, achieved, peak, unit, frac, traffic (HBM bytes per launch from the committed --pmc
    passes, or null) + the duration the fraction is taken from and the same op's in-step event bracket as a second figure"""
    if not r:
        return None
    t = r.get("traffic")
    o = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac"))
    o["traffic"] = t.get("bytes_per_launch") if isinstance(t, dict) else t
    if isinstance(t, dict) and t.get("algorithmic_bytes_per_launch"):
        o["algorithmic_bytes"] = t["algorithmic_bytes_per_launch"]
    if r.get("avg_launch_ms") is not None:
        o["avg_launch_us"] = round(r["avg_launch_ms"] * 1e3, 1)
    b = r.get("avg_launch_ms_in_step_event_bracket")
    if b and r.get("avg_launch_ms") and o.get("frac") is not None:
        o["avg_launch_us_in_step_bracket"] = round(b * 1e3, 1)
        o["frac_in_step_bracket"] = round(o["frac"] * r["avg_launch_ms"] / b, 4)
    return o


def compact_record(out, detail=None):
    """the one stdout line: the contract keys and nothing else (everything `measure()` collects beyond them lives in bench_detail.json)"""
    rec = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    rec["config"] = _pick(out.get("config") or {}, ("workload", "global_batch", "parallelism"))
    rec["roofline"] = compact_roofline(out.get("roofline"))
    if out.get("cpu_baseline"):
        rec["cpu_baseline"] = _pick(out["cpu_baseline"], ("value", "unit", "cores", "kind", "sample"))
    for k in ("ranks_seen", "exposed_allreduce_ms_per_step"):
        if out.get(k) is not None:
            rec[k] = out[k]
    ex = {}
    for wl, e in (out.get("extra_workloads") or {}).items():
        if "error" in e:
            ex[wl] = {"error": str(e["error"])[:120]}
            continue
        ex[wl] = {"value": e.get("value"), "unit": e.get("unit"), "ms_per_step": e.get("ms_per_step"),
                  "roofline_frac": (e.get("roofline") or {}).get("frac"), "cpu_baseline_value": (e.get("cpu_baseline") or {}).get("value")}
    if ex:
        rec["extra_workloads"] = ex
    if detail:
        rec["detail"] = detail
    return rec


def write_detail(out):
    """the full report: ROOT/bench_detail.json (git-ignored) and, when the directory exists, gpurun_out/ (travels back from the GPU box)"""
    paths = []
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if not os.path.isdir(d):
            continue
        try:
            fn = os.path.join(d, "bench_detail.json")
            with open(fn, "w") as f:
                json.dump(out, f, indent=1)
            paths.append(fn)
        except OSError:
            pass
    print("[bench] detail: " + (", ".join(paths) if paths else "not written (read-only tree)"), file=sys.stderr, flush=True)
    return paths


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["la", "acdc", "pancreas"], default="la",
                    help="la = BASELINE.json's metric (configs[1]); acdc / pancreas = the north_star's secondary lines (configs[3], [4])")
    ap.add_argument("--batch_size", type=int, default=None, help="per-GPU batch (la: 4 = configs[1]; acdc: 24 = configs[3]; pancreas: 4)")
    ap.add_argument("--labeled_bs", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=3, help="steps run with per-op HIP events after the timed region (0: no kernels table)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="library tuning switch (bcp_set_option) for A/B measurements; the product defaults need none")
    ap.add_argument("--opt-late", action="append", default=[], metavar="NAME=VALUE",
                    help="MEASUREMENT ONLY: a library switch set after the first set-up step (the `whatif` bits: launches left out, results wrong)")
    ap.add_argument("--no-roofline", action="store_true", help="same as --profile-steps 0 (A/B runs)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="N > 1 ranks on ONE GPU (every rank uses cuda:0, gradient exchange over gloo: RCCL refuses two ranks per device): times "
                         "the bucketed exchange / overlap logic when no multi-GPU node is at hand -- NOT a scaling measurement")
    ap.add_argument("--no-extra", action="store_true", help="la, 1 GPU: skip the secondary lines (extra_workloads: acdc, pancreas; cpu_only: configs[0])")
    args = ap.parse_args()
    if args.batch_size is None:
        args.batch_size = {"la": 4, "acdc": 24, "pancreas": 4}[args.workload]
    if args.labeled_bs is None:
        args.labeled_bs = args.batch_size // 2
    if args.no_roofline:
        args.profile_steps = 0

    from bcp_amd.dp import DataParallel
    from bcp_amd.hip_ops import Ops

    if args.share_gpu:
        os.environ["LOCAL_RANK"] = "0"
        os.environ.setdefault("BCP_DP_BACKEND", "gloo")
    dp = DataParallel()
    assert dp.world == args.gpus or (args.gpus == 1 and dp.world == 1), f"--gpus {args.gpus} but WORLD_SIZE={dp.world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    dev = torch.device("cuda", dp.local_rank)
    torch.cuda.set_device(dev)
    from bcp_amd import plan
    plan.use_real_stream(dev)       # as the training scripts do (a real stream is also what --opt graphs=1 needs: the null stream cannot be captured)
    ops = Ops.product()
    for kv in args.opt:
        k, _, v = kv.partition("=")
        if k == "graphs":             # host-side switch (bcp_amd/plan.py)
            plan.GRAPHS = int(v)
            continue
        if k == "teacher_prio":       # host-side switch (train_step.py): HIP priority of the teacher's side stream
            from bcp_amd import train_step as _ts
            _ts.TEACHER_STREAM_PRIORITY = int(v)
            continue
        if k == "pack_partial":       # host-side switch (networks/_hipnet.py): only the observed sections of the weight packs in front of replays
            from bcp_amd.networks._hipnet import HipNet as _hn5
            _hn5.PACK_PARTIAL = bool(int(v))
            continue
        if k == "wgrad_prio":         # host-side switch (networks/_hipnet.py): HIP priority of the weight-gradient side stream
            from bcp_amd.networks._hipnet import HipNet as _hn
            _hn.WGRAD_STREAM_PRIORITY = int(v)
            continue
        if k == "fuse_c1":            # host-side switch (networks/VNet.py, unet.py): first layer + norm with recompute
            from bcp_amd.networks.VNet import VNet as _vn
            from bcp_amd.networks.unet import UNet_2d as _un
            _vn.fuse_c1 = _un.fuse_c1 = bool(int(v))
            continue
        if k == "defer_dgrad_pack":   # host-side switch (networks/_hipnet.py): dgrad weight packs on the side stream, off the forward's critical path
            from bcp_amd.networks._hipnet import HipNet as _hn2
            _hn2.DEFER_DGRAD_PACK = bool(int(v))
            continue
        if k == "step_total":         # host-side switch (train_step.py): total loss summed by the second mix_loss launch, cached unit gradient
            from bcp_amd import train_step as _ts2
            _ts2.STEP_TOTAL = bool(int(v))
            continue
        if k == "inline_dropout":     # host-side switch (networks/unet.py): Dropout keep bits evaluated in the norm kernels (no mask tensors)
            from bcp_amd.networks.unet import UNet_2d as _un3
            _un3.inline_dropout = bool(int(v))
            continue
        if k == "skip_in_concat":     # host-side switch (networks/unet.py): encoder outputs written into the decoder's concat buffers
            from bcp_amd.networks.unet import UNet_2d as _un2
            _un2.skip_in_concat = bool(int(v))
            continue
        if k in ("wgrad_cumask", "teacher_cumask"):      # MEASUREMENT ONLY: the side stream as a CU-masked stream (hipExtStreamCreateWithCUMask); v = 8 hex words "w0:w1:..:w7", bit = CU
            import ctypes as _C
            _hip = _C.CDLL("libamdhip64.so")
            words = [int(w, 16) for w in v.split(":")]
            arr = (_C.c_uint32 * len(words))(*words)
            st = _C.c_void_p()
            rc = _hip.hipExtStreamCreateWithCUMask(_C.byref(st), len(words), arr)
            assert rc == 0, f"hipExtStreamCreateWithCUMask failed: {rc}"
            ext = torch.cuda.ExternalStream(st.value, device=dev)
            if k == "wgrad_cumask":
                from bcp_amd.networks._hipnet import HipNet as _hn7
                _hn7._side_streams[dev] = ext
            else:
                from bcp_amd import train_step as _ts7
                _ts7._SIDE[dev] = ext
            print(f"[bench] {k}: {sum(bin(w).count('1') for w in words)} CUs", file=sys.stderr)
            continue
        if k == "plabel_cc_fused":    # host-side switch (train_step.py): pseudo-label + largest-CC as one chain
            import bcp_amd.train_step as _ts10
            _ts10.PLABEL_CC_FUSED = bool(int(v))
            continue
        if k == "up_recompute_grad":  # host-side switch (networks/VNet.py): the recomputing transposed conv + norm also in forwards that save for a backward pass
            from bcp_amd.networks.VNet import VNet as _vn9
            _vn9.UP_RECOMPUTE_GRAD = bool(int(v))
            continue
        if k == "wgrad_defer":        # host-side switch (networks/VNet.py, unet.py): small layers' weight gradients fork in batches of this many
            from bcp_amd.networks.VNet import VNet as _vn6
            from bcp_amd.networks.unet import UNet_2d as _un6
            _vn6.WGRAD_DEFER = _un6.WGRAD_DEFER = int(v)
            continue
        if k == "fuse_head":          # host-side switch (networks/VNet.py), not a library option
            from bcp_amd.networks.VNet import VNet
            VNet.fuse_head = bool(int(v))
            continue
        ops.set_option(k, v)
    out = measure(args, dp, dev)
    if dp.rank == 0:
        if dp.world == 1 and args.workload == "la" and not args.no_extra:
            # the north_star's other single-GPU lines in the same driver-run record (same timing contract, own roofline + cpu_baseline),
            # bounded: fewer timed steps and a shorter CPU sample each
            import copy
            extra = {}
            for wl in ("acdc", "pancreas", "la_b8"):
                a2 = copy.copy(args)
                if wl == "la_b8":
                    # the reference's DEFAULT LA batch (LA_BCP_train.py:39-40: batch_size 8, labeled_bs 4; SURVEY 8d "also report"): GPU line only
                    a2.workload, a2.batch_size, a2.labeled_bs, a2.no_cpu_baseline = "la", 8, 4, True
                    a2.steps, a2.warmup = min(args.steps, 10), min(args.warmup, 3)
                else:
                    a2.workload, a2.batch_size, a2.labeled_bs = wl, {"acdc": 24, "pancreas": 4}[wl], {"acdc": 12, "pancreas": 2}[wl]
                a2.profile_steps = min(args.profile_steps, 2)
                try:
                    extra[wl] = measure(a2, dp, dev, cpu_budget_s=20.0, trim=True)
                except Exception as e:      # the headline line must survive a failure of a secondary one
                    extra[wl] = {"error": f"{type(e).__name__}: {e}"}
            out["extra_workloads"] = extra
            if not args.no_cpu_baseline:
                try:
                    # BASELINE.json configs[0]: ACDC batch 8 (4 labeled) -- the reference's CPU-runnable case: the oracle on the host cores
                    out["cpu_only"] = {"configs[0] ACDC 2D U-Net batch 8 (4 labeled) 256x256, CPU": cpu_baseline("acdc", 8, 4, budget_s=15.0)}
                except Exception as e:
                    out["cpu_only"] = {"error": f"{type(e).__name__}: {e}"}
        # the full report (per-op tables, both rooflines with their provenance, prose) goes to a FILE and to stderr; stdout's last line is
        # the compact record of the contract keys (round 5's 24.6 KB line did not survive the driver's stdout tail: BENCH_r05.json parsed = null)
        detail_paths = write_detail(out)
        line = json.dumps(compact_record(out, detail=os.path.relpath(detail_paths[0], ROOT) if detail_paths else None), separators=(",", ":"))
        assert len(line) < MAX_LINE_BYTES, f"bench record is {len(line)} bytes (limit {MAX_LINE_BYTES})"
        sys.stderr.flush()
        print(line, flush=True)
    dp.shutdown()


if __name__ == "__main__":
    main()
