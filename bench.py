#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric: training volumes/sec of the BCP self-training step on the LA 3-D V-Net
(112x112x80 patches), per-GPU batch 4 (2 labeled + 2 unlabeled; BASELINE.json configs[1]), synthetic data,
fp32, random-init weights, everything inside the timed region that the reference's loop does per iteration
(LA_BCP_train.py:235-270): teacher forward x2, pseudo-label + largest-CC x2, box draw, copy-paste mix x2,
student forward x2, masked Dice+CE x2, backward, [gradient all-reduce when N>1], SGD, EMA.

  python bench.py [--gpus N --steps K --warmup W]
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `roofline` = the dominant kernel (k_conv3_res, the fp32-MFMA implicit-GEMM
3x3x3 conv at the 16->16 @112x112x80 layer: 13.87 GFLOP algorithmic per launch) timed with HIP events on
the launch stream in this process; `cpu_baseline` = the oracle (CPU restatement of the reference,
oracle/bcp_oracle.py) timed on this box's host cores on a bounded sample of the same workload.
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
STEP_GFLOP_PER_VOLUME = 160.0       # 1/2 teacher fwd + 1/2 student fwd+bwd per input volume


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


def dominant_kernel_roofline(dev):
    """k_conv3_res<3,4,4,16,1> at the shape the step launches it with -- the 16->16 layer at 112x112x80 over the grouped
    batch of 2 (the two teacher / student sub-batches go through every layer as ONE launch) -- HIP events on the launch stream"""
    from bcp_amd.hip_ops import Ops
    ops = Ops.product()
    N, sp, C = 2, (112, 112, 80), 16
    x = torch.randn(N, *sp, C, device=dev)
    w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
    b = torch.zeros(C, device=dev)
    wf, _ = ops.conv3_pack(w, 3)
    y = torch.empty(N, *sp, C, device=dev)
    for _ in range(3):
        ops.conv3_fwd(x, wf, b, C, 3, out=y)
    e0, e1 = ops.event(), ops.event()
    iters = 20
    ops.event_record(e0, x)
    for _ in range(iters):
        ops.conv3_fwd(x, wf, b, C, 3, out=y)
    ops.event_record(e1, x)
    ms = ops.event_elapsed_ms(e0, e1) / iters
    flops = 2.0 * N * sp[0] * sp[1] * sp[2] * 27 * C * C
    ach = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "k_conv3_res<3,4,4,16,1> (3x3x3 conv 16->16 @112x112x80 x batch 2 as launched in the step, fwd/dgrad)",
            "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
            "flop_per_launch": flops, "avg_launch_ms": round(ms, 4), "traffic": pmc_traffic()}


def pmc_traffic():
    """HBM-side bytes per launch of the same kernel from the committed rocprofv3 --pmc passes (tools/collect_pmc.sh: FETCH_SIZE
    and WRITE_SIZE in separate runs; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note on 16-B/lane streams).  bench.py
    cannot run the profiler itself; null when the file is absent."""
    p = os.path.join(ROOT, "profiles", "r01_pmc_conv3_c16_v3.json")
    try:
        d = json.load(open(p))["k_conv3_res<3,4,4,16,1>"]
        return {"bytes_per_launch": int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024), "fetch_kb_raw": d["FETCH_SIZE"],
                "write_kb": d["WRITE_SIZE"], "algorithmic_bytes": 2 * 2 * 1003520 * 16 * 4 + 27 * 16 * 16 * 4,
                "mfma_busy_frac": round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (d["GRBM_GUI_ACTIVE"] / 8), 4),
                "source": "profiles/r01_pmc_conv3_c16_v3.json"}
    except Exception:
        return None


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


def cpu_baseline(batch, labeled_bs):
    """the oracle's self-training step on the host cores: 1 warm-up + timed steps until ~20 s"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bcp_oracle as O  # checker / baseline only
    cores = usable_cores()
    torch.set_num_threads(cores)
    shapes = O.vnet_param_shapes()
    Ps = O.init_params(shapes, seed=1337)
    Pt = {k: v.clone() for k, v in Ps.items()}
    tkeys = O.trainable_keys(shapes)
    vol, lab = O.synth_la_batch(batch, seed=1337)
    rng = np.random.default_rng(0)
    box = O.box_la(lambda lo, hi: int(rng.integers(lo, hi)))
    bufs = {}
    times = []
    t_start = time.time()
    for it in range(3):
        drops = {k: {"x5": torch.from_numpy((rng.random((labeled_bs // 2, 256)) < 0.5).astype(np.float32)),
                     "x9": torch.from_numpy((rng.random((labeled_bs // 2, 16)) < 0.5).astype(np.float32))} for k in ("t_a", "t_b", "s_l", "s_u")}
        t0 = time.time()
        r = O.la_self_train_step(Ps, Pt, vol, lab, box, drops, labeled_bs // 2)
        O.sgd_step(Ps, r["grads"], bufs, tkeys, lr=0.01)
        O.ema_params(Ps, Pt, tkeys, 0.99)
        times.append(time.time() - t0)
        print(f"[bench] cpu_baseline step {it}: {times[-1]:.2f} s ({cores} threads)", file=sys.stderr, flush=True)
        if time.time() - t_start > 25:
            break
    steady = times[1:] if len(times) > 1 else times
    sec = float(np.median(steady))
    return {"value": round(batch / sec, 3), "unit": "volumes/s", "cores": cores, "kind": "port",
            "sample": f"{len(steady)} timed self-train step(s) (batch {batch}, 112x112x80) after 1 warm-up, torch-CPU oracle, {cores} threads",
            "sec_per_step": round(sec, 3)}


def secondary(args):
    """ACDC 2-D U-Net (configs[3]: batch 24 256x256, SGD, state-dict EMA) and pancreas IN-V-Net (configs[4]: 96^3, Adam) self-training
    steps: same timing contract and JSON shape as the LA line, without the dominant-kernel / CPU legs (those belong to the
    headline metric)."""
    from bcp_amd import synth, train_step
    from bcp_amd.dp import DataParallel
    from bcp_amd.hip_ops import Ops

    dp = DataParallel()
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    dev = torch.device("cuda", dp.local_rank)
    torch.cuda.set_device(dev)
    Ops.product()
    seed = 1337 + dp.rank
    np.random.seed(seed)
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
        unit, what = "slices/s", f"ACDC 2D U-Net BCP self-train step, per-GPU batch {args.batch_size} ({args.labeled_bs} labeled), 256x256 slices, SGD, state-dict EMA (BASELINE.json configs[3])"
        gflop_per_item = 5.90 * 2          # 5.90 GFLOP fwd / slice; step = 1/2 teacher fwd + 1/2 (fwd + 2x bwd) per input slice

        def step():
            return train_step.acdc_self_train_step(model, ema_model, opt, vol, lab, args.labeled_bs, dp=dp if dp.enabled else None)
    else:
        from bcp_amd.pancreas import train_pancreas as TP
        from bcp_amd.pancreas.Vnet import create_Vnet
        model, ema_model = create_Vnet(), create_Vnet(ema=True)
        ema_model.load_state_dict(model.state_dict())
        dp.broadcast_params(model); dp.broadcast_params(ema_model)
        opt = train_step.FlatAdam(model, lr=1e-3)
        assert args.batch_size % 4 == 0, "pancreas: four equal streams (lab_a, lab_b, unlab_a, unlab_b)"
        streams = TP._streams(dev, 4, args.batch_size // 4, seed=seed)
        unit, what = "volumes/s", f"Pancreas IN-V-Net BCP self-train step, per-GPU 4 streams x {args.batch_size // 4}, 96^3 patches, Adam 1e-3 (BASELINE.json configs[4])"
        gflop_per_item = 70.72 * 2

        def step():
            return {"loss": TP.ema_cutmix(model, ema_model, opt, streams, 1, dp=dp if dp.enabled else None)}

    for _ in range(args.warmup):
        step()
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dp.barrier()
    dt = dp.max_over_ranks(time.perf_counter() - t0)
    loss = float(r["loss"])
    assert np.isfinite(loss), "non-finite loss in the timed region"
    if dp.rank == 0:
        gb = args.batch_size * dp.world
        value = gb * args.steps / dt
        tf = value * gflop_per_item / 1e3 / dp.world
        print(json.dumps({
            "metric": f"training {unit.split('/')[0]}/sec ({args.workload} BCP self-training step)", "value": round(value, 3), "unit": unit,
            "n_gpus": dp.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": what, "global_batch": gb, "parallelism": f"dp{dp.world}", "last_loss": round(loss, 6)},
            "step_flops": {"gflop_per_item": gflop_per_item, "achieved_tflops_per_gpu": round(tf, 2),
                           "frac_of_f32_mfma_peak": round(tf / PEAK_F32_MFMA_TFLOPS, 4)}}), flush=True)
    dp.shutdown()


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
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="library tuning switch (bcp_set_option) for A/B measurements; the product defaults need none")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel roofline launches (A/B runs)")
    args = ap.parse_args()
    if args.opt:
        from bcp_amd.hip_ops import Ops as _Ops
        for kv in args.opt:
            k, _, v = kv.partition("=")
            _Ops.product().set_option(k, v)
    if args.batch_size is None:
        args.batch_size = {"la": 4, "acdc": 24, "pancreas": 4}[args.workload]
    if args.labeled_bs is None:
        args.labeled_bs = args.batch_size // 2
    if args.workload != "la":
        return secondary(args)

    from bcp_amd import synth, train_step
    from bcp_amd.dp import DataParallel
    from bcp_amd.hip_ops import Ops

    dp = DataParallel()
    assert dp.world == args.gpus or (args.gpus == 1 and dp.world == 1), f"--gpus {args.gpus} but WORLD_SIZE={dp.world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    dev = torch.device("cuda", dp.local_rank)
    torch.cuda.set_device(dev)
    Ops.product()
    seed = 1337 + dp.rank
    np.random.seed(seed)            # context_mask draws from the global numpy RNG as the reference does
    model, ema_model = build_models(dev, 1337)
    dp.broadcast_params(model)
    dp.broadcast_params(ema_model)
    opt = train_step.FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
    vol, lab = synth.la_batch(args.batch_size, seed=seed)
    vol, lab = vol.to(dev), lab.to(dev)

    def step():
        return train_step.la_self_train_step(model, ema_model, opt, vol, lab, args.labeled_bs, dp=dp if dp.enabled else None)

    for _ in range(args.warmup):
        step()
    dp.barrier()
    torch.cuda.synchronize()
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

    if dp.rank == 0:
        ms = dt / args.steps * 1e3
        global_batch = args.batch_size * dp.world
        value = global_batch * args.steps / dt
        roof = dominant_kernel_roofline(dev) if not args.no_roofline else {"achieved": None}
        step_tflops = value * STEP_GFLOP_PER_VOLUME / 1e3 / dp.world
        out = {
            "metric": "training volumes/sec (LA 112x112x80 V-Net, BCP self-training step)",
            "value": round(value, 3), "unit": "volumes/s", "n_gpus": dp.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"LA 3D V-Net BCP self-train step, per-GPU batch {args.batch_size} ({args.labeled_bs} labeled), "
                                   "112x112x80 patches, SGD m0.9 wd1e-4, EMA 0.99 (BASELINE.json configs[1])",
                       "global_batch": global_batch, "parallelism": f"dp{dp.world}", "last_loss": round(loss, 6)},
            "roofline": roof,
            "step_flops": {"gflop_per_volume": STEP_GFLOP_PER_VOLUME, "achieved_tflops_per_gpu": round(step_tflops, 2),
                           "frac_of_f32_mfma_peak": round(step_tflops / PEAK_F32_MFMA_TFLOPS, 4)},
        }
        print(f"[bench] gpu: {value:.2f} volumes/s, {ms:.2f} ms/step (host enqueue {t_enq / args.steps * 1e3:.2f} ms/step), "
              f"dominant kernel {roof['achieved']} TFLOP/s", file=sys.stderr, flush=True)
        if dp.world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.batch_size, args.labeled_bs)
        print(json.dumps(out), flush=True)
    dp.shutdown()


if __name__ == "__main__":
    main()
