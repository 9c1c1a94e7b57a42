"""Data-parallel path (bcp_amd/dp.py) with world_size 2 over gloo on CPU: N ranks == N sequential micro-batches
with averaged gradients (BatchNorm statistics stay rank-local).  The kernels run on the host
simulator here; the collective, the flat gradient bucket, the 1/world scaling inside the fused SGD launch and
the EMA are the product code."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "_emu", "libbcp_emu.so")
SHAPE = (32, 32, 16)


def _setup():
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bcp_oracle as O
    import net_checks as NC
    from bcp_amd import _lib, train_step
    from bcp_amd.hip_ops import Ops
    ops = Ops(_lib.Binding(EMU), allow_cpu=True)
    return O, NC, train_step, ops


def _inputs(O, rank):
    vol, lab = O.synth_la_batch(4, shape=SHAPE, seed=100 + rank)
    rng = np.random.default_rng(200 + rank)
    drops = {k: {"x5": torch.from_numpy((rng.random((1, 256)) < 0.5).astype(np.float32)),
                 "x9": torch.from_numpy((rng.random((1, 16)) < 0.5).astype(np.float32))} for k in ("t_a", "t_b", "s_l", "s_u")}
    box = (2 + rank, 4, 1 + rank, 21, 21, 10)
    return vol, lab, drops, box


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ["BCP_EMU_THREADS"] = "4"
    torch.set_num_threads(2)
    O, NC, train_step, ops = _setup()
    from bcp_amd.dp import DataParallel
    dp = DataParallel(backend="gloo")
    P = O.init_params(O.vnet_param_shapes(), seed=7 + rank, random_affine=True)   # different per rank: broadcast must fix it
    model, ema = NC.make_vnet(P, torch.device("cpu"), ops), NC.make_vnet(P, torch.device("cpu"), ops)
    dp.broadcast_params(model)
    dp.broadcast_params(ema)
    opt = train_step.FlatSGD(model, lr=0.01)
    vol, lab, drops, box = _inputs(O, rank)
    r = train_step.la_self_train_step(model, ema, opt, vol, lab, 2, box=box, drops=drops, dp=dp)
    torch.save({"flat": model.flat_params().clone(), "ema": ema.flat_params().clone(), "loss": float(r["loss"]),
                "collectives": dp.n_collectives}, os.path.join(out_dir, f"rank{rank}.pt"))
    dp.shutdown()


@pytest.mark.slow
def test_dp2_equals_two_averaged_microbatches(tmp_path):
    if not os.path.exists(EMU):
        import subprocess
        subprocess.check_call([os.path.join(ROOT, "tools", "emu", "build_emu.sh")])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["flat"], r1["flat"]), "students must be identical on every rank after the step"
    assert torch.equal(r0["ema"], r1["ema"]), "teachers stay identical because the students do"
    assert r0["collectives"] == r1["collectives"] >= 3, "37.8 MB of gradients in >= 8 MB buckets: several collectives, same count on every rank"
    # single-process reference: two micro-batches from the same start, gradients averaged, one SGD step
    O, NC, train_step, ops = _setup()
    P = O.init_params(O.vnet_param_shapes(), seed=7, random_affine=True)   # rank 0's weights were broadcast
    dev = torch.device("cpu")
    grads, losses = [], []
    for rank in range(2):
        m, e = NC.make_vnet(P, dev, ops), NC.make_vnet(P, dev, ops)
        vol, lab, drops, box = _inputs(O, rank)
        r = train_step.la_self_train_step(m, e, None, vol, lab, 2, box=box, drops=drops)
        grads.append(m.flat_trainable()[1].clone())
        losses.append(float(r["loss"]))
    m = NC.make_vnet(P, dev, ops)
    opt = train_step.FlatSGD(m, lr=0.01)
    m.begin_backward()
    m.flat_trainable()[1].copy_(grads[0] + grads[1])
    opt.grad_scale = 0.5
    opt.step()
    assert abs(losses[0] - r0["loss"]) < 1e-6 and abs(losses[1] - r1["loss"]) < 1e-6
    diff = float((m.flat_params() - r0["flat"]).abs().max())
    assert diff < 1e-7, f"DP step differs from the averaged-micro-batch step by {diff}"


def _worker_acdc(rank, world, port, out_dir, bucket_mb):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      BCP_DP_BUCKET_MB=bucket_mb)
    os.environ["BCP_EMU_THREADS"] = "4"
    torch.set_num_threads(2)
    O, NC, train_step, ops = _setup()
    from bcp_amd.dp import DataParallel
    dp = DataParallel(backend="gloo")
    P = O.init_params(O.unet_param_shapes(), seed=11 + rank, random_affine=True)
    model, ema = NC.make_unet(P, torch.device("cpu"), ops), NC.make_unet(P, torch.device("cpu"), ops)
    for p in ema.parameters():
        p.detach_()
    dp.broadcast_params(model)
    dp.broadcast_params(ema)
    opt = train_step.FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
    vol, lab = O.synth_acdc_batch(8, shape=(64, 64), seed=300 + rank)
    rng = np.random.default_rng(400 + rank)
    drops = {k: {f"d{i}": torch.from_numpy((rng.random((2, c, 64 >> i, 64 >> i)) < 0.8).astype(np.float32)) for i, c in enumerate(O.UNET_CH)}
             for k in ("t_a", "t_b", "s_unl", "s_l")}
    r = train_step.acdc_self_train_step(model, ema, opt, vol, lab, 4, box=(5 + rank, 9, 42, 42), drops=drops, dp=dp)
    # parameters only: BatchNorm running statistics are rank-local (dp.py), and update_model_ema carries them into the teacher
    torch.save({"flat": model.flat_params().clone(), "ema": ema.flat_params().clone(), "loss": float(r["loss"]),
                "collectives": dp.n_collectives}, os.path.join(out_dir, f"acdc{bucket_mb}_rank{rank}.pt"))
    dp.shutdown()


@pytest.mark.extended
def test_dp2_unet_buckets_equal_single_allreduce(tmp_path):
    """2-D U-Net (7.26 MB of gradients): 1 MB buckets started inside the backward pass == one all-reduce after it, bit for bit,
    and both ranks end the step with the same student and teacher"""
    out = {}
    for mb in ("1", "0"):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_worker_acdc, args=(2, port, str(tmp_path), mb), nprocs=2, join=True)
        out[mb] = [torch.load(tmp_path / f"acdc{mb}_rank{r}.pt") for r in range(2)]
        assert torch.equal(out[mb][0]["flat"], out[mb][1]["flat"]) and torch.equal(out[mb][0]["ema"], out[mb][1]["ema"])
    assert out["1"][0]["collectives"] >= 3 and out["0"][0]["collectives"] == 1
    assert torch.equal(out["1"][0]["flat"], out["0"][0]["flat"]), "bucketed and single all-reduce must give identical steps"
    assert out["1"][0]["loss"] == out["0"][0]["loss"]


# ---------------------------------------------------------------------------------------------- C5: pancreas, 4 ranks, Adam
PSHAPE = (16, 16, 16)


def _pancreas_inputs(O, rank):
    vol, lab = O.synth_la_batch(4, shape=PSHAPE, seed=500 + rank)
    return vol, lab, (1 + rank % 2, 2, 1 + rank // 2, 10, 10, 10)


def _worker_pancreas(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      BCP_DP_BUCKET_MB="8")
    os.environ["BCP_EMU_THREADS"] = "2"
    torch.set_num_threads(1)
    O, NC, train_step, ops = _setup()
    from bcp_amd.dp import DataParallel
    dp = DataParallel(backend="gloo")
    P = O.init_params(O.vnet_param_shapes(variant="pancreas"), seed=21 + rank, random_affine=True)     # different per rank: the broadcast fixes it
    model = NC.make_vnet(P, torch.device("cpu"), ops, variant="pancreas", has_dropout=False)
    ema = NC.make_vnet(P, torch.device("cpu"), ops, variant="pancreas", has_dropout=False)
    for p in ema.parameters():
        p.detach_()
    dp.broadcast_params(model)
    dp.broadcast_params(ema)
    opt = train_step.FlatAdam(model, lr=1e-3)
    vol, lab, box = _pancreas_inputs(O, rank)
    r = train_step.la_self_train_step(model, ema, opt, vol, lab, 2, box=box, variant="pancreas", connect_mode=2, dp=dp)
    torch.save({"flat": model.flat_params().clone(), "ema": ema.flat_params().clone(), "loss": float(r["loss"]),
                "collectives": dp.n_collectives}, os.path.join(out_dir, f"panc_rank{rank}.pt"))
    dp.shutdown()


@pytest.mark.slow
@pytest.mark.extended
def test_dp4_pancreas_adam_equals_four_averaged_microbatches(tmp_path):
    """BASELINE.json configs[4]'s partitioning: FOUR ranks, each with its own four pancreas streams and box, the
    InstanceNorm V-Net and Adam -- after one step every rank holds the same student / teacher, equal to one Adam step on the average
    of the four ranks' gradients (tiny 16^3 patches on the host simulator; the collective, buckets and 1/world scaling are the product's)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_pancreas, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    rs = [torch.load(tmp_path / f"panc_rank{r}.pt") for r in range(4)]
    for r in rs[1:]:
        assert torch.equal(rs[0]["flat"], r["flat"]) and torch.equal(rs[0]["ema"], r["ema"]), "ranks diverged"
        assert r["collectives"] == rs[0]["collectives"] >= 1
    O, NC, train_step, ops = _setup()
    P = O.init_params(O.vnet_param_shapes(variant="pancreas"), seed=21, random_affine=True)
    dev = torch.device("cpu")
    gsum = None
    for rank in range(4):
        m = NC.make_vnet(P, dev, ops, variant="pancreas", has_dropout=False)
        e = NC.make_vnet(P, dev, ops, variant="pancreas", has_dropout=False)
        vol, lab, box = _pancreas_inputs(O, rank)
        r = train_step.la_self_train_step(m, e, None, vol, lab, 2, box=box, variant="pancreas", connect_mode=2)
        assert abs(float(r["loss"]) - rs[rank]["loss"]) < 1e-6
        g = m.flat_trainable()[1].clone()
        gsum = g if gsum is None else gsum + g
    m = NC.make_vnet(P, dev, ops, variant="pancreas", has_dropout=False)
    opt = train_step.FlatAdam(m, lr=1e-3)
    m.begin_backward()
    m.flat_trainable()[1].copy_(gsum)
    opt.grad_scale = 0.25
    opt.step()
    # Adam's first update is lr * g / (|g| + eps): compare the UPDATES (summation order of the four ranks' gradients differs between a ring
    # all-reduce and the sequential sum, and elements with |g| ~ eps amplify that)
    P0 = NC.make_vnet(P, dev, ops, variant="pancreas", has_dropout=False).flat_params()
    du, dr = m.flat_params() - P0, rs[0]["flat"] - P0
    frac_close = float(((du - dr).abs() < 1e-5).float().mean())
    assert frac_close > 0.999, f"only {frac_close:.5f} of the Adam updates agree with the averaged-micro-batch step"


# ---------------------------------------------------------------------------------------------- RCCL unique-id exchange (no GPU needed)
def _worker_id(rank, world, port, out_dir, id_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      BCP_DP_LAUNCH_ID="t%d" % port)
    if id_dir:
        os.environ["BCP_DP_ID_DIR"] = id_dir
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from bcp_amd import dp
    got = []
    for k in range(2):            # two communicators in one launch: independent keys
        ident = bytes([(17 * k + i) % 251 for i in range(128)]) if rank == 0 else None
        got.append(dp._exchange_id(ident, world, rank))
    torch.save(got, os.path.join(out_dir, f"id_{'file' if id_dir else 'store'}_{rank}.pt"))


@pytest.mark.parametrize("transport", ["store", "file"])
def test_rccl_unique_id_exchange(tmp_path, transport):
    """bcp_amd/dp.py::_exchange_id: rank 0's 128-byte id reaches every rank over the c10d store at MASTER_ADDR:MASTER_PORT (default;
    multi-node capable) or, with BCP_DP_ID_DIR, through a per-launch file -- where a stale file of an earlier launch is never accepted"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    id_dir = ""
    if transport == "file":
        id_dir = str(tmp_path / "ids")
        os.makedirs(id_dir)
        stale = os.path.join(id_dir, f"bcp_rccl_id_{port}_none_3_t{port}_1")      # same name as this launch's first id, left by a "crashed" run
        open(stale, "wb").write(b"\xff" * 128)
        os.utime(stale, (1, 1))
    mp.spawn(_worker_id, args=(3, port, str(tmp_path), id_dir), nprocs=3, join=True)
    res = [torch.load(tmp_path / f"id_{transport}_{r}.pt") for r in range(3)]
    for k in range(2):
        want = bytes([(17 * k + i) % 251 for i in range(128)])
        assert all(r[k] == want for r in res), f"communicator {k}: ids differ"


def test_rccl_unique_id_exchange_under_torchrun(tmp_path):
    """the same exchange inside a REAL `python -m torch.distributed.run` launch (the driver's command line for bench.py --gpus N): there
    the env:// rendezvous is the elastic agent's store (TORCHELASTIC_USE_AGENT_STORE), not a TCPStore served by rank 0"""
    import subprocess
    script = tmp_path / "idx.py"
    script.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from bcp_amd import dp\n"
        "rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "for k in range(2):\n"
        "    want = bytes([(13 * k + i) % 251 for i in range(128)])\n"
        "    assert dp._exchange_id(want if rank == 0 else None, world, rank) == want\n"
        "print('ID_OK', rank, flush=True)\n")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.pop("BCP_DP_ID_DIR", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and r.stdout.count("ID_OK") == 2, (r.stdout[-2000:], r.stderr[-2000:])


def _worker_agree(rank, world, port, out_dir, id_dir, bad_rank):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      BCP_DP_LAUNCH_ID=f"t{port}")
    if id_dir:
        os.environ["BCP_DP_ID_DIR"] = id_dir
    else:
        os.environ.pop("BCP_DP_ID_DIR", None)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from bcp_amd import dp
    got = []
    # communicator 1: every rank got its id; communicator 2: ONE rank failed in its rank-local half (it never ran the id exchange)
    ident = bytes(range(128))
    assert dp._exchange_id(ident if rank == 0 else None, world, rank) == ident
    got.append(dp._agree(True, world, rank))
    got.append(dp._agree(rank != bad_rank, world, rank))
    torch.save(got, os.path.join(out_dir, f"agree_{rank}.pt"))


@pytest.mark.parametrize("transport", ["store", "file"])
def test_rank_local_failure_is_agreed_on_before_the_collective_init(tmp_path, transport):
    """bcp_amd/dp.py::_agree: between the rank-local half of building the RCCL communicator and ncclCommInitRank every rank
    learns whether EVERY rank got through -- one failing rank sends all of them the same way, nobody is left inside the collective"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    id_dir = ""
    if transport == "file":
        id_dir = str(tmp_path / "ids")
        os.makedirs(id_dir)
        # a FRESH flag of an earlier launch with the same tag saying "rank 1 was fine" for the second agreement: it carries
        # that launch's nonce, not this one's, and must not be read
        open(os.path.join(id_dir, f"bcp_rccl_ok_{port}_none_3_t{port}_2.0123456789ab.1"), "wb").write(b"1")
    mp.spawn(_worker_agree, args=(3, port, str(tmp_path), id_dir, 1), nprocs=3, join=True)
    res = [torch.load(tmp_path / f"agree_{r}.pt") for r in range(3)]
    assert res == [[True, False]] * 3, res
