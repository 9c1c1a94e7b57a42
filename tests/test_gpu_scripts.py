"""-m gpu: the train-script counterparts run end to end (pre-train -> checkpoint -> self-train) on the GPU."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_la_script(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from bcp_amd import LA_BCP_train as T
    T.main(["--labelnum", "8", "--batch_size", "4", "--labeled_bs", "2", "--pre_max_iteration", "3", "--self_max_iteration", "4", "--log_every", "1",
            "--val_every", "2", "--val_cases", "1"])       # sliding-window validation (eval-mode net) twice per phase
    sd = torch.load(tmp_path / "model/BCP/LA_BCP_8_labeled/self_train/VNet_best_model.pth")
    assert len(sd) == 259 and all(torch.isfinite(v.float()).all() for v in sd.values())


def test_acdc_script(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from bcp_amd import ACDC_BCP_train as T
    T.main(["--labelnum", "7", "--batch_size", "24", "--labeled_bs", "12", "--pre_iterations", "3", "--max_iterations", "3", "--log_every", "1",
            "--val_every", "2", "--val_cases", "1"])
    sd = torch.load(tmp_path / "model/BCP/ACDC_BCP_7_labeled/self_train/unet_best_model.pth")
    assert len(sd) == 226 and all(torch.isfinite(v.float()).all() for v in sd.values())


def test_scripts_with_device_input_pipeline(tmp_path, monkeypatch):
    """--augment: raw-size synthetic cases + the device-side transforms (DeviceRotFlipCrop / DeviceRandomGenerator)
    in front of the same loops"""
    monkeypatch.chdir(tmp_path)
    from bcp_amd import ACDC_BCP_train as TA
    from bcp_amd import LA_BCP_train as TL
    TL.main(["--labelnum", "8", "--batch_size", "4", "--labeled_bs", "2", "--pre_max_iteration", "2", "--self_max_iteration", "2", "--log_every", "1",
             "--augment", "--exp", "aug"])
    TA.main(["--labelnum", "7", "--batch_size", "24", "--labeled_bs", "12", "--pre_iterations", "2", "--max_iterations", "2", "--log_every", "1",
             "--augment", "--exp", "aug"])


def test_pancreas_script(tmp_path):
    from bcp_amd.pancreas import train_pancreas as T
    T.main(["--pretraining_epochs", "1", "--self_training_epochs", "1", "--steps_per_epoch", "2", "--batch_size", "1"])
    # with the validation gate (pancreas/test_util.py:test_calculate_metric, sliding window over a 104x100x96 case) and the
    # reference's checkpoint hand-off: best pre-trained {'net','opt','epoch'} -> both nets -> best self-trained {'net'}
    T.main(["--pretraining_epochs", "2", "--self_training_epochs", "2", "--steps_per_epoch", "1", "--batch_size", "1", "--val_every", "1",
            "--val_stride", "48", "48", "--result_dir", str(tmp_path / "cutmix")])
    # the four loader streams of the reference with the crops done on the device (pancreas/dataloaders.py RandomCrop / CenterCrop)
    T.main(["--pretraining_epochs", "1", "--self_training_epochs", "1", "--steps_per_epoch", "2", "--batch_size", "1", "--val_every", "0",
            "--device_input_pipeline", "1"])
    pre = torch.load(tmp_path / "cutmix/pretrain/best_ema20_pre.pth")
    assert set(pre) == {"net", "opt", "epoch"} and len(pre["net"]) == 60
    st = torch.load(tmp_path / "cutmix/self_train/best_ema_20_self.pth")
    assert len(st["net"]) == 60 and all(torch.isfinite(v.float()).all() for v in st["net"].values())


def test_pancreas_grouped_step_matches_two_calls():
    """ema_cutmix with the teacher / student sub-batches launched as one grouped forward each == the script's four
    separate network calls (InstanceNorm statistics are per sample); only the order in which the two students' weight
    gradients are added differs, so parameters agree to fp32 rounding after a step"""
    import numpy as np
    from bcp_amd import train_step
    from bcp_amd.pancreas import train_pancreas as T
    from bcp_amd.pancreas.Vnet import create_Vnet
    dev = torch.device("cuda", 0)
    res = []
    for grouped in (True, False):
        np.random.seed(7)
        torch.manual_seed(7)
        net, ema = create_Vnet(), create_Vnet(ema=True)
        ema.load_state_dict(net.state_dict())
        opt = train_step.FlatAdam(net, lr=1e-3)
        streams = T._streams(dev, 4, 1, seed=11)
        loss = T.ema_cutmix(net, ema, opt, streams, 1, grouped=grouped)
        res.append((float(loss), net.flat_params().clone(), ema.flat_params().clone()))
    (l0, p0, e0), (l1, p1, e1) = res
    assert abs(l0 - l1) < 1e-5, (l0, l1)
    # Adam's first step moves every weight by ~lr * sign(g): compare the updates, not the weights
    assert float((p0 - p1).abs().max()) < 2e-3 and float((p0 - p1).abs().mean()) < 2e-5
    assert float((e0 - e1).abs().max()) < 1e-4


_DP1 = r"""
import os, sys, socket
import numpy as np, torch
sys.path.insert(0, os.environ["BCP_ROOT"])
import bench
from bcp_amd import synth, train_step
from bcp_amd.dp import DataParallel
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dp = DataParallel(force=True)                     # one-rank RCCL communicator
vol, lab = synth.la_batch(4, seed=5); vol, lab = vol.to(dev), lab.to(dev)
out = []
for use_dp in (True, False):
    np.random.seed(3)
    model, ema = bench.build_models(dev, 21)
    for m in (model, ema):
        m._drop_seed = 99
    opt = train_step.FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
    for it in range(3):
        r = train_step.la_self_train_step(model, ema, opt, vol, lab, 2, dp=dp if use_dp else None)
    torch.cuda.synchronize()
    out.append((float(r["loss"]), model.flat_params().clone(), ema.flat_params().clone()))
assert dp.n_collectives >= 9, dp.n_collectives    # 3 steps x (>= 3 buckets of the 37.8 MB gradient buffer)
(l0, p0, e0), (l1, p1, e1) = out
d = float((p0 - p1).abs().max())
assert abs(l0 - l1) < 1e-5 and d < 1e-5 and float((e0 - e1).abs().max()) < 1e-5, (l0, l1, d)
dp.shutdown()
print("DP1 OK", dp.n_collectives, l0, l1, d)
"""


def test_dp_bucketed_allreduce_over_rccl_one_rank():
    """the bucketed gradient exchange on the real backend (RCCL, world_size 1: the sums are identities, the stream ordering
    between the weight-gradient side stream, the communicator's stream and the optimiser is the production one): three LA
    steps with it == three steps without.  World sizes > 1 are covered on CPU over gloo (tests/test_dp_gloo.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # "rccl": the library's own communicator through the C ABI (bcp_comm_init_rank / bcp_allreduce_f32); "nccl": torch.distributed
    for backend in ("rccl", "nccl"):
        env = dict(os.environ, BCP_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0", BCP_DP_BACKEND=backend)
        r = subprocess.run([sys.executable, "-c", _DP1], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "DP1 OK" in r.stdout, backend + ": " + r.stdout[-2000:] + r.stderr[-4000:]


_DP2 = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["BCP_ROOT"]); sys.path.insert(0, os.path.join(os.environ["BCP_ROOT"], "oracle"))
import bcp_oracle as O
from bcp_amd import train_step
from bcp_amd.dp import DataParallel
from bcp_amd.networks.VNet import VNet
rank = int(os.environ["RANK"])
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dp = DataParallel(backend="gloo")                  # two ranks share the one GPU of the box: gloo moves the CUDA buckets through the host
P = O.init_params(O.vnet_param_shapes(), seed=7 + rank, random_affine=True)      # different per rank: the broadcast must fix it
nets = []
for _ in range(2):
    n = VNet(n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True).to(dev)
    n.load_state_dict({k: P[k].clone() for k in n.state_dict()})
    nets.append(n.flatten_().train())
model, ema = nets
for p in ema.parameters():
    p.detach_()
dp.broadcast_params(model); dp.broadcast_params(ema)
opt = train_step.FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
vol, lab = O.synth_la_batch(4, shape=(32, 32, 16), seed=100 + rank)
vol, lab = vol.to(dev), lab.to(dev)
for m in (model, ema):
    m._drop_seed = 5 + rank
np.random.seed(11 + rank)
for it in range(3):
    r = train_step.la_self_train_step(model, ema, opt, vol, lab, 2, box=(2 + rank, 4, 1 + it, 21, 21, 10), dp=dp)
torch.cuda.synchronize()
torch.save({"flat": model.flat_params().cpu(), "ema": ema.flat_params().cpu(), "loss": float(r["loss"]), "n": dp.n_collectives},
           os.path.join(os.environ["BCP_OUT"], f"r{rank}_mb{os.environ['BCP_DP_BUCKET_MB']}.pt"))
dp.shutdown()
"""


def test_dp_two_ranks_on_one_gpu_buckets_equal_single_allreduce(tmp_path):
    """two REAL ranks (two processes, gloo over CUDA tensors, sharing cuda:0) run three LA steps with the bucketed exchange issued
    from inside the backward pass next to the weight-gradient side stream: both ranks end with identical students and teachers,
    and the result equals, bit for bit, the run with ONE all-reduce after the backward pass (BCP_DP_BUCKET_MB=0)"""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mb in ("8", "0"):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        procs = []
        for rank in range(2):
            env = dict(os.environ, BCP_ROOT=root, BCP_OUT=str(tmp_path), RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2",
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BCP_DP_BUCKET_MB=mb)
            procs.append(subprocess.Popen([sys.executable, "-c", _DP2], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate(timeout=600)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
        res[mb] = [torch.load(tmp_path / f"r{r}_mb{mb}.pt") for r in range(2)]
        assert torch.equal(res[mb][0]["flat"], res[mb][1]["flat"]) and torch.equal(res[mb][0]["ema"], res[mb][1]["ema"])
    assert res["8"][0]["n"] >= 9 and res["0"][0]["n"] == 3
    assert torch.equal(res["8"][0]["flat"], res["0"][0]["flat"]) and res["8"][0]["loss"] == res["0"][0]["loss"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (every gpurun box of rounds 1-4 had one): the first multi-GPU lease runs it")
def test_two_ranks_over_rccl_through_the_c_abi():
    """tools/rccl_smoke.py as a test -- two real ranks (one process per GPU) over bcp_comm_init_rank /
    bcp_allreduce_f32 (RCCL through the C ABI, unique id over the torch.distributed rendezvous store): the communicator spans both
    ranks, an all-reduce equals its closed form, and after three LA self-training steps with per-rank data the students and teachers
    of all ranks are bit-identical to rank 0's.  world-2 / world-4 equality with sequential micro-batches is covered on CPU over gloo
    (tests/test_dp_gloo.py); this is the same exchange on the real transport."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BCP_DP_BACKEND="rccl")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tools", "rccl_smoke.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "rccl_smoke OK: 2 ranks" in r.stdout, r.stdout[-3000:]
