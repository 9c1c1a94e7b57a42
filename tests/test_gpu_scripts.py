"""-m gpu: the train-script counterparts run end to end (pre-train -> checkpoint -> self-train) on the GPU."""
import os

import pytest
import torch

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
    env = dict(os.environ, BCP_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _DP1], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DP1 OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
