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


def test_pancreas_script():
    from bcp_amd.pancreas import train_pancreas as T
    T.main(["--pretraining_epochs", "1", "--self_training_epochs", "1", "--steps_per_epoch", "2", "--batch_size", "1"])


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
