"""-m gpu: the train-script counterparts run end to end (pre-train -> checkpoint -> self-train) on the GPU."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_la_script(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from bcp_amd import LA_BCP_train as T
    T.main(["--labelnum", "8", "--batch_size", "4", "--labeled_bs", "2", "--pre_max_iteration", "3", "--self_max_iteration", "4", "--log_every", "1"])
    sd = torch.load(tmp_path / "model/BCP/LA_BCP_8_labeled/self_train/VNet_best_model.pth")
    assert len(sd) == 259 and all(torch.isfinite(v.float()).all() for v in sd.values())


def test_acdc_script(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from bcp_amd import ACDC_BCP_train as T
    T.main(["--labelnum", "7", "--batch_size", "24", "--labeled_bs", "12", "--pre_iterations", "3", "--max_iterations", "3", "--log_every", "1"])
    sd = torch.load(tmp_path / "model/BCP/ACDC_BCP_7_labeled/self_train/unet_best_model.pth")
    assert len(sd) == 226 and all(torch.isfinite(v.float()).all() for v in sd.values())


def test_pancreas_script():
    from bcp_amd.pancreas import train_pancreas as T
    T.main(["--pretraining_epochs", "1", "--self_training_epochs", "1", "--steps_per_epoch", "2", "--batch_size", "1"])
