"""U-Net (ACDC) whole-network and self-training-step parity on the host simulator (CPU)."""
import json
import os

import pytest
import torch

import net_checks as NC
from test_emu_kernels import emu_ops  # noqa: F401  (fixture)

CPU = torch.device("cpu")


def test_unet_state_dict_keys(golden_dir):
    from bcp_amd.networks.unet import UNet_2d
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))
    n = UNet_2d(in_chns=1, class_num=4)
    assert [[k, list(v.shape)] for k, v in n.state_dict().items()] == meta["unet_keys"]
    assert [k for k, _ in n.named_parameters()] == meta["unet_param_names"]


def test_unet_golden_tiny(emu_ops, golden_dir):
    NC.check_unet_golden_tiny(emu_ops, CPU, golden_dir)


def test_dropout_bits_evaluated_in_the_norm_kernels_equal_the_mask_tensors(emu_ops):
    NC.check_unet_inline_dropout(emu_ops, CPU, hw=(32, 32), N=4)


def test_skip_written_into_the_concat_buffer_equals_the_copy(emu_ops):
    NC.check_unet_skip_in_concat(emu_ops, CPU, hw=(32, 32), N=4)


@pytest.mark.extended
def test_unet_smooth_grads(emu_ops):
    NC.check_unet_smooth(emu_ops, CPU)


@pytest.mark.extended
def test_acdc_self_train_trajectory(emu_ops, golden_dir):
    NC.check_acdc_step(emu_ops, CPU, golden_dir)


def test_unet_eval_mode(emu_ops):
    NC.check_unet_eval(emu_ops, CPU)


@pytest.mark.extended
def test_val_2d_single_volume(emu_ops):
    NC.check_val_2d(emu_ops, CPU)


def test_update_model_ema_state_dict_semantics(emu_ops):
    """A10 (ACDC flavour): update_model_ema averages the WHOLE state_dict -- parameters, BatchNorm running statistics, and the int64
    num_batches_tracked, which goes through float32 and is truncated on load (ACDC_BCP_train.py:123-129) -- vs the oracle"""
    import numpy as np
    import bcp_oracle as O
    from bcp_amd import train_step
    rng = np.random.default_rng(31)
    Ps, Pt = O.init_params(O.unet_param_shapes(), seed=41, random_affine=True), O.init_params(O.unet_param_shapes(), seed=42, random_affine=True)
    for P, nbt in ((Ps, 3), (Pt, 250)):
        for k in P:
            if k.endswith("running_mean"):
                P[k] = torch.from_numpy(rng.normal(0.0, 0.3, tuple(P[k].shape)).astype(np.float32))
            elif k.endswith("running_var"):
                P[k] = torch.from_numpy(rng.uniform(0.5, 1.5, tuple(P[k].shape)).astype(np.float32))
            elif k.endswith("num_batches_tracked"):
                P[k] = torch.tensor(nbt, dtype=torch.int64)
    student, teacher = NC.make_unet(Ps, CPU, emu_ops), NC.make_unet(Pt, CPU, emu_ops)
    for _ in range(2):                      # twice: the truncated counter feeds the next update
        train_step.update_model_ema(student, teacher, 0.99)
        O.ema_state_dict(Ps, Pt, 0.99)
    sd = teacher.state_dict()
    live = [k for k in sd if k.startswith("encoder") or k.startswith("decoder")]
    for k in live:
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(Pt[k]), (k, int(sd[k]), int(Pt[k]))
        else:
            assert torch.equal(sd[k], Pt[k]), k       # one multiply-add per element in fp32 on both sides: bit-exact


@pytest.mark.extended
def test_acdc_five_step_trajectory(emu_ops, golden_dir):
    NC.check_acdc_traj5(emu_ops, CPU, golden_dir)


@pytest.mark.extended
def test_unet_standard_regime_gradients_on_hip_pattern(emu_ops):
    NC.check_unet_pattern_grads(emu_ops, CPU)


def test_launch_plan_hygiene(emu_ops):
    NC.check_plan_hygiene(emu_ops, torch.device("cpu"))
