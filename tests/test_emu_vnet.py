"""Whole-network parity on the host simulator (CPU): the HIP-scheduled V-Net forward + hand-scheduled backward
and the LA self-training step against the oracle / the reference's golden vectors."""
import pytest
import torch

import net_checks as NC
from test_emu_kernels import emu_ops  # noqa: F401  (fixture)

CPU = torch.device("cpu")


def test_vnet_la_golden_tiny(emu_ops, golden_dir):
    NC.check_vnet_golden_tiny(emu_ops, CPU, golden_dir)


@pytest.mark.extended
def test_vnet_la_smooth_grads(emu_ops):
    NC.check_vnet_smooth(emu_ops, CPU, shape=(32, 32, 16), N=2)


@pytest.mark.extended
def test_vnet_pancreas_smooth(emu_ops):
    NC.check_vnet_smooth(emu_ops, CPU, shape=(32, 32, 32), variant="pancreas")


@pytest.mark.extended
def test_la_self_train_trajectory(emu_ops, golden_dir):
    NC.check_la_step(emu_ops, CPU, golden_dir)


@pytest.mark.extended
def test_grouped_forward_equals_separate_calls(emu_ops):
    NC.check_grouped_equals_separate(emu_ops, CPU)


@pytest.mark.extended
def test_sliding_window_validation(emu_ops, golden_dir):
    NC.check_sliding_window(emu_ops, CPU, golden_dir)


@pytest.mark.extended
def test_sliding_window_validation_pancreas(emu_ops, golden_dir):
    NC.check_sliding_window_pancreas(emu_ops, CPU, golden_dir)


@pytest.mark.extended
def test_pre_train_steps(emu_ops):
    NC.check_pre_train_steps(emu_ops, CPU)


@pytest.mark.extended
def test_la_step_reference_default_batch(emu_ops):
    NC.check_la_step_batch8(emu_ops, CPU)


@pytest.mark.extended
def test_pancreas_self_train_step(emu_ops):
    NC.check_pancreas_step(emu_ops, CPU, modes=(True,))   # (grouped == four separate calls is a GPU test: tests/test_gpu_scripts.py)



def test_optimizer_state_is_torch_format_la(emu_ops, golden_dir):
    NC.check_opt_state_compat(emu_ops, CPU, golden_dir, variant="la")


def test_optimizer_state_is_torch_format_pancreas(emu_ops, golden_dir):
    NC.check_opt_state_compat(emu_ops, CPU, golden_dir, variant="pancreas")


def test_dropout_streams_are_per_network():
    NC.check_dropout_streams(CPU)


@pytest.mark.extended
def test_la_five_step_trajectory(emu_ops, golden_dir):
    NC.check_la_traj5(emu_ops, CPU, golden_dir)


@pytest.mark.extended
def test_vnet_la_standard_regime_gradients_on_hip_pattern(emu_ops):
    NC.check_vnet_pattern_grads(emu_ops, CPU, "la", (32, 32, 16))


@pytest.mark.extended
def test_vnet_pancreas_standard_regime_gradients_on_hip_pattern(emu_ops):
    NC.check_vnet_pattern_grads(emu_ops, CPU, "pancreas", (32, 32, 32))


@pytest.mark.extended
def test_la_loop_body_as_the_reference_writes_it(emu_ops, golden_dir):
    NC.check_la_unfused_loop(emu_ops, CPU, golden_dir, steps=2)


@pytest.mark.extended
def test_recorded_launch_plans_equal_eager_path(emu_ops):
    from bcp_amd.utils import BCP_utils as BU
    BU.set_test_ops(emu_ops)
    NC.check_launch_plans(emu_ops, CPU, steps=2, cases=(("la", False),))      # unfused: plan + busy-plan fallback (all four workloads: the GPU suite)


@pytest.mark.extended
def test_volatile_io_replays_equal_eager_path(emu_ops):
    """round 5: the networks as the training scripts run them (volatile_io: no copies in and out of the recorded passes) == the eager path"""
    from bcp_amd.utils import BCP_utils as BU
    BU.set_test_ops(emu_ops)
    NC.check_launch_plans(emu_ops, CPU, steps=3, cases=(("la", True),), volatile=True)


@pytest.mark.extended
def test_up_recompute_forced_everywhere(emu_ops):
    """round 6 (built, measured, off by default): every transposed conv + norm the shape allows on the recomputing pair bcp_up_fwd_norm /
    bcp_up_norm_bwd -- every gradient tensor within 1e-4 of the fp64 oracle linearised on the HIP activation pattern"""
    from bcp_amd.utils import BCP_utils as BU
    BU.set_test_ops(emu_ops)
    from bcp_amd.networks.VNet import VNet as _VN
    emu_ops.set_option("up_recompute", 1)
    _VN.UP_RECOMPUTE_GRAD = True      # (the product takes the recomputing pair only in forwards without a backward pass: the teacher's)
    try:
        NC.check_vnet_pattern_grads(emu_ops, CPU, "la", (32, 32, 16))
    finally:
        emu_ops.set_option("up_recompute")
        _VN.UP_RECOMPUTE_GRAD = False


def test_partial_weight_packs(emu_ops):
    """round 6: only the observed sections of the weight packs are written in front of replays; an eager pass behind one repacks everything"""
    from bcp_amd.utils import BCP_utils as BU
    BU.set_test_ops(emu_ops)
    NC.check_partial_packs(emu_ops, CPU)


@pytest.mark.extended
def test_head_fused_with_last_norm_equals_separate_apply(emu_ops):
    """VNet.fuse_head: block_nine's norm + ReLU + Dropout3d applied inside the 1x1x1 head (its activation never stored)"""
    NC.check_fused_head(emu_ops, CPU, steps=1)


@pytest.mark.extended
def test_vnet_second_output_is_pooled_x5(emu_ops):
    NC.check_vnet_features(emu_ops, CPU)
