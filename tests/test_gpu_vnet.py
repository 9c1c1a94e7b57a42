"""-m gpu: the HIP V-Net and the LA self-training step on a real MI355X vs the oracle and the golden vectors
captured from the reference (same checks as tests/test_emu_vnet.py, plus config-shape cases)."""
import json
import os

import numpy as np
import pytest
import torch

import bcp_oracle as O
import kernel_checks as K
import net_checks as NC

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from bcp_amd.hip_ops import Ops
    return Ops.product()


def test_vnet_la_golden_tiny(ops, golden_dir):
    NC.check_vnet_golden_tiny(ops, DEV, golden_dir)


def test_vnet_la_smooth_grads(ops):
    NC.check_vnet_smooth(ops, DEV, shape=(32, 32, 16), N=2)


def test_vnet_pancreas_smooth(ops):
    NC.check_vnet_smooth(ops, DEV, shape=(32, 32, 32), variant="pancreas")


def test_la_self_train_trajectory(ops, golden_dir):
    NC.check_la_step(ops, DEV, golden_dir)


def test_vnet_la_full_shape_vs_reference_golden(ops, golden_dir):
    """112x112x80 forward + backward vs the checksums recorded from the reference (vnet_la_full.npz)"""
    g = np.load(os.path.join(golden_dir, "vnet_la_full.npz"))
    m = json.load(open(os.path.join(golden_dir, "meta.json")))["vnet_la_full"]
    P = O.init_params(O.vnet_param_shapes(), seed=m["param_seed"], random_affine=True)
    x, lab = O.synth_la_batch(1, seed=m["data_seed"])
    net = NC.make_vnet(P, DEV, ops)
    net.drop_masks = {"x5": torch.from_numpy(g["drop_x5"]), "x9": torch.from_numpy(g["drop_x9"])}
    out, feat = net(x.to(DEV))
    assert tuple(feat.shape) == (1, 256, 3, 3, 2)          # pool(x5), the reference's second return value (networks/VNet.py:286-290)
    from bcp_amd.utils import BCP_utils as BU
    loss = BU.sup_loss(out, lab.to(DEV))
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5, (float(loss.detach()), float(g["loss"]))
    o = out.detach().double().reshape(-1).cpu()
    st = g["logits_stats"]
    assert abs(float(o.abs().sum()) - st[1]) / st[1] < 1e-5 and abs(float((o * o).sum().sqrt()) - st[2]) / st[2] < 1e-5
    idx = torch.linspace(0, o.numel() - 1, 4096).long()
    K.close(out.detach().reshape(-1).cpu()[idx], torch.from_numpy(g["logits_sample"]), rtol=2e-4, msg="logits sample")
    # Dice of the thresholded prediction vs the synthetic label: HIP vs reference-arithmetic oracle (north_star: 1e-4)
    Po = {k: v.clone() for k, v in P.items()}
    oo = O.vnet_forward(Po, x, {"x5": torch.from_numpy(g["drop_x5"]), "x9": torch.from_numpy(g["drop_x9"])}, True, "la")
    d_ref = O.dice_metric(O.get_cut_mask(oo), lab)
    from bcp_amd import train_step
    d_hip = O.dice_metric(train_step.get_cut_mask(out.detach()).cpu(), lab)
    assert abs(d_ref - d_hip) < 1e-4, (d_ref, d_hip)
    loss.backward()
    params = dict(net.named_parameters())
    # SANITY ONLY (not a parity gate): every gradient tensor's L2 norm lands within 3 % of the reference's recorded norm -- two fp32
    # gradients of a half-ReLU network differ by 3e-3..7e-3 per tensor in the reference's own arithmetic.  The
    # full-size backward is PINNED by test_la_full_size_gradients_on_hip_activation_pattern (rel-L2 of the difference <= 1e-4).
    for n_, stg in zip([str(n) for n in g["grad_names"]], g["grad_stats"]):
        if NC.is_prenorm_bias(n_, params):
            continue
        l2 = float(params[n_].grad.double().norm())
        assert abs(l2 - stg[2]) / max(stg[2], 1e-12) < 3e-2, (n_, l2, stg[2])


def test_vnet_second_output_is_pooled_x5(ops):
    NC.check_vnet_features(ops, DEV)


def test_grouped_forward_equals_separate_calls(ops):
    NC.check_grouped_equals_separate(ops, DEV)


def test_sliding_window_validation(ops, golden_dir):
    NC.check_sliding_window(ops, DEV, golden_dir)


def test_sliding_window_validation_pancreas(ops, golden_dir):
    NC.check_sliding_window_pancreas(ops, DEV, golden_dir)


def test_pre_train_steps(ops):
    NC.check_pre_train_steps(ops, DEV)


def test_pancreas_self_train_step(ops):
    NC.check_pancreas_step(ops, DEV)


def test_la_step_reference_default_batch(ops):
    NC.check_la_step_batch8(ops, DEV)



def test_optimizer_state_is_torch_format(ops, golden_dir):
    """{'net','opt'} checkpoints interchange with torch.optim (= the reference's save_net_opt) on the device"""
    NC.check_opt_state_compat(ops, DEV, golden_dir, variant="la")
    NC.check_opt_state_compat(ops, DEV, golden_dir, variant="pancreas")


def test_la_five_step_trajectory(ops, golden_dir):
    """K = 5 at two fixture sizes; bounds = twice the reference's own fp32-vs-fp64 drift, see check_la_traj5"""
    for fx in ("la_traj5.npz", "la_traj5m.npz"):
        rep = []
        try:
            NC.check_la_traj5(ops, DEV, golden_dir, report=rep, fixture=fx)
        finally:
            for r in rep:
                print(fx, "step %d: |hip-ref32| %.2e  |hip-ref64| %.2e  ref drift %.2e  own plab xor %g (ref %g)" % r)


def test_la_five_step_trajectory_full_size(ops, golden_dir):
    """K = 5 at configs[1]'s size (4 x 112x112x80; la_traj5f.npz = the reference's own functions, fp32 / fp64 / fp64 forced + a 4-member
    1-ulp ensemble).  SURVEY 8d asks |dloss| <= 1e-4 over 5 steps: the REFERENCE's own fp32 run misses that against its fp64 run from
    step 2 on even at this size (forced pseudo-labels: 8.6e-8, 9.4e-6, 4.3e-4, 2.3e-4, 1.5e-3) -- the drift is the optimiser's (random
    init, lr 0.01, momentum), not the small fixtures' 4-value BatchNorm groups.  So: the 1e-4 gate where the reference itself meets it
    (steps 0-1), the ensemble bound of check_la_traj5 everywhere."""
    rep = []
    try:
        NC.check_la_traj5(ops, DEV, golden_dir, report=rep, fixture="la_traj5f.npz")
    finally:
        for r in rep:
            print("la_traj5f step %d: |hip - ref32| %.2e  |hip - ref64f| %.2e  (reference ensemble median %.2e)  own pseudo-label voxels differing %d (reference 32 vs 64: %d)" % r)
    g = np.load(os.path.join(golden_dir, "la_traj5f.npz"))
    ref_own = g["drift_ens"].max(axis=0)
    for it, d32, d64, dr, pl, plr in rep:
        if ref_own[it] <= 5e-5:
            assert d64 <= 1e-4, (it, d64)


def test_standard_regime_gradients_on_hip_activation_pattern(ops):
    """every gradient tensor of the LA and the InstanceNorm V-Net to 1e-4 rel-L2 of the difference vs the fp64 oracle linearised
    on the activation pattern the HIP forward took (no smooth-regime crutch): small and a mid-size volume"""
    print("la", NC.check_vnet_pattern_grads(ops, DEV, "la", (32, 32, 16)))
    print("pancreas", NC.check_vnet_pattern_grads(ops, DEV, "pancreas", (32, 32, 32)))
    print("la 64x48x32", NC.check_vnet_pattern_grads(ops, DEV, "la", (64, 48, 32), seed=21))


def test_la_loop_body_as_the_reference_writes_it(ops, golden_dir):
    """the import-swap claim of INTEGRATION.md: unfused calls, dense masks, torch.optim.SGD, update_ema_variables"""
    NC.check_la_unfused_loop(ops, DEV, golden_dir, steps=3)


def test_la_full_size_self_train_step_vs_oracle(ops):
    """configs[1]: batch 4 (labeled_bs 2), 112x112x80, one whole step against the fp32 oracle run beside it"""
    rep = {}
    NC.check_la_step_full(ops, DEV, report=rep)
    print("full-size step:", rep)


def test_pancreas_full_size_self_train_step_vs_oracle(ops):
    """configs[4] per rank: 4 streams x 1, 96^3, IN-V-Net, one whole step + the first Adam update against the fp32 oracle beside it"""
    rep = {}
    NC.check_pancreas_step_full(ops, DEV, report=rep)
    print("pancreas full-size step:", rep)


def test_la_full_size_gradients_on_hip_activation_pattern(ops):
    """configs[1] volume size (112x112x80): every gradient tensor of the V-Net to 1e-4 rel-L2 of the difference vs the fp64 oracle
    linearised on the activation pattern of the HIP forward"""
    print("la 112x112x80", NC.check_vnet_pattern_grads(ops, DEV, "la", (112, 112, 80), seed=31, N=1))


def test_network_parity_with_bf16_pipe_conv_forced_everywhere(ops, golden_dir):
    """csrc/conv3b.hip (fp32 numerics from three-piece bf16 operands) serves the 32- and 64-channel levels of full-size volumes by
    default; here it is forced onto EVERY eligible layer of the small fixtures (conv3_b6 = 2) and the strictest network-level
    checks are repeated: golden tiny net, every gradient tensor to 1e-4 on the HIP activation pattern (BatchNorm and
    InstanceNorm V-Net), the five-step trajectory."""
    ops.set_option("conv3_b6", 2)
    ops.set_option("wgrad_b6", 2)      # csrc/conv3bw.hip: the weight gradients too
    try:
        NC.check_vnet_golden_tiny(ops, DEV, golden_dir)
        print("la", NC.check_vnet_pattern_grads(ops, DEV, "la", (32, 32, 16)))
        print("pancreas", NC.check_vnet_pattern_grads(ops, DEV, "pancreas", (32, 32, 32)))
        print("la 64x48x32", NC.check_vnet_pattern_grads(ops, DEV, "la", (64, 48, 32), seed=21))
        NC.check_la_traj5(ops, DEV, golden_dir, fixture="la_traj5m.npz")
    finally:
        ops.set_option("conv3_b6")
        ops.set_option("wgrad_b6")


def test_fp16_backward_planes_hold_over_a_long_run(ops):
    """fp16 planes for dgrad / weight-gradient operands vs three bf16 planes, on identical forward bits, probed along 1000
    training steps (rel-L2 of the difference per parameter tensor)"""
    rep = []
    try:
        NC.check_fp16_backward_long_run(ops, DEV, report=rep)
    finally:
        for it, loss, w, med in rep:
            print("fp16-bwd long run step %4d: loss %.5f  worst tensor %.2e  median %.2e" % (it, loss, w, med))


def test_up_recompute_forced_everywhere(ops):
    """round 6 (built, measured, off by default: DESIGN.md section 5): every transposed conv + norm the shape allows on the recomputing pair
    bcp_up_fwd_norm / bcp_up_norm_bwd -- gradients within 1e-4 of the fp64 oracle on the HIP activation pattern (BatchNorm and InstanceNorm
    V-Nets), and replays == the eager path with it on"""
    from bcp_amd.networks.VNet import VNet as _VN
    ops.set_option("up_recompute", 1)
    _VN.UP_RECOMPUTE_GRAD = True      # (the product takes the recomputing pair only in forwards without a backward pass: the teacher's)
    try:
        NC.check_vnet_pattern_grads(ops, DEV, "la", (32, 32, 16))
        NC.check_vnet_pattern_grads(ops, DEV, "pancreas", (32, 32, 32))
        NC.check_launch_plans(ops, DEV, steps=3, cases=(("la", True), ("pancreas", True)))
    finally:
        ops.set_option("up_recompute")
        _VN.UP_RECOMPUTE_GRAD = False


def test_partial_weight_packs(ops):
    """round 6: only the observed sections of the weight packs are written in front of replays; an eager pass behind one repacks everything"""
    NC.check_partial_packs(ops, DEV)


def test_recorded_launch_plans_equal_eager_path(ops):
    """bcp_amd/plan.py: replayed passes == the eager Python path, bit for bit (LA grouped / unfused, pancreas, ACDC; live dropout)"""
    NC.check_launch_plans(ops, DEV)


def test_graph_replays_equal_eager_path(ops):
    """plan.GRAPHS: hipGraphLaunch per network pass == the eager path, bit for bit -- forward passes (1, the default) and forward + backward
    (2), with the teacher on the student's stream and on its own side stream"""
    NC.check_launch_plans(ops, DEV, steps=4, cases=(("la", True), ("pancreas", True), ("acdc", True)), graphs=1, overlap=False)
    NC.check_launch_plans(ops, DEV, steps=4, cases=(("la", True), ("acdc", True)), graphs=2, overlap=False)      # + the backward pass
    NC.check_launch_plans(ops, DEV, steps=4, cases=(("la", True), ("acdc", True)), graphs=1, overlap=True)
    NC.check_launch_plans(ops, DEV, steps=4, cases=(("acdc", True),), graphs=2, overlap=True)


@pytest.mark.gpu
def test_replayed_passes_beside_the_teacher_stream_equal_eager_path(ops):
    """per-launch replays from C (plan.GRAPHS = 0), teacher forward on its side stream under the student's -- bit for bit the eager path
    (serial reference), 20 times over with a LOAD GENERATOR on a third stream.  Round 4's build failed this on the driver's box; under the
    load generator it failed in 21-35 % of the runs on every box (k_bilinear2x_fwd)."""
    load = NC.LoadGenerator(DEV)
    try:
        for _ in range(20):
            NC.check_launch_plans(ops, DEV, steps=4, cases=(("acdc", True),), graphs=0, real_stream=True, load=load)
        for _ in range(4):
            NC.check_launch_plans(ops, DEV, steps=4, cases=(("la", True), ("pancreas", True)), graphs=0, real_stream=True, load=load)
    finally:
        load.finish()


@pytest.mark.gpu
def test_product_configuration_under_load_equals_eager_path(ops):
    """what bench.py times and the scripts run: the module defaults (forward passes as HIP graphs, backward as per-launch replays, teacher on
    its side stream, weight gradients on theirs) on a real stream, 20 times over beside the load generator; then forward + backward graphs"""
    from bcp_amd import plan
    assert plan.GRAPHS == 1 and plan.ENABLED and plan.C_REPLAY
    load = NC.LoadGenerator(DEV)
    try:
        for _ in range(20):
            NC.check_launch_plans(ops, DEV, steps=4, cases=(("acdc", True),), graphs=None, real_stream=True, load=load)
        for _ in range(4):
            NC.check_launch_plans(ops, DEV, steps=4, cases=(("la", True), ("pancreas", True)), graphs=None, real_stream=True, load=load)
        for _ in range(6):
            NC.check_launch_plans(ops, DEV, steps=4, cases=(("acdc", True),), graphs=2, real_stream=True, load=load)
        # ... and exactly as bench.py and the training scripts set the networks up (round 5, volatile_io: no copies in and out of the passes)
        for _ in range(6):
            NC.check_launch_plans(ops, DEV, steps=4, cases=(("acdc", True), ("la", True), ("pancreas", True)), graphs=None, real_stream=True, load=load,
                                  volatile=True)
        NC.check_launch_plans(ops, DEV, steps=4, cases=(("la", False),), graphs=None, real_stream=True, load=load, volatile=True)      # (the unfused loop: busy plans)
    finally:
        load.finish()


@pytest.mark.gpu
def test_head_fused_with_last_norm_equals_separate_apply(ops):
    """VNet.fuse_head: block_nine's norm + ReLU + Dropout3d applied inside the 1x1x1 head (its activation never stored)"""
    NC.check_fused_head(ops, DEV)
