"""-m gpu: every C-ABI kernel of libbcp_hip.so on a real MI355X against the oracle / torch-CPU fp32
(same checks as the simulator run, tests/kernel_checks.py), plus config-size cases."""
import pytest
import torch

import kernel_checks as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_ops():
    from bcp_amd.hip_ops import Ops
    assert torch.cuda.is_available(), "the gpu tests need a GPU"
    ops = Ops.product()  # raises loudly if libbcp_hip.so is missing
    return ops


@pytest.mark.parametrize("name", K.ALL_CHECKS)
def test_kernel(gpu_ops, golden_dir, name):
    fn = getattr(K, "check_" + name)
    dev = torch.device("cuda:0")
    if name in ("diceloss_class", "plabel", "cc", "mixloss", "augment", "augment_acdc", "augment_pancreas"):
        fn(gpu_ops, dev, golden_dir)
    else:
        fn(gpu_ops, dev)
    torch.cuda.synchronize()


def test_arch_is_gfx950(gpu_ops):
    import ctypes
    buf = ctypes.create_string_buffer(64)
    gpu_ops.b.call("bcp_device_arch", buf, 64)
    assert buf.value.decode().startswith("gfx950"), buf.value


def test_conv3_config_shapes(gpu_ops):
    """LA config-size layers (bigger tiles, multi-wave grids) vs torch CPU"""
    cases = ((1, 16, 16, (24, 20, 48), 3), (1, 32, 32, (16, 24, 24), 3), (1, 64, 64, (12, 12, 12), 3), (1, 256, 256, (7, 7, 5), 3),
             (1, 32, 32, (42, 44, 38), 3),   # 64K..256K voxels: the 4x8x8 forward / 4x4x8 wgrad tiles
             (2, 16, 16, (1, 64, 48), 1), (2, 128, 128, (1, 32, 32), 1))
    K.check_conv3(gpu_ops, torch.device("cuda:0"), cases=cases)


def test_full_size_properties(gpu_ops):
    """BASELINE.json's full LA size (2 x 112 x 112 x 80): size-independent properties the oracle is too slow to check"""
    ops, dev = gpu_ops, torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    N, sp, C = 2, (112, 112, 80), 16
    x1 = torch.randn(N, *sp, C, generator=g).to(dev)
    x2 = torch.randn(N, *sp, C, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, 3, generator=g) * 0.05).to(dev)
    wf, wd = ops.conv3_pack(w, 3)
    # (1) linearity of the conv (fwd and dgrad pack): conv(x1 + x2) = conv(x1) + conv(x2) to fp32 rounding
    for pack in (wf, wd):
        y12 = ops.conv3_fwd(x1 + x2, pack, None, C, 3)
        ysum = ops.conv3_fwd(x1, pack, None, C, 3) + ops.conv3_fwd(x2, pack, None, C, 3)
        assert float((y12 - ysum).abs().max()) < 2e-5 * float(ysum.abs().max())
    # (2) adjointness: <conv(x), d> = <x, dgrad(d)> and = <w, wgrad(x, d)>  (fp64 dot products on the host side of the check)
    d = torch.randn(N, *sp, C, generator=g).to(dev)
    y = ops.conv3_fwd(x1, wf, None, C, 3)
    dx = ops.conv3_fwd(d, wd, None, C, 3)
    dw = torch.zeros_like(w)
    ops.conv3_wgrad(x1, d, dw, 3)
    lhs = float((y.double() * d.double()).sum())
    r1, r2 = float((x1.double() * dx.double()).sum()), float((w.double() * dw.double()).sum())
    # fp32 rounding of ~3e7 products of O(1) terms with random signs: |error| ~ sqrt(n) * eps * sigma ~ 1e-2
    tol = 5e-2
    assert abs(lhs - r1) < tol and abs(lhs - r2) < tol, (lhs, r1, r2)
    # (3) grouped BatchNorm: every (group, channel) of the normalised tensor has mean beta and variance gamma^2
    gamma, beta = torch.rand(C, generator=g).to(dev) + 0.5, (torch.rand(C, generator=g) - 0.5).to(dev)
    a, _ = ops.norm_fwd(y, 2, gamma, beta, torch.zeros(C, device=dev), torch.ones(C, device=dev), 0)   # ACT_NONE
    for gi in range(2):
        z = a[gi].reshape(-1, C).double()
        assert float((z.mean(0) - beta.double()).abs().max()) < 1e-5
        assert float((z.var(0, unbiased=False).sqrt() - gamma.double()).abs().max()) < 1e-4
    # (4) copy-paste mix: the two complementary mixes partition the inputs; mix(a, a) = a
    img_a, img_b = torch.randn(1, *sp, 1, generator=g).to(dev), torch.randn(1, *sp, 1, generator=g).to(dev)
    box = (11, 7, 5, 74, 74, 53)
    m1, m2 = ops.mix_box(img_a, img_b, box), ops.mix_box(img_b, img_a, box)
    assert torch.equal(m1 + m2, img_a + img_b) and torch.equal(ops.mix_box(img_a, img_a, box), img_a)
    inside = int((m1 == img_b).sum()) - int((img_a == img_b).sum())
    assert inside == 74 * 74 * 53
    # (5) largest connected component is idempotent, a subset of its input, and connected (a second pass keeps everything)
    seg = (torch.rand(2, *sp, generator=g) < 0.35).to(torch.uint8).to(dev)
    cc1 = ops.cc_largest(seg, 1, 3)
    cc2 = ops.cc_largest(cc1, 1, 3)
    assert torch.equal(cc1, cc2) and bool((cc1 <= seg).all()) and 0 < int(cc1.sum()) <= int(seg.sum())
    # (5b, round 6) the teacher's tail as ONE chain from the logits (bcp_plabel_cc_largest: quads of voxels labelled from 16-byte loads,
    # the selection inside the size count, the wave-level pair exchange in the border pass) == pseudo-label launch + chain, bit for bit,
    # at the full size, for every connectivity; the optional per-tile size table gives the same map; same properties as (5)
    lg = torch.randn(2, *sp, 2, generator=g).to(dev)
    pl = ops.plabel_bin(lg, 0.5)
    for conn in (3, 2, 1):
        two = ops.cc_largest(pl, 1, conn)
        (one, onef), seg1 = ops.plabel_cc_largest(lg, 0.5, conn, want_f32=True, want_seg=True)
        assert torch.equal(seg1, pl) and torch.equal(one, two) and torch.equal(onef, two.float()), f"one chain vs two calls, connectivity {conn}"
        assert torch.equal(ops.cc_largest(one, 1, conn), one) and bool((one <= pl).all()) and 0 < int(one.sum()) <= int(pl.sum())
        for opt in ("cc_count_tile", "cc_border_dedupe", "cc_fuse_select"):
            ops.set_option(opt, 1 if opt == "cc_count_tile" else 0)
            try:
                assert torch.equal(ops.plabel_cc_largest(lg, 0.5, conn), one), f"{opt} flipped, connectivity {conn}"
            finally:
                ops.set_option(opt)
    # (6) EMA with alpha = 1 is the identity, with alpha = 0 a copy
    p, q = torch.randn(1 << 20, generator=g).to(dev), torch.randn(1 << 20, generator=g).to(dev)
    p0 = p.clone()
    ops.ema(p, q, 1.0)
    assert torch.equal(p, p0)
    ops.ema(p, q, 0.0)
    assert torch.equal(p, q)
    torch.cuda.synchronize()


def test_full_size_properties_acdc(gpu_ops):
    """BASELINE.json's ACDC size (grouped batch of 12 slices, 256 x 256, the U-Net's 16-channel level): size-independent
    properties + torch's own elementwise / pooling kernels on the same device as a second opinion"""
    import torch.nn.functional as F
    ops, dev = gpu_ops, torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    N, H, W, C = 12, 256, 256, 16
    x1 = torch.randn(N, 1, H, W, C, generator=g).to(dev)
    x2 = torch.randn(N, 1, H, W, C, generator=g).to(dev)
    d = torch.randn(N, 1, H, W, C, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.08).to(dev)
    wf, wd = ops.conv3_pack(w, 1)
    # (1) 3x3 conv: linearity of both packs, adjointness of fwd / dgrad / wgrad
    for pack in (wf, wd):
        y12 = ops.conv3_fwd(x1 + x2, pack, None, C, 1)
        ysum = ops.conv3_fwd(x1, pack, None, C, 1) + ops.conv3_fwd(x2, pack, None, C, 1)
        assert float((y12 - ysum).abs().max()) < 2e-5 * float(ysum.abs().max())
    y, dx = ops.conv3_fwd(x1, wf, None, C, 1), ops.conv3_fwd(d, wd, None, C, 1)
    dw = torch.zeros_like(w)
    ops.conv3_wgrad(x1, d, dw, 1)
    lhs = float((y.double() * d.double()).sum())
    r1, r2 = float((x1.double() * dx.double()).sum()), float((w.double() * dw.double()).sum())
    assert abs(lhs - r1) < 5e-2 and abs(lhs - r2) < 5e-2, (lhs, r1, r2)      # ~1.3e7 O(1) products: sqrt(n) * eps * sigma ~ 1e-2
    # (2) MaxPool2d(2) and its backward == torch's kernels on the same tensors (NCHW views of the NHWC buffers), bit for bit
    xn = x1[:, 0].permute(0, 3, 1, 2).requires_grad_(True)
    yp = ops.maxpool2d_fwd(x1)
    yt = F.max_pool2d(xn, 2)
    assert torch.equal(yp[:, 0].permute(0, 3, 1, 2), yt)
    dyp = torch.randn(N, 1, H // 2, W // 2, C, generator=g).to(dev)
    yt.backward(dyp[:, 0].permute(0, 3, 1, 2))
    dxp = ops.maxpool2d_bwd(x1, dyp, torch.empty_like(x1))
    assert torch.equal(dxp[:, 0].permute(0, 3, 1, 2), xn.grad)
    # (3) bilinear x2 (align_corners=True): vs torch within fp32 rounding; backward is its adjoint
    xs = torch.randn(N, 1, H // 2, W // 2, C, generator=g).to(dev)
    buf = torch.zeros(N, 1, H, W, C, device=dev)
    ops.bilinear2x_fwd(xs, buf, 0)
    ref = F.interpolate(xs[:, 0].permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True)
    assert float((buf[:, 0].permute(0, 3, 1, 2) - ref).abs().max()) < 1e-5
    dxs = ops.bilinear2x_bwd(d, 0, C)
    a1, a2 = float((buf.double() * d.double()).sum()), float((xs.double() * dxs.double()).sum())
    assert abs(a1 - a2) < 2e-2, (a1, a2)
    # (4) pseudo-labels: softmax -> first-max argmax == torch.argmax of the logits (softmax is monotone; random logits have no ties)
    lo = torch.randn(N, 1, H, W, 4, generator=g).to(dev)
    pl = ops.plabel_argmax4(lo)
    assert torch.equal(pl.long(), lo.argmax(-1))
    # (5) per-class largest connected component (8-connectivity): idempotent, a subset, never empty for a class that is present
    cc1 = ops.cc_largest(pl, 3, 2)
    cc2 = ops.cc_largest(cc1, 3, 2)
    assert torch.equal(cc1, cc2) and bool(((cc1 == pl) | (cc1 == 0)).all())
    for c in (1, 2, 3):
        per_slice = (cc1 == c).flatten(1).sum(1)
        assert bool((per_slice > 0).all()) and bool((per_slice <= (pl == c).flatten(1).sum(1)).all())
    # (5b, round 6) the same as ONE chain from the logits (bcp_plabel_cc_largest, four channels / three classes) == the two calls, bit for bit
    one, seg1 = ops.plabel_cc_largest(lo, 0.5, 2, want_seg=True)
    assert torch.equal(seg1, pl) and torch.equal(one, cc1)
    # (6) the two complementary copy-paste mixes partition their inputs
    a, b = torch.randn(N, 1, H, W, 1, generator=g).to(dev), torch.randn(N, 1, H, W, 1, generator=g).to(dev)
    box = (0, 40, 31, 1, 170, 170)
    assert torch.equal(ops.mix_box(a, b, box) + ops.mix_box(b, a, box), a + b)
    torch.cuda.synchronize()


def test_packed_fp32_hazard_canary(tmp_path):
    """DESIGN.md section 4.0 / VERDICT r05 item 9: the stand-alone reproducer of the gfx950 packed-multiply hazard (tools/probe/pkmul_mfma_repro.hip,
    built here with hipcc) beside a dense bf16-MFMA kernel.  The PLAIN form and every operand-selection form the product library still contains
    (tools/isa_scan.py watch list: low results from low halves) must stay at 0 wrong launches -- a stack or microcode update that widens the
    hazard turns this red before any training result moves.  The crossed form is reported, not asserted (a fixed stack would make it 0)."""
    import os
    import re
    import shutil
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    if not hipcc:
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "pkmul_mfma_repro")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-w", "-o", exe, os.path.join(root, "tools", "probe", "pkmul_mfma_repro.hip")], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, check=True).stdout
    print(out)
    m = re.search(r"crossed: (\d+) of (\d+) launches wrong.*plain: (\d+) of (\d+)", out)
    assert m, out
    assert int(m.group(3)) == 0, "the PLAIN packed multiply came out wrong beside 16-bit MFMAs: " + out
    forms = re.findall(r"FORM (.+?): (\d+) of (\d+)", out)
    assert len(forms) == 4, out
    for name, bad, n in forms:
        assert int(bad) == 0, f"packed form '{name}' (present in libbcp_hip.so) wrong in {bad} of {n} launches beside 16-bit MFMAs: " + out
