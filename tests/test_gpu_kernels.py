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
    if name in ("plabel", "cc", "mixloss", "augment"):
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
    # (6) EMA with alpha = 1 is the identity, with alpha = 0 a copy
    p, q = torch.randn(1 << 20, generator=g).to(dev), torch.randn(1 << 20, generator=g).to(dev)
    p0 = p.clone()
    ops.ema(p, q, 1.0)
    assert torch.equal(p, p0)
    ops.ema(p, q, 0.0)
    assert torch.equal(p, q)
    torch.cuda.synchronize()
