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
             (2, 16, 16, (1, 64, 48), 1), (2, 128, 128, (1, 32, 32), 1))
    K.check_conv3(gpu_ops, torch.device("cuda:0"), cases=cases)
