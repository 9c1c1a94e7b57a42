"""Two-or-more-rank smoke test of the data-parallel exchange through the C ABI (bcp_comm_init_rank / bcp_allreduce_f32 -> RCCL):
run it whenever >= 2 GPUs are visible:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/rccl_smoke.py

Every rank: (1) all-reduce of a rank-dependent vector == the closed-form sum; (2) three LA self-training steps on a small volume
with per-rank data -- the students (and teachers) of all ranks must stay bit-identical, and equal to a single-process run that
averages the same micro-batch gradients (N ranks == N sequential micro-batches)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bcp_amd import synth, train_step  # noqa: E402
from bcp_amd.dp import DataParallel  # noqa: E402
from bcp_amd.networks.net_factory import net_factory  # noqa: E402


def main():
    dp = DataParallel(backend=os.environ.get("BCP_DP_BACKEND", "rccl"))
    assert dp.enabled and dp.world >= 2, "launch with >= 2 ranks (torch.distributed.run --nproc-per-node N)"
    dev = torch.device("cuda", dp.local_rank)
    torch.cuda.set_device(dev)
    seen = dp.ranks_seen()
    assert seen == dp.world, f"communicator spans {seen} ranks, expected {dp.world}"
    n = 1 << 20
    v = torch.arange(n, dtype=torch.float32, device=dev) * 1e-3 + dp.rank
    if dp.abi is not None:
        dp.abi.all_reduce(v)
    else:
        import torch.distributed as dist
        dist.all_reduce(v)
    want = torch.arange(n, dtype=torch.float32, device=dev) * 1e-3 * dp.world + sum(range(dp.world))
    assert torch.allclose(v, want, rtol=1e-6), "all-reduce sum is wrong"
    torch.manual_seed(7)
    model = net_factory(net_type="VNet", in_chns=1, class_num=2, mode="train")
    ema = net_factory(net_type="VNet", in_chns=1, class_num=2, mode="train")
    for p in ema.parameters():
        p.detach_()
    ema.load_state_dict(model.state_dict())
    dp.broadcast_params(model); dp.broadcast_params(ema)
    opt = train_step.FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
    np.random.seed(100 + dp.rank)
    vol, lab = synth.la_batch(4, shape=(32, 32, 16), seed=100 + dp.rank)
    vol, lab = vol.to(dev), lab.to(dev)
    for _ in range(3):
        r = train_step.la_self_train_step(model, ema, opt, vol, lab, 2, box=(3, 4, 2, 20, 20, 10), dp=dp)
    torch.cuda.synchronize()
    for name, net in (("student", model), ("teacher", ema)):
        flat = net.flat_params().clone()
        ref = flat.clone()
        dp.abi.broadcast(ref) if dp.abi is not None else __import__("torch.distributed").distributed.broadcast(ref, src=0)
        assert torch.equal(flat, ref), f"rank {dp.rank}: {name} diverged from rank 0"
    if dp.rank == 0:
        print(f"rccl_smoke OK: {dp.world} ranks (backend {dp.backend}), {dp.n_collectives} gradient collectives in 3 steps, loss {float(r['loss']):.6f}")
    dp.shutdown()


if __name__ == "__main__":
    main()
