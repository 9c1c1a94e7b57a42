"""k_c3q / k_c3p (LDS-DMA pipelines) against their register-staged twins on the GPU: max |diff| per shape and slab mode."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bcp_amd.hip_ops import Ops
from bcp_amd import _lib
ops = Ops(_lib.Binding(sys.argv[1])) if len(sys.argv) > 1 else Ops.product()
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
def R(*s): return torch.from_numpy(rng.standard_normal(s).astype(np.float32))
for flat in (1, 4, 3, 0):
    ops.set_option("conv3_b6_flat", flat)
    for (N, Cin, Cout, sp) in ((2, 64, 64, (4, 8, 12)), (1, 32, 128, (6, 9, 5)), (1, 64, 64, (7, 7, 5)), (2, 128, 128, (14, 14, 10)), (2, 256, 256, (7, 7, 5)), (1, 16, 64, (5, 5, 5))):
        x = R(N, *sp, Cin).to(dev)
        w = (R(Cout, Cin, 3, 3, 3) * 0.1).to(dev)
        wf, _ = ops.conv3_pack(w, 3)
        res = {}
        for pipe in (1, 0):
            ops.set_option("conv3_b6_pipe", pipe)
            ys = []
            for rep in range(3):
                ys.append(ops.conv3_fwd(x, wf, None, Cout, 3).clone())
            torch.cuda.synchronize()
            res[pipe] = ys
        d = [float((res[1][r] - res[0][0]).abs().max()) for r in range(3)]
        d0 = float((res[0][1] - res[0][0]).abs().max())
        bad = (res[1][0] != res[0][0])
        nb = int(bad.sum())
        where = ""
        if nb:
            idx = bad.nonzero()
            where = f" first bad {idx[0].tolist()} last bad {idx[-1].tolist()} bad channels {sorted(set(idx[:, -1].tolist()))[:8]}... bad voxels {len(set(map(tuple, idx[:, :-1].tolist())))}"
        print(f"flat={flat} N={N} {Cin}->{Cout} {sp}: pipe vs nopipe max|diff| {d} (nopipe repeat {d0}) mismatches {nb}/{bad.numel()}{where}")
ops.set_option("conv3_b6_flat"); ops.set_option("conv3_b6_pipe")
