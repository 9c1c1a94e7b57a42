import sys, torch
sys.path.insert(0, "/root/repo")
from bcp_amd import hip_ops as H
from bcp_amd.hip_ops import Ops
ops = Ops.product(); dev = torch.device("cuda:0")
for C, sp in ((16, (112, 112, 80)), (32, (56, 56, 40))):
    y = torch.randn(2, *sp, C, device=dev); dy = torch.randn(2, *sp, C, device=dev)
    g, be = torch.ones(C, device=dev), torch.zeros(C, device=dev); rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    a = torch.empty_like(y)
    for _ in range(5):
        _, st = ops.norm_fwd(y, 2, g, be, rm, rv, H.ACT_RELU, out=a)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ops.norm_bwd(y, dy, 2, st, H.ACT_RELU, dg, db, out=a)
torch.cuda.synchronize()
