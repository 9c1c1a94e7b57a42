"""Driver of tools/probe/pkmov_hazard.hip (GPU): every variant of "first consumer behind a partial s_waitcnt vmcnt" is launched RING times
per round on its own stream BESIDE the library's bf16-pipe 2-D convs (LDS-DMA weight stages) on other streams and a copy / GEMM load;
counts the lanes whose consumer saw something else than the loaded value.

  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probe/libpkmov.so tools/probe/pkmov_hazard.hip
  python tools/probe/pkmov_hazard.py [rounds=20] [conv=2] [load=1]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
from bcp_amd.hip_ops import Ops

kv = dict(a.split("=") for a in sys.argv[1:])
ROUNDS, CONV, LOAD, RING = int(kv.get("rounds", 20)), int(kv.get("conv", 2)), int(kv.get("load", 1)), int(kv.get("ring", 64))
NB = kv.get("nb", "")            # stand-alone neighbour instead of the library's convs: mfma_bf16 | mfma_f16 | mfma_f32 | ldsdma | valu | lds
NB_KIND = {"mfma_bf16": 0, "mfma_f16": 1, "mfma_f32": 2, "ldsdma": 3, "valu": 4, "lds": 5}.get(NB)
NB_ITERS, NB_BLOCKS = int(kv.get("nb_iters", 400)), int(kv.get("nb_blocks", 1024))
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpkmov.so"))
lib.pk_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
lib.nb_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
ops = Ops.product(); dev = torch.device("cuda:0")
Ops.AMAX = False
ROWS, C4 = 4096 + 9, 8
x = torch.randn(ROWS * C4 * 4, device=dev)
n = (ROWS - 9) * C4
st = torch.cuda.Stream(device=dev)
ls = torch.cuda.Stream(device=dev)
la = torch.randn(32 << 20, device=dev); lb = torch.empty_like(la); lm = torch.randn(2048, 2048, device=dev); lo = torch.empty_like(lm)
convs = []
for k in range(CONV):
    items = []
    for (cc, hh) in ((16, 64), (32, 32), (64, 16), (128, 8)):
        xi = torch.randn(4, 1, hh, hh, cc, device=dev)
        w = (torch.randn(cc, cc, 3, 3, device=dev) * 0.1).contiguous()
        wf, _ = ops.conv3_pack(w, 1)
        items.append((xi, wf, torch.zeros(cc, device=dev), cc))
    convs.append((torch.cuda.Stream(device=dev), items))
ring = [torch.zeros(n * 3, device=dev) for _ in range(RING)]
nb_stream = torch.cuda.Stream(device=dev)
nb_sink, nb_src = torch.zeros(256, device=dev), torch.randn(1 << 22, device=dev)
torch.cuda.synchronize()
NAMES = {0: "v_pk_mov_b32 op_sel:[1,0] behind vmcnt(1)", 1: "v_mov_b32 behind vmcnt(1)", 2: "v_pk_mov_b32 op_sel:[1,0] behind vmcnt(0)",
         3: "v_pk_mov_b32 op_sel:[0,0] behind vmcnt(1), then v_mov_b32 of the dword", 4: "v_pk_mov_b32 behind vmcnt(1) + s_nop 7", 5: "v_pk_add_f32 op_sel:[1,0] behind vmcnt(1)",
         6: "two v_pk_mov_b32 op_sel:[1,0] with destination == second source, behind vmcnt(0)", 7: "the same behind vmcnt(1)",
         8: "v_pk_mul_f32 S0=loaded pair, S1=(2.0,1.0) op_sel:[0,1] op_sel_hi:[1,0] (second source's halves crossed)", 9: "the same, S0 VALU-written",
         10: "plain v_pk_mul_f32 on pre-swapped constants", 11: "v_pk_add_f32 with the second source's halves crossed", 12: "v_pk_mul_f32 with the FIRST source's halves crossed",
         13: "v_pk_fma_f32 with the second source's halves crossed", 14: "v_pk_mul_f32 op_sel:[0,1] (S1.hi to both halves)", 15: "v_pk_mul_f32 op_sel_hi:[1,0] (S1.lo to both halves)",
         16: "v_pk_mul_f32 with BOTH sources' halves crossed", 17: "the guilty form with S1 = (1.0, 1.0)"}
for variant in [int(v) for v in kv.get('variants', '0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17').split(',')]:
    bad_seen = bad_later = launches_bad = 0
    lanes = {}
    sentinel = 0
    for r in range(ROUNDS):
        if LOAD:
            with torch.cuda.stream(ls):
                for _ in range(8):
                    lb.copy_(la); torch.mm(lm, lm, out=lo); la[: 1 << 20].add_(1.0)
        for i in range(RING):
            for cs, items in convs:
                with torch.cuda.stream(cs):
                    xi, wf, bias, cc = items[i % len(items)]
                    ops.conv3_fwd(xi, wf, bias, cc, 1)
            if NB_KIND is not None:
                assert lib.nb_run(NB_KIND, nb_sink.data_ptr(), nb_src.data_ptr(), NB_ITERS, NB_BLOCKS, nb_stream.cuda_stream) == 0
            rc = lib.pk_run(variant, x.data_ptr(), ring[i].data_ptr(), ROWS, C4, st.cuda_stream)
            assert rc == 0
        torch.cuda.synchronize()
        for o in ring:
            v = o.view(n, 3)
            b0 = (v[:, 0] != v[:, 2]).nonzero().reshape(-1)
            if b0.numel():
                launches_bad += 1
                bad_seen += b0.numel()
                sentinel += int((v[b0, 0].view(torch.int32) == -1056969216).sum())          # 0xc0ffee00
                if launches_bad <= 2:
                    k = b0[:4]
                    print(f"   sample lanes {[int(t) % 64 for t in k]}: saw {[float(t) for t in v[k, 0]]} expected {[float(t) for t in v[k, 2]]}", flush=True)
                for t in b0.tolist():
                    q = (t % 64) // 16
                    lanes[q] = lanes.get(q, 0) + 1
            bad_later += int((v[:, 1] != v[:, 2]).sum())
    print(f"[neighbour {NB or ('library convs x%d' % CONV)}, load {LOAD}] variant {variant} [{NAMES[variant]}]: {launches_bad} of {ROUNDS * RING} launches, {bad_seen} lanes saw a wrong value ({sentinel} of them the "
          f"register's OLD contents), quarter-wave histogram {dict(sorted(lanes.items()))}; the plain read 8 wait states later was wrong {bad_later} times", flush=True)
