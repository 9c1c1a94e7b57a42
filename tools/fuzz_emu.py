"""Randomised shape fuzzing of the C-ABI kernels on the host simulator (test infrastructure; not part of the pytest suites).
Draws ragged shapes / channel counts the fixed cases do not hit and runs the same checks against torch CPU / the oracle:
   python tools/fuzz_emu.py [seconds] [seed]
Prints every failing case; exit code = number of failures."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import kernel_checks as K  # noqa: E402
import bcp_oracle as O  # noqa: E402
from bcp_amd import _lib  # noqa: E402
from bcp_amd.hip_ops import Ops  # noqa: E402
from bcp_amd.utils import BCP_utils as BU  # noqa: E402
import bcp_amd.hip_ops as H  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ops = Ops(_lib.Binding(os.environ.get("BCP_EMU_LIB") or os.path.join(ROOT, "tests", "_emu", "libbcp_emu.so")), allow_cpu=True)   # BCP_EMU_LIB: the ASan build
BU.set_test_ops(ops)
dev = torch.device("cpu")
fails, runs = [], 0
t0 = time.time()


def attempt(tag, fn):
    global runs
    runs += 1
    try:
        fn()
    except _lib.BcpError as e:            # a shape the ABI rejects loudly is not a failure
        print("rejected", tag, str(e)[:120], flush=True)
    except AssertionError as e:
        fails.append((tag, str(e)[:200]))
        print("FAIL", tag, str(e)[:200], flush=True)


while time.time() - t0 < budget:
    kind = rng.integers(0, 9)
    if kind == 0:     # 3x3x3 / 3x3 conv: fwd, dgrad, wgrad
        KD = 3 if rng.random() < 0.7 else 1
        deep = rng.random() < 0.15        # deep levels: many channels (streaming kernel, split-K, wgrad deep reduce), tiny extents
        cin, cout = int(rng.choice([128, 144, 256] if deep else [4, 8, 16, 20, 32, 48, 64])), int(rng.choice([128, 256] if deep else [4, 8, 16, 20, 32, 64]))
        if deep:
            sp = (int(rng.integers(1, 5)), int(rng.integers(1, 8)), int(rng.integers(1, 8))) if KD == 3 else (1, int(rng.integers(1, 12)), int(rng.integers(1, 12)))
        else:
            sp = (int(rng.integers(1, 9)), int(rng.integers(1, 12)), int(rng.integers(1, 21))) if KD == 3 else (1, int(rng.integers(1, 24)), int(rng.integers(1, 40)))
        case = (int(rng.integers(1, 4)), cin, cout, sp, KD)
        # half of the cases with the bf16-pipe kernels forced onto every eligible shape (the product only takes them from size
        # thresholds these extents never reach), in a random variant: persistent grid of 1-3 workgroups, direct / staged weights,
        # 2-D 32-channel slabs on either tile
        forced = {}
        if rng.random() < 0.5 and cin % 4 == 0:
            forced = {"conv3_b6": 3, "wgrad_b6": 2, "conv3_p": int(rng.integers(0, 4)), "conv3_b6_cfg2d": int(rng.choice([0, 2])),
                      "conv3_b6_direct": int(rng.choice([0, 1, 2])), "conv3_b6_flat": int(rng.choice([0, 1]))}
        for k_, v_ in forced.items():
            ops.set_option(k_, v_)
        try:
            attempt(f"conv3 {case} {forced}", lambda: K.check_conv3(ops, dev, cases=[case]))
        finally:
            for k_ in forced:
                ops.set_option(k_)
    elif kind == 1:   # largest connected component vs the oracle
        D, Hh, W = int(rng.integers(1, 12)), int(rng.integers(1, 40)), int(rng.integers(1, 40))
        p = float(rng.choice([0.1, 0.3, 0.5, 0.7]))
        seg = torch.from_numpy((rng.random((int(rng.integers(1, 3)), D, Hh, W)) < p).astype(np.uint8))
        conn = int(rng.integers(1, 4))
        big = rng.random() < 0.5
        def cc():
            if big:
                ops.set_option("cc_tile", 2)
            try:
                got = ops.cc_largest(seg, 1, conn).float()
            finally:
                ops.set_option("cc_tile")
            assert torch.equal(got, O.largest_cc(seg.long(), None if conn == 3 else conn)), "cc mismatch"
        attempt(f"cc {tuple(seg.shape)} p={p} conn={conn} big={big}", cc)
    elif kind == 2:   # norm fwd / bwd, random groups and channels
        C = int(rng.choice([16, 32, 64, 128]))
        G = int(rng.choice([1, 2, 3]))
        rows = int(rng.integers(2, 400))
        x = K.R(rng, G * rows, C)
        def norm():
            y = x.clone().requires_grad_(True)
            gam, bet = K.R(rng, C) * 0.5 + 1.0, K.R(rng, C) * 0.1
            ref = torch.cat([torch.relu(torch.nn.functional.batch_norm(y[g * rows:(g + 1) * rows], None, None, gam, bet, True, 0.1, 1e-5)) for g in range(G)])
            da = K.R(rng, *ref.shape)
            ref.backward(da)
            xv = x.view(G, rows, 1, 1, C) if False else x.view(G * rows, 1, 1, 1, C)
            a, st = ops.norm_fwd(x.view(G, rows, 1, 1, C).reshape(G, rows, 1, 1, C), G, gam, bet, torch.zeros(C), torch.ones(C), H.ACT_RELU)
            K.close(a.reshape(-1, C), ref, rtol=2e-4, msg="norm fwd")
            dg, db = torch.zeros(C), torch.zeros(C)
            dy = ops.norm_bwd(x.view(G, rows, 1, 1, C), da.view(G, rows, 1, 1, C), G, st, H.ACT_RELU, dg, db, False)
            if rows > 2:
                K.close(dy.reshape(-1, C), y.grad, rtol=2e-3, msg="norm bwd")
        attempt(f"norm C={C} G={G} rows={rows}", norm)
    elif kind == 3:   # copy-paste mix with random boxes (incl. degenerate ones)
        sp = (int(rng.integers(1, 10)), int(rng.integers(1, 20)), int(rng.integers(1, 20)))
        a, b = K.R(rng, 2, *sp, 1), K.R(rng, 2, *sp, 1)
        o = [int(rng.integers(0, s)) for s in sp]
        e = [int(rng.integers(0, s - oo + 1)) for s, oo in zip(sp, o)]
        def mix():
            got = ops.mix_box(a, b, tuple(o) + tuple(e))
            m = torch.ones(sp)
            m[o[0]:o[0] + e[0], o[1]:o[1] + e[1], o[2]:o[2] + e[2]] = 0
            ref = a * m.view(1, *sp, 1) + b * (1 - m.view(1, *sp, 1))
            assert torch.equal(got, ref), "mix mismatch"
        attempt(f"mix {sp} box={o + e}", mix)
    elif kind == 5:   # k2s2 down / up convs and the 1x1 conv: fwd, dgrad, wgrad
        cin, cout = int(rng.choice([16, 32, 64, 128])), int(rng.choice([16, 32, 64, 128, 256]))
        sp = tuple(int(2 * rng.integers(1, 8)) for _ in range(3))
        case = (int(rng.integers(1, 3)), cin, cout, sp)
        pw = (int(rng.integers(1, 4)), int(rng.choice([32, 64, 128, 256])), int(rng.choice([16, 32, 64, 128])), (int(rng.integers(1, 12)), int(rng.integers(1, 12))))
        chunks = rng.random() < 0.3
        def k2():
            if chunks:
                ops.set_option("tn_groups", 1)
            try:
                K.check_k2(ops, dev, cases=[case], pw_cases=[pw])
            finally:
                ops.set_option("tn_groups")
        attempt(f"k2 {case} pw {pw} one-group={chunks}", k2)
    elif kind == 6:   # MaxPool2d(2) / bilinear x2 (align_corners) and their backward passes vs torch
        import torch.nn.functional as F
        N, C, Hh, W = int(rng.integers(1, 4)), int(rng.choice([16, 32, 64])), int(2 * rng.integers(1, 12)), int(2 * rng.integers(1, 12))
        def pool():
            x = K.R(rng, N, C, Hh, W).requires_grad_(True)
            yr = F.max_pool2d(x, 2)
            dy = K.R(rng, *yr.shape)
            yr.backward(dy)
            xcl = K.to_cl(x.detach())
            assert torch.equal(K.from_cl(ops.maxpool2d_fwd(xcl), True), yr.detach()), "maxpool fwd"
            assert torch.equal(K.from_cl(ops.maxpool2d_bwd(xcl, K.to_cl(dy), torch.empty_like(xcl)), True), x.grad), "maxpool bwd"
            xs = K.R(rng, N, C, Hh // 2 + 1, W // 2 + 1).requires_grad_(True)
            up = F.interpolate(xs, scale_factor=2, mode="bilinear", align_corners=True)
            du = K.R(rng, *up.shape)
            up.backward(du)
            buf = torch.zeros(N, 1, up.shape[2], up.shape[3], C)
            ops.bilinear2x_fwd(K.to_cl(xs.detach()), buf, 0)
            K.close(K.from_cl(buf, True), up, msg="bilinear fwd")
            K.close(K.from_cl(ops.bilinear2x_bwd(K.to_cl(du), 0, C), True), xs.grad, msg="bilinear bwd")
        attempt(f"pool2d N={N} C={C} {Hh}x{W}", pool)
    elif kind == 7:   # pseudo-labels vs torch (ties excluded: random logits)
        n = int(rng.integers(1, 5000))
        lo2, lo4 = K.R(rng, 1, 1, 1, n, 2), K.R(rng, 1, 1, 1, n, 4)
        def pl():
            assert torch.equal(ops.plabel_bin(lo2, 0.5).long(), (torch.softmax(lo2, -1)[..., 1] >= 0.5).long()), "plabel_bin"
            assert torch.equal(ops.plabel_argmax4(lo4).long(), torch.softmax(lo4, -1).argmax(-1)), "plabel_argmax4"
        attempt(f"plabel n={n}", pl)
    elif kind == 8:   # masked Dice + CE (both flavours), value and d/dlogits vs the oracle, random boxes incl. empty / full ones
        from bcp_amd import train_step
        three_d = rng.random() < 0.5
        N = int(rng.integers(1, 4))
        sp = (int(rng.integers(1, 7)), int(rng.integers(1, 10)), int(4 * rng.integers(1, 6))) if three_d else (int(rng.integers(1, 14)), int(4 * rng.integers(1, 8)))
        Cc = 2 if three_d else 4
        o = [int(rng.integers(0, d)) for d in sp]
        e = [int(rng.integers(0, d - oo + 1)) for d, oo in zip(sp, o)]
        unlab = bool(rng.random() < 0.5)
        def ml():
            logits = K.R(rng, N, Cc, *sp)
            a, b = torch.from_numpy(rng.integers(0, Cc, (N,) + sp)), torch.from_numpy(rng.integers(0, Cc, (N,) + sp))
            box = tuple(o) + tuple(e)
            dense = O.box_to_mask(box, sp, N)[1].to(torch.float32)
            lo_ref = logits.clone().requires_grad_(True)
            lo_hip = logits.clone().requires_grad_(True)
            if three_d:
                ref = O.mix_loss_la(lo_ref, a, b, dense, u_weight=0.5, unlab=unlab)
                got = BU.mix_loss(lo_hip, a, b, BU.BoxMask(box, sp, N, False, dev), u_weight=0.5, unlab=unlab)
                ref.backward(); got.backward()
                assert abs(float(got.detach()) - float(ref.detach())) < 2e-5, (float(got.detach()), float(ref.detach()))
            else:
                rd, rc = O.mix_loss_acdc(lo_ref, a, b, dense, u_weight=0.5, unlab=unlab)
                gd, gc = train_step.acdc_mix_loss(lo_hip, a, b, BU.BoxMask(box, sp, N, False, dev), u_weight=0.5, unlab=unlab)
                (rd + rc).backward(); (gd + gc).backward()
                assert abs(float(gd.detach()) - float(rd.detach())) < 2e-5 and abs(float(gc.detach()) - float(rc.detach())) < 2e-5
            K.close(lo_hip.grad, lo_ref.grad, rtol=2e-4, atol_scale=1e-4, msg="dloss/dlogits")
        attempt(f"mixloss {'la' if three_d else 'acdc'} N={N} {sp} box={o + e} unlab={unlab}", ml)
    else:             # ACDC augment gather vs the oracle's scipy restatement
        Hh, W = int(rng.integers(2, 70)), int(rng.integers(2, 70))
        out_hw = (int(rng.integers(2, 80)), int(rng.integers(2, 80)))
        img = rng.random((Hh, W)).astype(np.float32)
        mode = int(rng.integers(0, 3))
        k, axis, ang = int(rng.integers(0, 4)), int(rng.integers(0, 2)), int(rng.integers(-20, 20))
        def aug():
            if mode == 0:
                ref = O._nearest_zoom(img, out_hw)
                got = ops.acdc_augment(torch.from_numpy(img), out_hw, 0)
            elif mode == 1:
                ref = O._nearest_zoom(np.flip(np.rot90(img, k), axis=axis).copy(), out_hw)
                got = ops.acdc_augment(torch.from_numpy(img), out_hw, 1, k, axis)
            else:
                ref = O._nearest_zoom(O._nearest_rotate(img, ang), out_hw)
                got = ops.acdc_augment(torch.from_numpy(img), out_hw, 2, 0, 0, O.rotate_affine(ang, img.shape))
            assert np.array_equal(got.numpy(), ref), "augment mismatch"
        attempt(f"augment {Hh}x{W}->{out_hw} mode={mode} k={k} axis={axis} ang={ang}", aug)
print(f"{runs} cases, {len(fails)} failures in {time.time() - t0:.0f} s")
sys.exit(len(fails))
