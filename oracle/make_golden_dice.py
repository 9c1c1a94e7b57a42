"""oracle/make_golden_dice.py -- golden vectors for the `DiceLoss` CLASS seam, captured by IMPORTING the reference's
utils/losses.py (build container only; reads /root/reference/code, writes tests/golden/diceloss_class.npz: data only).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_dice.py

Cases (utils/losses.py:113-134): probabilities in + dense [N,1,H,W] mask (how ACDC_BCP_train.py:175-176 calls it), the
complementary mask, no mask (`_dice_loss`, the same smooth 1e-10, utils/losses.py:94), per-class `weight`, `softmax=True` on logits, a class absent from target and prediction (the smooth term decides); each with the gradient w.r.t.
the tensor that was passed in."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/code"
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True

import importlib.util  # noqa: E402

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_losses", os.path.join(REF, "utils", "losses.py"))
ref_losses = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_losses)


def main():
    rng = np.random.default_rng(77)
    N, C, H, W = 3, 4, 24, 40
    logits = torch.from_numpy(rng.standard_normal((N, C, H, W), dtype=np.float32) * 2)
    target = torch.from_numpy(rng.integers(0, C, (N, 1, H, W)))
    mask = torch.ones(N, 1, H, W, dtype=torch.int64)
    mask[:, :, 5:16, 7:25] = 0
    weight = [0.5, 1.0, 2.0, 0.25]
    dl = ref_losses.DiceLoss(C)
    out = {"logits": logits.numpy(), "target": target.numpy(), "mask": mask.numpy(), "weight": np.asarray(weight, np.float32)}

    def run(name, fn, leaf):
        leaf = leaf.clone().requires_grad_(True)
        v = fn(leaf)
        v.backward()
        out[name] = np.float64(v.item())
        out["g_" + name] = leaf.grad.numpy().copy()

    probs = F.softmax(logits, dim=1)
    run("masked", lambda p: dl(p, target, mask), probs)
    run("masked_c", lambda p: dl(p, target, 1 - mask), probs)
    run("nomask", lambda p: dl(p, target), probs)
    run("weighted", lambda p: dl(p, target, mask, weight=weight), probs)
    run("softmax", lambda x: dl(x, target, mask, softmax=True), logits)
    # a class that is ABSENT from the target and all but absent from the prediction (sum p^2 ~ 1e-8): the only regime in which the smooth
    # term decides the value -- 1e-10 gives that class a Dice near 0.01, a 1e-5 would give 0.999 (the unmasked branch)
    target_abs = torch.where(target == 3, torch.zeros_like(target), target)
    logits_abs = logits.clone()
    logits_abs[:, 3] -= 13.0
    probs_abs = F.softmax(logits_abs, dim=1)
    out["target_abs"], out["logits_abs"] = target_abs.numpy(), logits_abs.numpy()
    run("absent", lambda p: dl(p, target_abs), probs_abs)
    run("absent_masked", lambda p: dl(p, target_abs, mask), probs_abs)
    np.savez_compressed(os.path.join(OUT, "diceloss_class.npz"), **out)
    print({k: float(v) for k, v in out.items() if np.ndim(v) == 0})


if __name__ == "__main__":
    main()
