"""Synthetic inputs of the benchmark / smoke runs: no dataset ships with the build.
LA: image ~N(0,1) [B,1,112,112,80] (LA volumes are z-scored), label {0,1} = one ellipsoid (~8 % foreground)
plus a few small satellites.  ACDC: image ~U[0,1) [B,1,256,256], label {0..3} = three nested discs."""
import numpy as np
import torch


def la_batch(batch, shape=(112, 112, 80), seed=1337):
    rng = np.random.default_rng(seed)
    X, Y, Z = shape
    img = rng.standard_normal((batch, 1, X, Y, Z), dtype=np.float32)
    gx, gy, gz = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij")
    lab = np.zeros((batch, X, Y, Z), dtype=np.int64)
    for b in range(batch):
        c = np.array([X, Y, Z]) * (0.5 + 0.1 * (rng.random(3) - 0.5))
        r = np.array([X, Y, Z]) * (0.27 + 0.04 * rng.random(3))
        m = ((gx - c[0]) / r[0]) ** 2 + ((gy - c[1]) / r[1]) ** 2 + ((gz - c[2]) / r[2]) ** 2 <= 1.0
        for _ in range(3):
            sc = rng.random(3) * np.array([X, Y, Z])
            sr = 2.0 + 2.0 * rng.random()
            m |= ((gx - sc[0]) ** 2 + (gy - sc[1]) ** 2 + (gz - sc[2]) ** 2) <= sr * sr
        lab[b] = m
        img[b, 0] += 1.5 * m
    return torch.from_numpy(img), torch.from_numpy(lab)


def acdc_batch(batch, shape=(256, 256), seed=1337):
    rng = np.random.default_rng(seed)
    H, W = shape
    img = rng.random((batch, 1, H, W), dtype=np.float32)
    gy, gx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    lab = np.zeros((batch, H, W), dtype=np.int64)
    for b in range(batch):
        cy, cx = H * (0.4 + 0.2 * rng.random()), W * (0.4 + 0.2 * rng.random())
        d2 = (gy - cy) ** 2 + (gx - cx) ** 2
        base = min(H, W)
        for c, rr in ((1, 0.30), (2, 0.20), (3, 0.10)):
            lab[b][d2 <= (rr * base) ** 2] = c
        img[b, 0] = 0.6 * img[b, 0] + 0.1 * lab[b]
    return torch.from_numpy(img), torch.from_numpy(lab)
