"""VERDICT r03 item 4, the gate BEFORE any kernel work: is a TWO-plane fp16 split (a0 = fp16(a), a1 = fp16(a - a0);
a*b ~ a0*b0 + a0*b1 + a1*b0, fp32 accumulation: three v_mfma_f32_16x16x32_f16 per K block instead of six bf16 ones) fp32-equivalent?
Pure numpy (fp16 x fp16 products are exact in fp32, so the arithmetic of the matrix core is reproduced up to its summation order):
dot products of conv length K = 27 * Cin against fp64, next to (a) plain fp32 accumulation = the fp32-MFMA kernels' class and
(b) the three-plane bf16 split the product uses.  Error metric = max |err| / max |exact| over the outputs (check_conv3_b6's).
Run: python tools/probe/f16split_numeric.py"""
import numpy as np


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split_bf16(x):
    p0 = bf16(x); r = x - p0; p1 = bf16(r); p2 = bf16(r - p1)
    return p0, p1, p2


def split_f16(x, scale=1.0):
    xs = (x * np.float32(scale)).astype(np.float32)
    h0 = xs.astype(np.float16).astype(np.float32)
    h1 = (xs - h0).astype(np.float16).astype(np.float32)
    return h0, h1


def dot32(a, b):        # fp32 products, fp32 running sum in blocks of 32 (a matrix-core-like order), a: [M,K], b: [K]
    acc = np.zeros(a.shape[0], np.float32)
    for k0 in range(0, a.shape[1], 32):
        acc = (acc + (a[:, k0:k0 + 32] * b[k0:k0 + 32]).astype(np.float32).sum(1, dtype=np.float32)).astype(np.float32)
    return acc


def run(name, a, b, sa=1.0, sb=1.0):
    exact = a.astype(np.float64) @ b.astype(np.float64)
    den = np.abs(exact).max()
    e32 = np.abs(dot32(a, b) - exact).max() / den
    a0, a1, a2 = split_bf16(a); b0, b1, b2 = split_bf16(b)
    y = dot32(a2, b0) + dot32(a1, b1) + dot32(a0, b2) + dot32(a1, b0) + dot32(a0, b1) + dot32(a0, b0)
    eb = np.abs(y.astype(np.float64) - exact).max() / den
    h0, h1 = split_f16(a, sa); g0, g1 = split_f16(b, sb)
    y = (dot32(h1, g0) + dot32(h0, g1) + dot32(h0, g0)).astype(np.float64) / (sa * sb)
    ef = np.abs(y - exact).max() / den
    ovf = bool(np.isinf(h0).any() or np.isinf(g0).any())
    print(f"{name:46s} fp32 {e32:.2e}  bf16x3 {eb:.2e} ({eb / e32:4.1f}x)  f16x2 {ef:.2e} ({ef / e32:6.1f}x){'  OVERFLOW' if ovf else ''}")
    return e32, eb, ef


def main():
    rng = np.random.default_rng(0)
    M = 2048
    for cin in (16, 64, 256):
        K = 27 * cin
        relu = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)                # post-norm + ReLU activations
        w = (rng.standard_normal(K) * np.sqrt(2.0 / K)).astype(np.float32)                   # He-initialised weights (|w| ~ 0.07 .. 0.017)
        run(f"Cin {cin}: relu(N(0,1)) x He weights", relu, w)
        run(f"Cin {cin}: same, weights pre-scaled 2^8", relu, w, 1.0, 256.0)
        run(f"Cin {cin}: same, both pre-scaled (2^4, 2^8)", relu, w, 16.0, 256.0)
    K = 27 * 64
    w = (rng.standard_normal(K) * np.sqrt(2.0 / K)).astype(np.float32)
    for s in (1e-4, 1e-2, 1.0, 1e2, 1e3):
        a = (np.maximum(rng.standard_normal((M, K)), 0) * s).astype(np.float32)
        run(f"activations uniformly ~{s:g}", a, w, 1.0, 256.0)
    a = (np.maximum(rng.standard_normal((M, K)), 0) * 10.0 ** rng.uniform(-4, 3, (M, K))).astype(np.float32)
    run("activations spanning 1e-4 .. 1e3 per element", a, w, 1.0, 256.0)
    run("  ... with the activations pre-scaled 2^5", a, w, 32.0, 256.0)
    dy = (rng.standard_normal((M, K)) * 1e-6).astype(np.float32)                              # backward operands (dY ~ 1e-6 at batch 4)
    run("dY ~ 1e-6 (backward), no pre-scale", dy, w, 1.0, 256.0)
    run("dY ~ 1e-6, pre-scaled 2^20", dy, w, 2.0 ** 20, 256.0)


if __name__ == "__main__":
    main()
