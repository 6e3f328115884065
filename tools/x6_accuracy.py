#!/usr/bin/env python3
"""Accuracy of the scorer's bf16 x 6 scheme (csrc/pmi_kernel.hip: pmi_score_x6_kernel), emulated in numpy.

An fp32 value is split by truncation into three bf16 values (x = hi + mid + lo, exact up to 2^-24 |x|); a product
x * w is taken as the six bf16 products of total order <= 2 (hh, hm, mh, hl, lh, mm), each exact in fp32, accumulated in
fp32.  Compared with an fp64 reference on a [512 x 384] x [384 x 128] layer of post-ReLU activations:
the three-term variant (hh, hm, mh) is printed too -- it is why six are needed."""
import numpy as np


def trunc_bf16(a):
    return (a.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split3(a):
    h = trunc_bf16(a)
    r1 = (a - h).astype(np.float32)
    m = trunc_bf16(r1)
    r2 = (r1 - m).astype(np.float32)
    return h, m, trunc_bf16(r2)


def mm_f32acc(a, b, step=16):
    """exact bf16 products, fp32 accumulation in k-steps of 16 (one MFMA)"""
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(0, a.shape[1], step):
        acc = (acc + (a[:, k:k + step].astype(np.float64) @ b[k:k + step].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return acc


def main():
    rng = np.random.RandomState(0)
    K, N, M = 384, 128, 512
    x = np.maximum(rng.randn(M, K).astype(np.float32) * 3, 0)
    w = (rng.randn(K, N) * 0.1).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64)
    xh, xm, xl = split3(x)
    wh, wm, wl = split3(w)
    six = mm_f32acc(xh, wl) + mm_f32acc(xl, wh) + mm_f32acc(xm, wm) + mm_f32acc(xh, wm) + mm_f32acc(xm, wh) + mm_f32acc(xh, wh)
    three = mm_f32acc(xh, wm) + mm_f32acc(xm, wh) + mm_f32acc(xh, wh)
    chain = np.zeros((M, N), np.float32)
    for k in range(K):          # an fp32 fmaf chain in k order (what v_mfma_f32_32x32x2_f32 computes)
        chain = (chain.astype(np.float64) + x[:, k:k + 1].astype(np.float64) * w[k:k + 1].astype(np.float64)).astype(np.float32)
    print(f"mean |result| {np.abs(ref).mean():.3f}")
    for name, v in (("fp32 fma chain", chain), ("bf16 x 6", six), ("bf16 x 3", three)):
        print(f"{name:16s} max abs err {np.abs(v - ref).max():.3e}   rms {np.sqrt(((v - ref) ** 2).mean()):.3e}")
    assert np.abs(six - ref).max() <= np.abs(chain - ref).max() * 2


if __name__ == "__main__":
    main()
