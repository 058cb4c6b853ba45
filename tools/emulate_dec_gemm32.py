#!/usr/bin/env python3
"""Index-math check of the DRAFT kernel csrc/experimental/dec_gemm32.hip on the CPU: every lane of every wave is emulated with the
fragment layout of v_mfma_f32_32x32x16_f16 that gemm.hip already relies on on hardware (a: lane -> row l & 31, k 8 (l >> 5) .. +8;
b: lane -> column l & 31, same k; d: lane -> column l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)).  It checks the addressing of
the weight / activation fragments, the cross-wave reduction through s_red and the epilogue thread mapping - not the hardware."""
import numpy as np


def mfma_32x32x16(a_frag, b_frag, acc):
    """a_frag, b_frag: [64 lanes][8]; acc: [64 lanes][16].  D[m][n] += sum_k A[m][k] B[k][n]."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for l in range(64):
        A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = a_frag[l]
        B[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = b_frag[l]
    D = A @ B
    for l in range(64):
        for r in range(16):
            acc[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return acc


def emulate(W, act, batch, n0):
    N, K = W.shape
    kq = K // 4
    s_red = np.zeros((4, 32, 33))
    for wave in range(4):
        acc = np.zeros((64, 16))
        for k0 in range(0, kq, 16):
            a_frag = np.zeros((64, 8)); b_frag = np.zeros((64, 8))
            for l in range(64):
                b, kh = l & 31, l >> 5
                k = wave * kq + k0 + kh * 8
                a_frag[l] = W[n0 + b, k:k + 8]                                   # wrow + k0 + s * 16
                b_frag[l] = act[b, k:k + 8] if b < batch else 0.0                # s_act[b][wave * kq + k0 + s * 16 + kh * 8]
            acc = mfma_32x32x16(a_frag, b_frag, acc)
        for l in range(64):
            b, kh = l & 31, l >> 5
            for r in range(16):
                s_red[wave, (r & 3) + 8 * (r >> 2) + 4 * kh, b] = acc[l, r]
    y = np.zeros((batch, 32))
    for t in range(256):
        bb, ng = t & 31, (t >> 5) * 4
        if bb >= batch:
            continue
        for j in range(4):
            y[bb, ng + j] = ((s_red[0, ng + j, bb] + s_red[1, ng + j, bb]) + s_red[2, ng + j, bb]) + s_red[3, ng + j, bb]
    return y


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for batch, N, K in ((32, 64, 128), (20, 96, 320), (8, 32, 64)):
        W = rng.standard_normal((N, K)); act = rng.standard_normal((batch, K))
        for n0 in range(0, N, 32):
            y = emulate(W, act, batch, n0)
            want = act @ W[n0:n0 + 32].T
            assert np.allclose(y, want, atol=1e-9), (batch, N, K, n0, np.abs(y - want).max())
    print("dec_gemm32 index math ok")
