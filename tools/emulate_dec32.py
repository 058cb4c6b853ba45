#!/usr/bin/env python3
"""CPU emulation of the index math of whisperkit_amd/csrc/decoder32.hip (development tool, no GPU needed).

Transcribes, expression by expression, what the kernels do with lanes / waves / workgroups - weight tiling, activation planes,
the v_mfma_f32_32x32x16_f16 fragment and accumulator maps, the K split over waves and workgroups, the epilogue coordinates, the
LayerNorm fold with Chan-combined row-tile statistics - and checks the result against plain numpy (LayerNorm -> matmul).
Run: python tools/emulate_dec32.py
"""
import numpy as np

rng = np.random.default_rng(0)


def tile_weights(W):                       # d32_tile_weights_kernel
    N, K = W.shape
    n_rt, KT = (N + 31) // 32, K // 16
    out = np.zeros((n_rt, KT, 64, 8), np.float32)
    for o in range(n_rt * KT * 64):
        lane, tile = o & 63, o >> 6
        kt, rt = tile % KT, tile // KT
        row, k = rt * 32 + (lane & 31), kt * 16 + 8 * (lane >> 5)
        if row < N:
            out[rt, kt, lane] = W[row, k:k + 8]
    return out


def plane_index(b, n, K):                  # dec_shared.h
    return ((((b >> 5) * (K >> 4) + (n >> 4)) * 2 + ((n >> 3) & 1)) * 32 + (b & 31)) * 8 + (n & 7)


def mfma(a_frag, b_frag, acc):             # a_frag, b_frag [64][8]; acc [64][16]; first operand rows -> D rows
    A = np.zeros((32, 16), np.float64); Bm = np.zeros((16, 32), np.float64)
    for l in range(64):
        A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = a_frag[l]
        Bm[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = b_frag[l]
    D = A @ Bm
    for l in range(64):
        for r in range(16):
            acc[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return acc


def proj(Wt, N, K, zhi, zlo, ks, bt=0):
    """dec32_proj_kernel up to the summed 32 x 32 tile per thread: returns v[rt][tid][4]"""
    n_rt, KT = (N + 31) // 32, K // 16
    tw = K // (64 * ks)
    out = np.zeros((n_rt, 256, 4))
    zh = zhi.reshape(-1, 64, 8); zl = zlo.reshape(-1, 64, 8) if zlo is not None else None
    for rt in range(n_rt):
        part = np.zeros((ks, 256, 4))
        for ksi in range(ks):
            red = np.zeros((4, 16, 64))
            for wave in range(4):
                kt0 = (ksi * 4 + wave) * tw
                acc_h = np.zeros((64, 16)); acc_l = np.zeros((64, 16))
                for t in range(tw):
                    w = Wt[rt, kt0 + t]                       # wp + t*64 + lane
                    h = zh[bt * KT + kt0 + t]                  # hp: ((bt*KT + kt0)*64 + lane) + t*64
                    acc_h = mfma(w, h, acc_h)
                    if zl is not None:
                        acc_l = mfma(w, zl[bt * KT + kt0 + t], acc_l)
                for lane in range(64):
                    for r in range(16):
                        red[wave, r, lane] = acc_h[lane, r] + (acc_l[lane, r] / 2048.0 if zl is not None else 0.0)
            for tid in range(256):
                wave, lane = tid >> 6, tid & 63
                for i in range(4):
                    part[ksi, tid, i] = sum(red[w, 4 * wave + i, lane] for w in range(4))
        out[rt] = part.sum(0)
    return out


def split_hilo(z):
    hi = z.astype(np.float16).astype(np.float32)
    lo = ((z - hi) * 2048.0).astype(np.float16).astype(np.float32)
    return hi, lo


def make_planes(x, gamma, K, n_slots=32):
    """producer side: planes of z = gamma * x for slots b < n_slots (d32_resid_tail), + row-tile statistics"""
    B, d = x.shape
    zhi = np.zeros(((B + 31) // 32) * 32 * K, np.float32); zlo = np.zeros_like(zhi)
    z = (gamma[None, :] * x).astype(np.float32)
    hi, lo = split_hilo(z)
    for b in range(B):
        for n in range(K):
            zhi[plane_index(b, n, K)] = hi[b, n]; zlo[plane_index(b, n, K)] = lo[b, n]
    n_rt = d // 32
    stat = np.zeros((n_rt, 32, 2), np.float32)
    for rt in range(n_rt):
        for b in range(B):
            seg = x[b, rt * 32:(rt + 1) * 32].astype(np.float32)
            m = np.float32(seg.sum() / 32)
            stat[rt, b] = (m, ((seg - m) ** 2).sum())
    return zhi, zlo, stat


def chan_stats(stat, j, n_stat, d):
    st_l = np.zeros((8, 3))
    for sub in range(8):
        cn = cm = cM2 = 0.0
        for i in range(5):
            idx = sub + 8 * i
            if idx < n_stat:
                mb, M2b = stat[idx, j]
                nn = cn + 32.0; delta = mb - cm
                cm += delta * (32.0 / nn); cM2 += M2b + delta * delta * (cn * 32.0 / nn); cn = nn
        st_l[sub] = (cn, cm, cM2)
    cn = cm = cM2 = 0.0
    for s in range(8):
        nb, mb, M2b = st_l[s]
        if nb == 0:
            continue
        nn = cn + nb; delta = mb - cm
        cm += delta * (nb / nn); cM2 += M2b + delta * delta * (cn * nb / nn); cn = nn
    return cm, 1.0 / np.sqrt(cM2 / d + 1e-5)


def check(d, N, ks, B):
    K = d
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float16).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32); beta = (0.1 * rng.standard_normal(K)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(N)).astype(np.float32)
    x = (rng.standard_normal((B, d)) * 2 + 3.0).astype(np.float32)            # large mean: stresses the fold's cancellation
    zhi, zlo, stat = make_planes(x, gamma, K)
    Wt = tile_weights(W)
    g = (W.astype(np.float64) @ gamma).astype(np.float32); c = (W.astype(np.float64) @ beta + bias).astype(np.float32)
    v = proj(Wt, N, K, zhi, zlo, ks)
    mu = x.mean(1, keepdims=True); var = ((x - mu) ** 2).mean(1, keepdims=True)
    ref = ((x - mu) / np.sqrt(var + 1e-5) * gamma + beta).astype(np.float64) @ W.T.astype(np.float64) + bias
    worst = 0.0
    n_rt = (N + 31) // 32
    for rt in range(n_rt):
        for tid in range(256):
            j, sub = tid & 31, tid >> 5
            n = rt * 32 + 4 * sub
            if j >= B:
                continue
            m, rstd = chan_stats(stat, j, d // 32, d)
            for i in range(4):
                if n + i < N:
                    y = rstd * (v[rt, tid, i] - m * g[n + i]) + c[n + i]
                    worst = max(worst, abs(y - ref[j, n + i]))
    print(f"d={d} N={N} ks={ks} B={B}: max |fold-MFMA - LayerNorm-matmul| = {worst:.2e}")
    assert worst < 2e-4, worst


if __name__ == "__main__":
    check(128, 128, 2, 5)
    check(128, 384, 1, 32)
    check(128, 70, 1, 3)        # partial last row tile (logits)
    print("ok")
