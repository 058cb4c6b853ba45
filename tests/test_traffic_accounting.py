"""What the decoder projections' PMC traffic above their algorithmic bytes is made of (CPU only: it reads the committed counters).

profiles/r06zz_pmc_traffic.json holds rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of 16 decoder steps of one 256-slot large-v3 session on the final binary.
FETCH_SIZE counts what the eight XCD-private L2s request from the fabric; a projection launch spreads its weight-row tiles over all eight XCDs (decoder32.hip: workgroup ids
x + 8 t share slab x), so every weight byte crosses once and the hi | lo activation planes - counted once by the algorithmic bytes - enter each of the eight L2s once.
The test re-derives the measured fetch of every projection from that statement within a few per cent: the excess is L2 replication of the planes, not slab re-fetching."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, D, XCDS = 256, 1280, 8
MB = 1e6


def _fetch_mb(counters, name):
    return counters[name]["fetch_size_kb"] * 1024 * 2 / MB          # x 2: the gfx950 correction of FETCH_SIZE (MI355X_MICROARCH.md, tools/pmc_traffic.py)


def _planes(k):
    return B * k * 2 * 2 / MB                                        # f16 hi + f16 lo plane of K channels for 256 slots


def test_projection_fetch_is_weights_once_plus_planes_into_every_xcd_l2():
    t = json.load(open(os.path.join(ROOT, "profiles", "r06zz_pmc_traffic.json")))
    assert t["chunks_per_step"] == B and t["model"] == "large-v3"
    c = t["counters"]
    x_f32 = B * D * 4 / MB                                           # the fp32 residual a RESID epilogue reads
    stats = B * (D // 32) * 8 * XCDS / MB                            # LayerNorm partials (mean, M2) per row tile and slot, read in every XCD
    cases = {
        # name: (N, K, other fetched bytes in MB)
        "dec_proj_oproj": (D, D, x_f32),
        "dec_proj_coproj": (D, D, x_f32),
        "dec_proj_cq": (D, D, stats),
        "dec_proj_fc1": (4 * D, D, stats),
        "dec_proj_fc2": (D, 4 * D, x_f32 + 4 * B * D * 4 / MB),      # + the partial tiles of 4 K slices, re-read by the finishing workgroups
    }
    for name, (n, k, other) in cases.items():
        weights = n * k * 2 / MB
        predicted = weights + XCDS * _planes(k) + other
        measured = _fetch_mb(c, name)
        # fc2's planes (5.2 MB) do not fit a 4 MB L2 beside the weight stream: a tenth of them is fetched a second time
        assert measured == pytest.approx(predicted, rel=0.09 if name == "dec_proj_fc2" else 0.05), (name, predicted, measured)
        # the statement the round-4 / round-5 reviews asked for: no slab is fetched twice - with the planes counted once the weights would have to cross 3 - 4 times
        assert (measured - XCDS * _planes(k) - other) / weights == pytest.approx(1.0, abs=0.4 if name == "dec_proj_fc2" else 0.2), name
    # qkv is profiled under its own name in the bench's table only; its bytes_per_launch minus its stores follows the same rule
    qkv_total = t["bytes_per_launch"]["dec_proj_qkv"] / MB
    qkv_writes = (B * D * 4 + 2 * B * D * 2) / MB                    # q in fp32, k and v rows in Float16
    assert qkv_total - qkv_writes == pytest.approx(3 * D * D * 2 / MB + XCDS * _planes(D) + stats, rel=0.06)


def test_dominant_kernel_fetches_its_algorithmic_bytes_once():
    t = json.load(open(os.path.join(ROOT, "profiles", "r06zz_pmc_traffic.json")))
    H = 20
    alg = B * 1500 * D * 2 + B * H * D * 4 + B * H * (D * 4 + 8)
    assert t["bytes_per_launch"]["dec_cross_attn"] / alg == pytest.approx(1.0, abs=0.02)
