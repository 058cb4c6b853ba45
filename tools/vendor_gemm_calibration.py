"""Calibration, not product: what the vendor GEMM library (hipBLASLt / rocBLAS behind torch.mm) and torch's fused attention reach on
THIS chip for the encoder's shapes, beside csrc/gemm.hip's and csrc/attention.hip's own launch durations (bench.py `roofline.kernels`).

The dense-f16 MFMA peak DESIGN.md prices the encoder against (2.5 PFLOP/s) is a clock x width product; what a tuned library sustains on
a K = 1280 problem is the practical ceiling for a 256 x 256 tile design.  This script times the PLAIN products (no bias, no GELU, no
residual, f16 output) of the four encoder projections of large-v3 at 64 and 128 chunks, so the library's number is an upper bound for a
kernel that also has to run the epilogue.  Nothing in whisperkit_amd/ imports torch.mm: the encoder stays csrc/gemm.hip.

    python tools/vendor_gemm_calibration.py [--out gpurun_out/r05q_vendor_gemm_calibration.json]
"""
import argparse
import json
import os
import sys

import torch

SHAPES = {"enc_qkv": (3840, 1280), "enc_o": (1280, 1280), "enc_fc1": (5120, 1280), "enc_fc2": (1280, 5120)}      # name -> (N, K), large-v3


def time_us(fn, warmup=5, iters=20):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r05q_vendor_gemm_calibration.json")
    ap.add_argument("--bench-json", default="profiles/r05i_bench_steps20_warmup5.json", help="bench line whose per-kernel durations (128 chunks) are printed beside")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    own = {}
    if os.path.exists(args.bench_json):
        with open(args.bench_json) as f:
            ks = json.load(f)["roofline"]["kernels"]
        own = {k[len("gemm_"):]: v["avg_us"] for k, v in ks.items() if k.startswith("gemm_enc_")}
        own["enc_attention"] = ks["enc_attention"]["avg_us"]
    rep = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "peak_tflops_dense_f16": 2500.0, "gemm": {}, "attention": {}}
    for chunks in (64, 128):
        M = chunks * 1500
        for name, (N, K) in SHAPES.items():
            A = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g) * 0.5
            W = torch.randn(N, K, device=dev, dtype=torch.float16, generator=g) * 0.03
            out = torch.empty(M, N, device=dev, dtype=torch.float16)
            us = time_us(lambda: torch.mm(A, W.t(), out=out))
            flop = 2.0 * M * N * K
            row = {"M": M, "N": N, "K": K, "vendor_us": round(us, 1), "vendor_tflops": round(flop / us * 1e-6, 1), "vendor_frac_of_peak": round(flop / us * 1e-6 / 2500.0, 4)}
            if chunks == 128 and name in own:
                row["whisperhip_us_with_epilogue"] = own[name]
                row["whisperhip_tflops"] = round(flop / own[name] * 1e-6, 1)
                row["whisperhip_over_vendor_time"] = round(own[name] / us, 3)
            rep["gemm"][f"{name}@{chunks}"] = row
            del A, W, out
        # encoder self-attention: 20 heads x 64, 1500 keys, no mask; 4 * T * T * 64 flop per head and chunk
        try:
            q = torch.randn(chunks, 20, 1500, 64, device=dev, dtype=torch.float16, generator=g)
            k = torch.randn(chunks, 20, 1500, 64, device=dev, dtype=torch.float16, generator=g)
            v = torch.randn(chunks, 20, 1500, 64, device=dev, dtype=torch.float16, generator=g)
            us = time_us(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
            flop = 4.0 * chunks * 20 * 1500 * 1500 * 64
            row = {"vendor_us": round(us, 1), "vendor_tflops": round(flop / us * 1e-6, 1)}
            if chunks == 128 and "enc_attention" in own:
                row["whisperhip_us"] = own["enc_attention"]
                row["whisperhip_over_vendor_time"] = round(own["enc_attention"] / us, 3)
            rep["attention"][f"sdpa@{chunks}"] = row
            del q, k, v
        except Exception as e:      # no fused attention backend in this build: recorded, not fatal
            rep["attention"][f"sdpa@{chunks}"] = {"error": repr(e)[:200]}
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rep, f, indent=1, sort_keys=True)
    json.dump(rep, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
