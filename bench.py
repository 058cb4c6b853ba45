#!/usr/bin/env python3
"""Headline benchmark: audio-seconds per second of the Whisper hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one pass of the whole hot path over one batch of synthetic 30 s chunks that are already resident in HBM:
log-mel -> audio encoder -> cross-K/V projection -> greedy token loop (WhisperKit decodeText semantics, 223 decoder
forward passes, filters + sampler on device) -> result records on the host (+ all-gather over RCCL when N > 1).
Default workload = BASELINE.json configs[1]: whisper-tiny.en, one 30 s 16 kHz chunk, greedy, 1 GPU.  Weights are
random-init (no checkpoints in the image), so EOT is never the argmax and the loop runs to the reference's length cap
(sampleLength 224 -> 223 steps) - the decode length is therefore fixed and comparable across runs.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant decoder kernel, HIP-event timed
on the session stream) and `cpu_baseline` (the CPU oracle on the host cores, same chunk, rank 0, N == 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KERNEL_NAMES = ["dec_gemv<QKV>", "dec_self_attn", "dec_gemv<CQ>", "dec_cross_attn", "dec_gemv<FC1>", "dec_gemv<FC2>",
                "dec_gemv<LOGITS>", "sampler"]
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(kind: int, dims, B: int, avg_len: float) -> float:
    """Bytes one launch of each decoder kernel must move (fp16 weights/KV, fp32 activations), DESIGN.md section 5."""
    d, H, L, V = dims.n_text_state, dims.n_text_head, dims.n_text_layer, dims.n_vocab
    act = B * d * 4
    if kind == 0:   # LN1 + QKV: W[3d][d] + x in, q/k/v out
        return 3 * d * d * 2 + 3 * d * 4 + act + B * 3 * d * 2
    if kind == 1:   # self attention: K,V rows of <= len positions + W_o + partial out
        return B * 2 * avg_len * d * 2 + d * d * 2 + act + B * H * d * 4
    if kind == 2:   # combine + LN2 + cross query
        return d * d * 2 + B * H * d * 4 + 2 * act
    if kind == 3:   # cross attention: 1500 K and V rows per slot + W_o + partial out
        return B * 2 * 1500 * d * 2 + d * d * 2 + act + B * H * d * 4
    if kind == 4:   # combine + LN3 + fc1
        return 4 * d * d * 2 + B * H * d * 4 + act + B * 4 * d * 2
    if kind == 5:   # fc2
        return 4 * d * d * 2 + B * 4 * d * 2 + 2 * act
    if kind == 6:   # final LN + tied-embedding logits
        return V * d * 2 + act + B * V * 4
    return B * V * 4    # sampler: one pass over the logits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="tiny.en")
    ap.add_argument("--batch", type=int, default=1, help="30 s chunks per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE {world}")

    import torch
    import torch.distributed as dist
    from whisperkit_amd import _lib as L
    from whisperkit_amd import api, parallel, weights
    from whisperkit_amd.synth import synthetic_chunk

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the whisperhip product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    dims = weights.MODEL_DIMS[args.model]
    B = args.batch
    model = api.Model.synthetic(args.model, seed=0, device=local_rank)
    sess = api.Session(model, B)
    # weak scaling: every rank owns B chunks; global chunk index = rank * B + b
    first, _ = parallel.partition_chunks(world * B, world, rank)
    chunks = [synthetic_chunk(1234 + first + b) for b in range(B)]
    for b, x in enumerate(chunks):
        sess.padOrTrim(x, b)                       # PCM resident in HBM before the timed region
    opts = api.DecodingOptions(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None,
                               noSpeechThreshold=None, temperatureFallbackCount=0)
    prompt = sess.prefillPrompt(opts)

    def hot_path():
        sess.logMelSpectrogram(B)
        sess.encodeFeatures(B)
        sess.prepareDecoderInputs(B)
        res = sess.decodeText(prompt, opts, batch=B)
        recs = np.stack([parallel.pack_record(first + b, r.tokens, 0, r.steps, r.avgLogProb, r.temperature, r.compressionRatio)
                         for b, r in enumerate(res)])
        allrecs = parallel.gather_records(recs, B, device=dev if world > 1 else None)
        return res, allrecs

    def fence():
        sess.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        res, allrecs = hot_path()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, allrecs = hot_path()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert len(allrecs) == world * B, (len(allrecs), world, B)
    dec_steps = [r.steps for r in res]
    audio_s = world * B * 30.0 * args.steps
    value = audio_s / elapsed

    # ---- stage split (rank 0): mel + encoder milliseconds per chunk, decode tokens/s
    stage = {}
    if rank == 0:
        ts = []
        for _ in range(5):
            sess.synchronize(); a = time.perf_counter()
            sess.logMelSpectrogram(B); sess.synchronize(); b_ = time.perf_counter()
            sess.encodeFeatures(B); sess.synchronize(); c = time.perf_counter()
            sess.prepareDecoderInputs(B); sess.synchronize(); d_ = time.perf_counter()
            r2 = sess.decodeText(prompt, opts, batch=B); e = time.perf_counter()
            ts.append((b_ - a, c - b_, d_ - c, e - d_))
        med = np.median(np.array(ts), axis=0)
        stage = {"logmels_ms_per_chunk": med[0] * 1e3 / B, "encoder_ms_per_chunk": med[1] * 1e3 / B,
                 "cross_kv_ms_per_chunk": med[2] * 1e3 / B, "decode_ms_per_chunk": med[3] * 1e3 / B,
                 "decoder_steps": int(r2[0].steps), "tokens_per_s": B * r2[0].steps / med[3],
                 "us_per_decoder_step": med[3] * 1e6 / max(r2[0].steps, 1)}

    # ---- roofline of the dominant decoder kernel (HIP events around every launch, on the session stream)
    roofline = None
    if rank == 0 and not args.no_roofline:
        n_meas = 64
        avg = (ctypes.c_double * 8)()
        cnt = (ctypes.c_int32 * 8)()
        api._check(sess.lib.wh_measure_decoder_kernels(sess.handle, B, n_meas, avg, cnt))
        tot = [avg[k] * cnt[k] for k in range(8)]
        dom = int(np.argmax(tot))
        avg_len = (n_meas + 1) / 2.0
        kernels = {KERNEL_NAMES[k]: {"avg_us": round(avg[k], 3), "launches": int(cnt[k]),
                                     "alg_bytes": int(algorithmic_bytes(k, dims, B, avg_len)),
                                     "GBps": round(algorithmic_bytes(k, dims, B, avg_len) / (avg[k] * 1e-6) / 1e9, 1) if avg[k] > 0 else None}
                   for k in range(8)}
        by = algorithmic_bytes(dom, dims, B, avg_len)
        ach = by / (avg[dom] * 1e-6) / 1e9
        roofline = {"kernel": KERNEL_NAMES[dom], "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "avg_us": round(avg[dom], 3),
                    "alg_bytes_per_launch": int(by), "share_of_step_time": round(tot[dom] / sum(tot), 3), "kernels": kernels,
                    "note": "eager launches with an event pair per kernel; weights (59 MB) sit in the 256 MiB Infinity Cache, so "
                            "the HBM peak is the conservative denominator"}

    # ---- CPU baseline: the oracle (port of the same algorithm) on the host cores, same chunk, whole window
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import decode as OD
        from oracle import mel as omel
        from oracle.model import OracleWhisper
        torch.set_num_threads(os.cpu_count() or 1)
        sd = weights.synthetic_state_dict(dims, seed=0)
        om = OracleWhisper(dims, sd)
        st, langs = OD.special_tokens_for_vocab(dims.n_vocab)
        oopts = OD.DecodingOptions(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None,
                                   noSpeechThreshold=None, temperatureFallbackCount=0)
        c0 = time.perf_counter()
        omel_ = omel.log_mel_spectrogram(chunks[0], dims.n_mels).astype(np.float32)
        c1 = time.perf_counter()
        enc = om.encode(omel_)
        c2 = time.perf_counter()
        state = om.new_state(enc)
        ores = OD.decode_text(lambda t, p: state.step(t, p, want_alignment=False), OD.prefill_prompt(oopts, st, dims.is_multilingual),
                              OD.GreedyTokenSampler(0.0, st.endToken, oopts), oopts, st, dims.is_multilingual, langs)
        c3 = time.perf_counter()
        same = ores.tokens == res[0].tokens if first == 0 else None
        cpu = {"value": round(30.0 / (c3 - c0), 3), "unit": "audio-sec/sec", "cores": os.cpu_count(), "kind": "port",
               "sample": f"one 30 s chunk end to end ({ores.steps} decoder steps): mel {c1 - c0:.2f} s, encoder {c2 - c1:.2f} s, "
                         f"decode {c3 - c2:.2f} s (torch fp32, {torch.get_num_threads()} threads)",
               "tokens_equal_gpu": same}

    if rank == 0:
        out = {
            "metric": "audio-sec/sec (1/RTF), 30 s chunks: log-mel + encoder + greedy decode",
            "value": round(value, 2), "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"whisper-{args.model}, {B} x 30 s 16 kHz chunk per GPU, greedy (T=0), "
                                   f"{dec_steps[0]} decoder steps/chunk, random-init weights, PCM resident in HBM",
                       "chunks_per_gpu": B, "parallelism": f"chunk-dp{world}", "decoder_steps": dec_steps[0],
                       "arith": "fp16 operands, fp32 accumulate/residual/softmax; mel fp32"},
            "rtf": round(elapsed / audio_s, 6),
            "stages": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in stage.items()},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
