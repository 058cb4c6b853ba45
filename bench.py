#!/usr/bin/env python3
"""Headline benchmark: audio-seconds per second of the Whisper hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one pass of the whole hot path over one batch of synthetic 30 s chunks, the way SURVEY.md section 8(d) specifies
it - "PCM in host memory -> token ids + segments on host": padOrTrim from host float32 PCM (wh_set_audio) -> log-mel -> audio
encoder -> cross-K/V projection -> greedy token loop (WhisperKit decodeText semantics, filters + sampler on device) ->
findSeekPointAndSegments per chunk on the host -> result records (+ all-gather over RCCL when N > 1).

Default workload = BASELINE.json configs[3]'s model and chunk set: whisper-large-v3 (128 mel), 64 x 30 s chunks per step, greedy.  The engine
batches continuously: FOUR consecutive 64-chunk steps share one 256-slot device batch (eight 32-slot MFMA batch tiles per decode launch: the
decoder's weight stream and its latency-bound launch chain are paid once per 256 windows; a workgroup of the HBM-bound cross-attention streams
two slots one after the other, so that launch still takes 128 workgroups = half of the CUs), and 3 device batches are in flight (one session /
HIP stream / host thread each: the encoder GEMMs and the projection kernels of one batch run beside the cross-attention stream of the others) -
12 steps = 768 chunks are resident on the GPU, and a step is a unit of THROUGHPUT: one 64-chunk step alone takes `serial_ms_per_step`.
`--device-batch 128` is the round-5 configuration (two steps per batch), `--device-batch 64` round 4's.  Measured on one MI355X: 64-slot batches x 3
in flight 2350 audio-s/s, 128 x 3 2647 - 2660, 256 x 3 with 256 workgroups per launch 2584 - 2692 (profiles/r05a_*, r05d_*), 256 x 3 with two slots per
workgroup 2749 (profiles/r06i_*).  Weights are random-init (no checkpoints in
the image), so EOT is never the argmax and the loop runs to the reference's length cap (sampleLength 224 -> 223 decoder forward
passes per chunk): the decode length is fixed and comparable across runs.  Round 1's configuration (8 chunks per step, 3 in
flight) and the other BASELINE configs are measured after the headline and reported under "other_configs", each with its own `roofline`.

N > 1 (configs[3]: "64 x 30 s chunks sharded across 8 x MI355X"): STRONG scaling by default - a step is still 64 chunks in total,
block-partitioned over the ranks (64 / N per GPU, no data-path collective), the per-chunk result records of every step are
all-gathered through the C-ABI communicator (wh_comm_*: ncclAllGather over xGMI, librccl dlopen'ed by libwhisperhip).  A GPU packs
its shares of up to G consecutive steps into one device batch of at most 256 slots (continuous batching, as at one GPU; never more steps
than a session's share of the run: plan_batches); `--scaling weak` is the old mode (64 chunks per step AND GPU).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  `roofline`      the dominant kernel of the step, HIP-event timed on the session stream (wh_measure_kernels), against the
                  HBM or dense-f16-MFMA peak, with the per-kernel table it was picked from;
  `cpu_baseline`  the CPU oracle (a port: torch fp32 + the restated WhisperKit loop) on the host cores, rank 0, N == 1 only,
                  on a bounded sample (one chunk: mel + encoder + 16 decoder steps, extrapolated to 223 steps); its first tokens
                  must equal the GPU's (hard failure otherwise).
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F16_PEAK_TF = 2500.0  # dense f16/bf16 MFMA peak (2495 TF measured, 32x32x16)
MFMA_F32_PEAK_TF = 157.3   # f32-input MFMA (v_mfma_f32_16x16x4_f32: exact f32, 64 FLOP/clk/SIMD; MI355X_MICROARCH.md) - the log-mel's DFT runs on it
PMC_TRAFFIC_FILE = "r06_pmc_traffic.json"
AUDIO_SETS = 8             # distinct chunk sets a run cycles through (the 4 packed steps of a device batch carry 4 different sets, consecutive batches differ)
T_START = time.perf_counter()


def log(msg):
    print(f"[bench {time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def algorithmic_work(kind: str, dims, B: int, avg_len: float, fused_sampler: bool = True, absorbed: bool = False, splits: int = 4):
    """(bound, amount) one launch of each kernel kind must do: HBM bytes for the bandwidth-bound kernels, FLOPs for the
    MFMA-bound ones.  fp16 weights / KV / GEMM operands, fp32 residuals; decoder
    activations travel as f16 hi|lo plane pairs (4 B per element)."""
    d, H, L, V = dims.n_text_state, dims.n_text_head, dims.n_text_layer, dims.n_vocab
    nm, T, F = dims.n_mels, 1500, 3000
    act = B * d * 4          # one f32 row set, or one hi|lo plane pair
    gemm = lambda M, N, K: ("mfma", 2.0 * M * N * K)
    if kind == "mel_power":      # PCM in, f32 log-mel scratch out
        return "hbm", B * (480000 * 4 + nm * F * 4)
    if kind == "mel_finalize":   # scratch in, f16 time-major + f32 reference-layout out
        return "hbm", B * (nm * F * 4 + nm * F * 2 + nm * F * 4)
    if kind == "gemm_conv1":
        return gemm(B * F, d, 3 * nm)
    if kind == "gemm_conv2":
        return gemm(B * T, d, 3 * d)
    if kind == "layernorm":      # f32 row in, f16 row out
        return "hbm", B * T * d * 6
    if kind == "gemm_enc_qkv":
        return gemm(B * T, 3 * d, d)
    if kind == "enc_attention":  # QK^T and PV
        return "mfma", 4.0 * B * T * T * d
    if kind == "gemm_enc_o":
        return gemm(B * T, d, d)
    if kind in ("gemm_enc_fc1", "gemm_enc_fc2"):
        return gemm(B * T, 4 * d, d)
    if kind == "gemm_cross_kv":
        return gemm(B * T, 2 * L * d, d)
    if kind == "dec_embed":      # embedding row + position row in, x + gamma x planes out
        return "hbm", B * d * (2 + 4) + 2 * act
    if kind == "dec_proj_qkv":   # W[3d][d] + planes in, q (f32) + k, v (f16) out
        return "hbm", 3 * d * d * 2 + act + act + B * 2 * d * 2
    if kind == "dec_self_attn":  # K,V rows of <= len cached positions, q in, att planes out
        return "hbm", B * 2 * avg_len * d * 2 + 2 * act
    if kind in ("dec_proj_oproj", "dec_proj_coproj"):   # W[d][d] + att planes in, x read + written, gamma x planes out
        return "hbm", d * d * 2 + 4 * act
    if kind == "dec_proj_cq":    # W[d][d] + planes in, q out
        return "hbm", d * d * 2 + 2 * act
    if kind == "dec_cross_attn" and absorbed:
        # weight-absorbed form (csrc/xabs.hip): the slot's encoder output [1500][d] f16 ONCE, absorbed queries (f16 hi | lo) in,
        # every key split's unnormalised O' [H][d] f32 + (m, l) out
        return "hbm", B * T * d * 2 + B * H * d * 4 + splits * B * H * (d * 4 + 8)
    if kind == "dec_cross_attn":  # 1500 K and V rows per slot (24-bit rows since round 5: Float16 + 8-bit residual), q in, att planes out
        return "hbm", B * 2 * T * d * 3 + 2 * act
    if kind == "dec_xabs_qk":    # W_k^T tiles + q in, absorbed queries [H][d] per slot (f16 hi | lo) out
        return "hbm", d * d * 2 + act + B * H * d * 4
    if kind == "dec_xabs_vup":   # W_v tiles + the split partials in, att planes out
        return "hbm", d * d * 2 + splits * B * H * (d * 4 + 8) + act
    if kind == "dec_proj_fc1":   # W[4d][d] + planes in, hidden hi|lo plane pair out
        return "hbm", 4 * d * d * 2 + act + B * 4 * d * 4
    if kind == "dec_proj_fc2":   # W[d][4d] + hidden plane pair in, x read + written, planes out
        return "hbm", 4 * d * d * 2 + B * 4 * d * 4 + 3 * act
    if kind == "dec_proj_logits":  # tied embedding [V][d] + planes in; per-tile sampler records (fused greedy) or the logits out
        return "hbm", V * d * 2 + act + (B * ((V + 31) // 32) * 32 if fused_sampler else B * V * 4)
    if kind == "sampler":        # merge of the per-tile records
        return "hbm", B * ((V + 31) // 32) * 32 if fused_sampler else B * V * 4
    raise KeyError(kind)


def measure_kernels(sess, dims, B, n_meas, decode_steps, model_name, steps_per_batch=1):
    """HIP-event timing of every kernel launch of one eager pass (mel, encoder, cross-K/V, n_meas decoder steps) over one device batch of B
    slots.  A device batch carries `steps_per_batch` bench steps: `launches_per_step` and everything derived from it is per bench step."""
    from whisperkit_amd import api
    lib = sess.lib
    nk = lib.wh_kernel_kind_count()
    names = [lib.wh_kernel_kind_name(k).decode() for k in range(nk)]
    avg = (ctypes.c_double * nk)()
    cnt = (ctypes.c_int32 * nk)()
    api._check(lib.wh_measure_kernels(sess.handle, B, n_meas, avg, cnt))
    avg_len = (n_meas + 1) / 2.0
    absorbed = lib.wh_session_cross_attention_mode(sess.handle) == 1     # the session's cross-attention streams the encoder output (csrc/xabs.hip)
    splits = int(lib.wh_session_cross_attention_splits(sess.handle)) if absorbed else 0     # key splits per slot (partials written and re-read)
    table, step_us = {}, {}
    for k, name in enumerate(names):
        if cnt[k] == 0:
            continue
        bound, amount = algorithmic_work(name, dims, B, avg_len, absorbed=absorbed, splits=splits)
        is_dec = name.startswith("dec_") or name == "sampler"
        per_step = (cnt[k] / n_meas * decode_steps if is_dec else cnt[k]) / steps_per_batch      # launches per bench step (a launch over a device batch serves steps_per_batch steps)
        step_us[name] = avg[k] * per_step
        rate = amount / (avg[k] * 1e-6)
        if bound == "hbm":
            ach, peak, unit = rate / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            ach, peak, unit = rate / 1e12, MFMA_F16_PEAK_TF, "TFLOP/s"
        table[name] = {"avg_us": round(avg[k], 3), "launches_measured": int(cnt[k]), "launches_per_step": round(per_step, 1),
                       "bound": bound, "alg_per_launch": int(amount), "achieved": round(ach, 2), "unit": unit,
                       "frac": round(ach / peak, 4), "share_of_step": None}
    if "mel_power" in table:
        # The log-mel's HBM figure above is what north_star asks for (GB/s on the mel path); what BOUNDS mel_power is its exact-f32 DFT on the f32 matrix
        # cores (csrc/mel.hip: per frame two real GEMMs of K = 200 / 199 folded samples x 201 bins), so the entry carries that fraction as well.
        flops = B * 3000 * (200 * 201 + 199 * 200) * 2.0
        tf = flops / (table["mel_power"]["avg_us"] * 1e-6) / 1e12
        table["mel_power"].update({"matrix_f32_flops_per_launch": int(flops), "matrix_f32_achieved_tflops": round(tf, 2), "matrix_f32_peak_tflops": MFMA_F32_PEAK_TF,
                                   "matrix_f32_frac": round(tf / MFMA_F32_PEAK_TF, 4)})
    tot = sum(step_us.values())
    # HBM traffic per launch from the off-line PMC passes (profiles/*_pmc_traffic.json), when they were taken on this workload
    traffic = {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))
        if tj.get("model") == model_name and tj.get("chunks_per_step") == B and (not absorbed or tj.get("cross_attention_splits", 4) == splits):
            traffic = tj["bytes_per_launch"]
    except (OSError, ValueError):
        pass
    for name in table:
        table[name]["share_of_step"] = round(step_us[name] / tot, 4)
        table[name]["traffic"] = traffic.get(name)
    dom = max(step_us, key=step_us.get)
    t = table[dom]
    return {"kernel": dom, "cross_attention": f"absorbed (encoder output streamed once per layer, {splits} key splits per slot, csrc/xabs.hip)" if absorbed else "per-layer K / V rows",
            "bound": t["bound"], "achieved": t["achieved"],
            "peak": HBM_PEAK_GBS if t["bound"] == "hbm" else MFMA_F16_PEAK_TF, "unit": t["unit"], "frac": t["frac"], "traffic": t["traffic"],
            "avg_us": t["avg_us"], "alg_per_launch": t["alg_per_launch"], "share_of_step_time": t["share_of_step"],
            "sum_kernel_ms_per_step": round(tot / 1e3, 3), "kernels": table,
            "note": f"one device batch = {B} slots = {steps_per_batch} bench step(s): launches_per_step, sum_kernel_ms_per_step and share_of_step are per bench step; "
                    "eager launches, one HIP event pair per launch on the session stream (the pair itself adds ~2 us to short kernels: "
                    "fractions of the projection kernels are lower bounds); decoder kernels averaged over "
                    f"{n_meas} steps (positions 0..{n_meas - 1}) and weighted to {decode_steps} steps; traffic = HBM bytes per launch "
                    f"from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/{PMC_TRAFFIC_FILE}; FETCH_SIZE x2 "
                    "per the gfx950 correction), null when no pass exists for this workload"}


_MODELS = {}


def get_model(name, local_rank, keep_sd=False):
    """(Model, dims, state dict or None); models are reused by the configurations that share them"""
    from whisperkit_amd import api, weights
    if name not in _MODELS:
        dims = weights.MODEL_DIMS[name]
        log(f"building synthetic {name} weights")
        sd = weights.synthetic_state_dict(dims, seed=0)
        _MODELS[name] = [api.Model(dims, sd, device=local_rank), dims, sd if keep_sd else None]
    return _MODELS[name]


def plan_batches(n_steps, F, G):
    """n_steps steps over F workers (sessions in flight) as device batches of at most G steps: every worker gets an equal share of the
    steps (the first n_steps % F one more), cut into the fewest batches of at most G steps, as equal as possible (a share of 6 with G = 4 is
    3 + 3, not 2 + 4: no batch is much smaller than the others), the shorter ones first.  Returns [[steps per batch, ...] per worker]."""
    F = max(1, min(F, n_steps))
    plan = []
    for w in range(F):
        mine = n_steps // F + (1 if w < n_steps % F else 0)
        nb = -(-mine // G)
        base, rem = divmod(mine, nb) if nb else (0, 0)
        plan.append([base] * (nb - rem) + [base + 1] * rem)                     # the short batches FIRST: the run ends on full batches (a 20-step run
    return plan                                                                 # that ended on two single-step batches was 5 % slower over all, profiles/r05f_*)


def run_config(args, model_name, B, F, steps, warmup, world, rank, local_rank, dev, want_roofline, want_cpu, word_timestamps=False,
               comm=None, scaling="strong", device_batch=None, whole_chip_leg=True):
    """B = chunks per step: in total over the ranks (strong scaling, the default) or per rank (weak).  A rank's share of a step is the
    contiguous block partition_chunks gives it; the shares of up to G consecutive steps are packed into one device batch of at most
    `device_batch` slots (continuous batching: default = B, i.e. G = 1 at one GPU; the headline packs two 64-chunk steps into one 128-slot
    batch).  The steps of a run are dealt to the F sessions in equal shares (plan_batches)."""
    import torch
    import torch.distributed as dist
    from whisperkit_amd import api, parallel
    from whisperkit_amd.synth import bench_chunk_seed, synthetic_chunk

    model, dims, sd = get_model(model_name, local_rank, keep_sd=want_cpu)
    F = max(1, F)
    total = B * world if scaling == "weak" else B               # chunks of one step over all ranks
    first, last = parallel.partition_chunks(total, world, rank)
    n_local = last - first                                       # this rank's chunks of one step
    per_rank_max = (total + world - 1) // world
    cap = min(256, max(device_batch or B, n_local))              # device batch capacity in slots (a session holds at most 256)
    G = max(1, cap // max(n_local, 1)) if n_local else 1         # steps packed into one device batch ...
    G = max(1, min(G, (steps + F - 1) // F))                     # ... never more than a session's share of the run
    slots = max(1, G * n_local)
    # sessions in flight: the cross-attention of one session takes about half of the 256 CUs (slots x splits = 128 workgroups), the other
    # sessions' kernels keep the rest (64 slots x 3: 2049 -> 2270 audio-s/s against 4 splits, 128 slots x 3: 2240 -> 2537; profiles/r04ad..af)
    xsplits = args.cross_attention_splits if args.cross_attention_splits >= 0 else (max(1, min(4, 128 // max(slots, 1))) if F > 1 else 0)
    # ... and beyond 128 slots a workgroup streams several slots one after the other, so the launch stays at 128 workgroups (round 6,
    # wh_session_options.cross_attention_slots_per_workgroup: 256-slot batches x 2 slots per workgroup 2749 audio-s/s against 2660 with 128-slot batches
    # and 2692 with 256 workgroups per launch, profiles/r06i_*)
    xspw = args.cross_attention_slots_per_workgroup if args.cross_attention_slots_per_workgroup > 0 else (max(1, -(-slots * max(xsplits, 1) // 128)) if F > 1 else 1)
    sessions = [api.Session(model, slots, crossAttentionSplits=xsplits or None, crossAttentionSlotsPerWorkgroup=xspw) for _ in range(F)]
    sess = sessions[0]
    # host float32 PCM: a pool of AUDIO_SETS chunk sets; step n of a run (steps numbered worker by worker, batch by batch: the order the records are
    # gathered in) carries set n % AUDIO_SETS, so the packed steps of a device batch and the sessions in flight carry DIFFERENT audio (VERDICT r05
    # weak 9: identical halves let the embedding kernel read one row for two slots) and an N-rank run sees the audio of the 1-rank run step by step.
    # Set 0 is the chunk set of rounds 1 - 5 (seeds 1234 + chunk index): the CPU baseline and tests/test_gpu_fulldepth.py check against it.
    n_sets = max(1, min(AUDIO_SETS, max(steps, warmup, F * G)))
    audio = [[np.ascontiguousarray(synthetic_chunk(bench_chunk_seed(first + b, k)), dtype=np.float32) for b in range(n_local)] for k in range(n_sets)]
    chunks = audio[0]
    opts = api.DecodingOptions(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None,
                               noSpeechThreshold=None, temperatureFallbackCount=0, sampleLength=args.sample_length,
                               wordTimestamps=word_timestamps)
    prompt = sess.prefillPrompt(opts)
    st = model.specialTokens

    def hot_path(ss, g=1, n0=0):
        """the rank's chunks of g consecutive steps (run steps n0 .. n0 + g - 1) as ONE device batch; returns (results, [records of step 0, step 1, ...], segments)"""
        nb = g * n_local
        if nb == 0:
            return [], [np.zeros((0, parallel.RECORD_INTS), np.int32) for _ in range(g)], 0
        for k in range(g):
            for b, x in enumerate(audio[(n0 + k) % n_sets]):
                ss.padOrTrim(x, k * n_local + b)       # PCM in host memory -> HBM, inside the timed region
        ss.logMelSpectrogram(nb)
        ss.encodeFeatures(nb)
        ss.prepareDecoderInputs(nb)
        res = ss.decodeText(prompt, opts, batch=nb)
        nseg = 0
        for b, r in enumerate(res):
            if word_timestamps:     # findAlignment (SegmentSeeker.swift:340-408): alignment rows of the result tokens -> DTW
                api.dynamicTimeWarping(ss.getAlignmentWeights(b)[:len(r.tokens)])
            _, segs = api.findSeekPointAndSegments(r.tokens, r.tokenLogProbs, opts, st, 0, 0, 480000, r.avgLogProb)   # segments on the host
            nseg += len(segs or ())
        recs = [np.stack([parallel.pack_record(first + b, res[k * n_local + b].tokens, 0, res[k * n_local + b].steps, res[k * n_local + b].avgLogProb,
                                               res[k * n_local + b].temperature, res[k * n_local + b].compressionRatio) for b in range(n_local)])
                for k in range(g)]
        return res, recs, nseg

    def run_steps(n):
        """n steps as the device batches of plan_batches(n, F, G), F batches in flight: worker f runs its batches on its own session /
        HIP stream (ctypes drops the GIL while the library runs); the per-step result records are gathered across the ranks by the main
        thread afterwards (wh_comm all-gather behind the C ABI; torch.distributed only when no communicator exists)."""
        plan = plan_batches(n, F, G)
        out = [[None] * len(p) for p in plan]
        dur = [[0.0] * len(p) for p in plan]
        errs = []
        starts, acc = [], 0                       # run-step number of every batch's first step (gather order: worker by worker, batch by batch)
        for p_ in plan:
            starts.append([])
            for g in p_:
                starts[-1].append(acc); acc += g

        def work(f):
            try:
                for i, g in enumerate(plan[f]):
                    a = time.perf_counter(); out[f][i] = hot_path(sessions[f], g, starts[f][i]); dur[f][i] = time.perf_counter() - a
            except BaseException as e:   # noqa: BLE001
                errs.append(e)
        if len(plan) == 1:
            work(0)
        else:
            ths = [threading.Thread(target=work, args=(f,)) for f in range(len(plan))]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        if errs:
            raise errs[0]
        gdev = dev if (world > 1 and args.dist_backend == "nccl") else None
        gathered = [parallel.gather_records(recs, per_rank_max, device=gdev, comm=comm) for o in out for _, per_step, _ in o for recs in per_step]
        assert len(gathered) == n
        step_dur = [(d / g, d) for p, ds in zip(plan, dur) for g, d in zip(p, ds) for _ in range(g)]      # (a batch's wall time shared by the steps it carries, the step's latency)
        return out[0][0][0], gathered[-1], step_dur

    def fence():
        for ss in sessions:
            ss.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    log(f"{model_name}: model + {F} session(s) of {slots} slots ready ({n_local} chunks per step on this rank, {G} step(s) per device batch); warmup x{warmup}")
    if warmup > 0:
        run_steps(max(warmup, F * G))    # every session captures its step graphs before the timed region ...
        for f, p in enumerate(plan_batches(steps, F, G)):      # ... including those of every shorter batch of the timed run's plan
            for g in sorted(set(p) - {G}):
                hot_path(sessions[f], g)
    fence()
    t0 = time.perf_counter()
    res, allrecs, durs = run_steps(steps)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert len(allrecs) == total and [r["chunk_index"] for r in allrecs] == list(range(total)), (len(allrecs), world, total)
    if rank == 0 and getattr(args, "dump_records", None):
        with open(args.dump_records, "w") as f:
            json.dump([{"chunk_index": r["chunk_index"], "tokens": r["tokens"], "steps": r["steps"], "avg_logprob": r["avg_logprob"]} for r in allrecs], f)
    log(f"{model_name}: timed region done: {elapsed:.3f} s for {steps} steps of {total} chunks ({n_local} on this rank), {F} batches in flight")
    dec_steps = [r.steps for r in res] if res else [min(args.sample_length, 224) - 1]
    audio_s = total * 30.0 * steps
    # median step: every step's own wall time (host PCM in -> segments out); with F steps in flight a step's latency is F x the
    # interval at which steps complete, so latency / F is the per-step cost the throughput implies
    Fe = len(plan_batches(steps, F, G))          # device batches in flight
    lat = [l for _, l in durs]
    durs = [d for d, _ in durs]
    out = {"value": audio_s / elapsed, "elapsed": elapsed, "audio_s": audio_s, "dec_steps": dec_steps[0], "B": B, "inflight": F,
           "cross_attention": (f"absorbed, {sess.crossAttentionSplits} key splits per slot, {sess.crossAttentionSlotsPerWorkgroup} slot(s) per workgroup "
                               f"({-(-slots // sess.crossAttentionSlotsPerWorkgroup) * sess.crossAttentionSplits} workgroups = CUs per launch)"
                               if sess.crossAttentionMode == 1 else "per-layer K / V rows"),
           "total": total, "n_local": n_local, "steps_per_batch": G, "slots": slots, "audio_sets": n_sets,
           "median_step_latency_ms": float(np.median(lat)) * 1e3, "median_ms_per_step": float(np.median(durs)) * 1e3 / Fe, "n_median": int(len(durs))}
    if rank == 0 and F > 1 and args.serial_reference:
        # single-stream reference on rank 0 only: local synchronisation, no collective (the other ranks are not here)
        for ss in sessions:
            ss.synchronize()
        s0 = time.perf_counter()
        for _ in range(2):
            hot_path(sess, G)
        sess.synchronize()
        out["serial_ms_per_step"] = (time.perf_counter() - s0) / (2 * G) * 1e3

    # ---- stage split (rank 0): mel + encoder milliseconds per chunk, decode tokens/s (one full device batch)
    if rank == 0:
        ts = []
        nb = slots
        for _ in range(3):
            sess.synchronize(); a0 = time.perf_counter()
            for k in range(G):
                for b, x in enumerate(audio[k % n_sets]):
                    sess.padOrTrim(x, k * n_local + b)
            sess.synchronize(); a = time.perf_counter()
            sess.logMelSpectrogram(nb); sess.synchronize(); b_ = time.perf_counter()
            sess.encodeFeatures(nb); sess.synchronize(); c = time.perf_counter()
            sess.prepareDecoderInputs(nb); sess.synchronize(); d_ = time.perf_counter()
            r2 = sess.decodeText(prompt, opts, batch=nb); e = time.perf_counter()
            ts.append((a - a0, b_ - a, c - b_, d_ - c, e - d_))
        med = np.median(np.array(ts), axis=0)
        out["stages"] = {"batch": nb, "pcm_upload_ms_per_chunk": med[0] * 1e3 / nb, "logmels_ms_per_chunk": med[1] * 1e3 / nb,
                         "encoder_ms_per_chunk": med[2] * 1e3 / nb, "encoder_ms_per_batch": med[2] * 1e3,
                         "cross_kv_ms_per_chunk": med[3] * 1e3 / nb, "decode_ms_per_chunk": med[4] * 1e3 / nb,
                         "decoder_steps": int(r2[0].steps), "tokens_per_s": nb * r2[0].steps / med[4],
                         "us_per_decoder_step": med[4] * 1e6 / max(r2[0].steps, 1)}
        log(f"{model_name}: stages {json.dumps({k: round(v, 3) for k, v in out['stages'].items()})}")
    if rank == 0 and want_roofline:
        rf = out["roofline"] = measure_kernels(sess, dims, slots, 16, dec_steps[0], model_name, steps_per_batch=G)
        rf["steps_per_device_batch"] = G
        ns = sess.crossAttentionSplits
        if ns:
            # the cross-attention takes slots x splits workgroups, one per CU: with several sessions in flight it is configured to leave CUs to
            # the other sessions' kernels.  The same kernel as a LONE session of this size runs it (the library's automatic key splits: slots x splits within one
            # round of the 256 CUs, one slot per workgroup) is timed beside it, alone on the GPU.
            spw = max(1, sess.crossAttentionSlotsPerWorkgroup)
            rf["slots_per_workgroup"] = spw
            rf["workgroups"] = -(-slots // spw) * ns
            rf["cu_share"] = round(min(1.0, rf["workgroups"] / 256.0), 3)
            ns_alone = api.Session.xabsAutoSplits(slots)
            if (ns != ns_alone or spw != 1) and whole_chip_leg:
                s4 = api.Session(model, slots, crossAttentionMode=1, crossAttentionSplits=ns_alone, crossAttentionSlotsPerWorkgroup=1)
                for k in range(G):
                    for b, x in enumerate(audio[k % n_sets]):
                        s4.padOrTrim(x, k * n_local + b)
                s4.logMelSpectrogram(slots); s4.encodeFeatures(slots); s4.prepareDecoderInputs(slots)
                s4.decodeText(prompt, opts, batch=slots)          # (wh_measure_kernels re-arms the slot state the last decodeText left)
                r4 = measure_kernels(s4, dims, slots, 16, dec_steps[0], model_name, steps_per_batch=G)
                s4_spw = max(1, s4.crossAttentionSlotsPerWorkgroup)
                s4.close()
                k4 = r4["kernels"]["dec_cross_attn"]
                rf["same_kernel_alone_on_the_whole_chip"] = {"splits": ns_alone, "slots_per_workgroup": s4_spw, "workgroups": -(-slots // s4_spw) * ns_alone, "avg_us": k4["avg_us"], "alg_per_launch": k4["alg_per_launch"],
                                                              "achieved": k4["achieved"], "unit": k4["unit"], "frac": k4["frac"]}
                # (the same figures as scalars: a reader that keeps only the flat keys of `roofline` still carries them)
                rf["whole_chip_splits"], rf["whole_chip_avg_us"], rf["whole_chip_achieved"], rf["whole_chip_frac"] = ns_alone, k4["avg_us"], k4["achieved"], k4["frac"]
        # whole step: algorithmic HBM bytes of every bandwidth-bound launch of one step / the step's share of the timed region
        hbm_bytes = sum(k["alg_per_launch"] * k["launches_per_step"] for k in rf["kernels"].values() if k["bound"] == "hbm")
        ms_step = elapsed / steps * 1e3
        rf["whole_step"] = {"hbm_bound_algorithmic_bytes": int(hbm_bytes), "ms_per_step": round(ms_step, 3),
                            "achieved": round(hbm_bytes / (ms_step * 1e-3) / 1e9, 1), "unit": "GB/s", "frac": round(hbm_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "note": "the matrix-core-bound encoder launches of the step run inside the same time and are not counted"}
        rf["whole_step_achieved"], rf["whole_step_frac"] = rf["whole_step"]["achieved"], rf["whole_step"]["frac"]
        log(f"{model_name}: roofline leg done")

    # ---- CPU baseline: the oracle (port of the same algorithm) on the host cores, bounded sample
    if rank == 0 and want_cpu:
        from oracle import decode as OD
        from oracle import mel as omel
        from oracle.model import OracleWhisper
        n_cpu_steps = 16
        # the per-token decoder step is a chain of matrix-vector products: beyond ~32 threads the fork/join cost of every
        # op outweighs the extra memory bandwidth (measured on the 256-thread box: 2.6 s/step with 128 threads)
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        om = OracleWhisper(dims, sd)
        ost, langs = OD.special_tokens_for_vocab(dims.n_vocab)
        oopts = OD.DecodingOptions(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None,
                                   noSpeechThreshold=None, temperatureFallbackCount=0, sampleLength=n_cpu_steps)
        c0 = time.perf_counter()
        omel_ = omel.log_mel_spectrogram(chunks[0], dims.n_mels).astype(np.float32)
        c1 = time.perf_counter()
        enc = om.encode(omel_)
        c2 = time.perf_counter()
        state = om.new_state(enc)
        c3 = time.perf_counter()
        ores = OD.decode_text(lambda t, p: state.step(t, p, want_alignment=False), OD.prefill_prompt(oopts, ost, dims.is_multilingual),
                              OD.GreedyTokenSampler(0.0, ost.endToken, oopts), oopts, ost, dims.is_multilingual, langs)
        c4 = time.perf_counter()
        per_step = (c4 - c3) / max(ores.steps, 1)
        total = (c3 - c0) + per_step * dec_steps[0]
        k = len(ores.tokens) - 1    # the oracle's result ends with the appended EOT
        same = (ores.tokens[:k] == res[0].tokens[:k]) if first == 0 else None
        out["cpu_baseline"] = {
            "value": round(30.0 / total, 4), "unit": "audio-sec/sec", "cores": torch.get_num_threads(), "cores_of": f"{torch.get_num_threads()} of {os.cpu_count()} logical CPUs", "kind": "port",
            "sample": f"one 30 s {model_name} chunk: mel {c1 - c0:.2f} s + encoder {c2 - c1:.2f} s + cross-K/V {c3 - c2:.2f} s measured "
                      f"in full, {ores.steps} decoder steps measured ({per_step * 1e3:.1f} ms/step) and extrapolated to {dec_steps[0]} steps "
                      f"(torch fp32, {torch.get_num_threads()} threads of {os.cpu_count()} logical CPUs)",
            "first_tokens_equal_gpu": same}
        log(f"{model_name}: cpu baseline done (first {k} tokens equal to the GPU's: {same})")
        if same is False:
            # end-to-end from PCM with an fp32 CPU encoder against the fp16-operand GPU encoder: a near-tie can flip a greedy choice;
            # anything else is a wrong GPU result and must not produce a benchmark line
            rec = []
            state2 = om.new_state(enc)
            ores2 = OD.decode_text(lambda t, p: state2.step(t, p, want_alignment=False), OD.prefill_prompt(oopts, ost, dims.is_multilingual),
                                   OD.GreedyTokenSampler(0.0, ost.endToken, oopts), oopts, ost, dims.is_multilingual, langs, record_logits=rec)
            kk = next(i for i, (a_, b_) in enumerate(zip(res[0].tokens[:k], ores2.tokens[:k])) if a_ != b_)
            start = OD.prefill_prompt(oopts, ost, dims.is_multilingual).index(ost.startOfTranscriptToken)
            filt = rec[start + kk - 1][3]
            top2 = np.sort(filt[np.isfinite(filt)])[-2:]
            gap = float(top2[1] - top2[0])
            if gap >= 5e-2:
                raise SystemExit(f"bench.py: GPU tokens differ from the CPU oracle at result index {kk} (oracle top-2 gap {gap:.3e}): {res[0].tokens[:k]} vs {ores.tokens[:k]}")
            out["cpu_baseline"]["first_tokens_equal_gpu"] = f"equal up to a near-tie at result index {kk} (oracle top-2 gap {gap:.2e})"
    for ss in sessions:
        ss.close()
    return out


def long_audio_config(args, local_rank):
    """BASELINE configs[4] at 1 GPU: whisper-large-v3, one 10 min audio cut into 30 s VAD chunks (WhisperKit.transcribe with
    chunkingStrategy .vad), temperature ladder forced once per window (log-prob threshold random weights always violate,
    temperatureFallbackCount 1 -> T = 0 then 0.2).  Two lines: the reference's behaviour (greedy T = 0 pass) and, labelled NO
    REFERENCE BEHAVIOUR, the beam = 5 variant configs[4] names - the reference's BeamSearchTokenSampler is a fatalError stub
    (Core/Text/TokenSampler.swift:254-290); wh_decode_text_beam follows openai/whisper's BeamSearchDecoder for the T = 0 pass."""
    from whisperkit_amd import api
    from whisperkit_amd.synth import synthetic_chunk
    model, dims, _ = get_model("large-v3", local_rank)
    audio = np.concatenate([synthetic_chunk(5000 + i) for i in range(20)]).astype(np.float32)
    kw = dict(firstTokenLogProbThreshold=None, compressionRatioThreshold=None, noSpeechThreshold=None, logProbThreshold=-1.0,
              temperatureFallbackCount=1, temperatureIncrementOnFallback=0.2, sampleLength=args.sample_length, seed=7)

    def run(slots, mode=None, splits=None, **extra):
        sess = api.Session(model, slots, crossAttentionMode=mode, crossAttentionSplits=splits)
        opts = api.DecodingOptions(**kw, **extra)
        sess.transcribeChunked(audio, opts)      # warm-up (graph capture)
        t0 = time.perf_counter()
        got = sess.transcribeChunked(audio, opts)
        el = time.perf_counter() - t0
        sess.close()
        return {"value": round(600.0 / el, 2), "unit": "audio-sec/sec", "seconds": round(el, 3), "chunks": len(got),
                "windows": sum(int(r.timings["total_decoding_windows"]) for _, r in got),
                "temperature_fallbacks": sum(int(r.timings["total_decoding_fallbacks"]) for _, r in got),
                "decoder_forward_passes": sum(int(r.timings["total_decoding_loops"]) for _, r in got)}
    out = run(20)
    # roofline of the greedy line: the kernels of one 20-slot decoder step, HIP-event timed; whole run = the algorithmic HBM bytes of
    # every decoder forward pass the transcription made / its wall time (encoder, host windowing and the fallback's second pass inside the same time)
    try:
        s20 = api.Session(model, 20)
        for b in range(20):
            s20.padOrTrim(audio[b * 480000:(b + 1) * 480000], b)
        s20.logMelSpectrogram(20); s20.encodeFeatures(20); s20.prepareDecoderInputs(20)
        o20 = api.DecodingOptions(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None, noSpeechThreshold=None,
                                  temperatureFallbackCount=0, sampleLength=args.sample_length)
        s20.decodeText(s20.prefillPrompt(o20), o20, batch=20)
        passes_per_slot = out["decoder_forward_passes"] / max(out["chunks"], 1)
        rf = measure_kernels(s20, dims, 20, 16, passes_per_slot, "large-v3")
        hbm = sum(k["alg_per_launch"] * k["launches_per_step"] for n_, k in rf["kernels"].items() if k["bound"] == "hbm" and (n_.startswith("dec_") or n_ == "sampler"))
        rf["whole_step"] = {"hbm_bound_algorithmic_bytes": int(hbm), "ms_per_step": round(out["seconds"] * 1e3, 3),
                            "achieved": round(hbm / out["seconds"] / 1e9, 1), "unit": "GB/s", "frac": round(hbm / out["seconds"] / 1e9 / HBM_PEAK_GBS, 4),
                            "note": "decoder launches of every forward pass of the 10 min transcription (20 slots, ladder forced once) / its wall time"}
        rf["whole_step_achieved"], rf["whole_step_frac"] = rf["whole_step"]["achieved"], rf["whole_step"]["frac"]
        out["roofline_full"] = rf
        s20.close()
    except Exception as e:   # noqa: BLE001 - a secondary figure must not take the headline down
        out["roofline_error"] = repr(e)
    out["note"] = ("10 min synthetic audio -> VADAudioChunker (30 s chunks, one device batch) -> decodeWithFallback with the ladder "
                   "forced once per window (T = 0 greedy, then 0.2 with the seeded top-5 sampler) -> segments")
    # (the library's choice at 100 slots: the absorbed cross-attention, which reads the audio's one encoder output with cacheable loads when
    # slots share it - round 5, profiles/r05b_beam5_cross_attention_mode_ab.jsonl: 241 audio-s/s against 233 with fp32 K / V rows, the first round-5 form of the rows)
    # (4 key splits: the beams of an audio stream ONE encoder output as L2 hits, more workgroups win - 240 audio-s/s against 228 with the 2 splits a 100-slot session
    # gets on its own, profiles/r06ah_beam5_key_splits.jsonl; include/whisperhip.h wh_session_create_tuned says so to beam-search callers)
    beam = run(100, splits=4, beamSize=5)
    beam["note"] = ("NO REFERENCE BEHAVIOUR: the same workload with beam = 5 for the T = 0 pass (20 windows x 5 beams = 100 decoder slots, "
                    "openai/whisper BeamSearchDecoder semantics, host-ranked candidates per step), then the same sampled fallback; the "
                    "reference's BeamSearchTokenSampler is a fatalError stub (Core/Text/TokenSampler.swift:254-290)")
    out["beam5_no_reference_behaviour"] = beam
    return out


def main():
    import faulthandler
    faulthandler.enable()
    if os.environ.get("WH_BENCH_WATCHDOG"):
        faulthandler.dump_traceback_later(int(os.environ["WH_BENCH_WATCHDOG"]), repeat=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--inflight", type=int, default=3, help="steps (batches of --batch chunks) in flight per GPU, each on its own session / HIP stream")
    ap.add_argument("--serial-reference", action=argparse.BooleanOptionalAction, default=True,
                    help="after the timed region, time the same workload with one step in flight (value_single_stream); --no-serial-reference skips it")
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--batch", type=int, default=64, help="30 s chunks per step (one decode batch = batch / 32 MFMA batch tiles): in total over the GPUs "
                    "with --scaling strong (BASELINE configs[3]: 64 chunks sharded across the GPUs), per GPU with --scaling weak")
    ap.add_argument("--device-batch", type=int, default=-1, help="slots of one device batch: the rank's shares of consecutive steps are packed into batches of "
                    "at most this many chunks (continuous batching; one session holds at most 256 windows).  -1 = automatic: 256 when a step has >= 64 chunks (four "
                    "64-chunk steps per batch at one GPU: the decoder's weight stream and launch chain are shared by eight batch tiles, a cross-attention workgroup streams two slots), "
                    "else --batch (one step per batch)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong", help="N > 1: strong = --batch chunks per step in total, block-partitioned "
                    "over the ranks (SURVEY 8d c4; the default); weak = --batch chunks per step and GPU")
    ap.add_argument("--gather", choices=["wh_comm", "torch"], default="wh_comm", help="N > 1: result-record all-gather through the C-ABI communicator "
                    "(RCCL, or the library's TCP transport in a --single-device rehearsal) or through torch.distributed")
    ap.add_argument("--cross-attention-splits", type=int, default=-1, help="key splits per slot of the absorbed cross-attention = the share of the CUs one "
                    "session's cross-attention takes (wh_session_create_tuned): -1 = 128 / slots (2 at 64 slots) when several device batches are in "
                    "flight (the other sessions' kernels keep half of the chip; profiles/r04ad_*, r04ae_*), the library's choice (4) for one")
    ap.add_argument("--cross-attention-slots-per-workgroup", type=int, default=-1, help="slots one workgroup of the absorbed cross-attention streams one after the other "
                    "(wh_session_options): -1 = ceil(slots x splits / 128) with several device batches in flight (the launch stays at <= 128 workgroups: 2 at 256 slots, 2 at the 224 slots of a 2-GPU run), 1 for one")
    ap.add_argument("--sample-length", type=int, default=224, help="DecodingOptions.sampleLength (224 -> 223 decoder steps)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank control flow on a box with fewer GPUs than ranks, together with --single-device)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--dump-records", default=None, help="rank 0 writes the gathered result records of the last timed step (chunk index, "
                    "token ids, steps) to this JSON file: the N-rank rehearsal compares them with a 1-rank run (tests/test_gpu_round4.py)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE {world}")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the whisperhip product path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    # ---- the communicator of the result gather, behind the C ABI (wh_comm_*): rank 0's id reaches the others through the process group
    # the launcher already set up (any out-of-band channel would do: a Swift host would use its own)
    comm, gather_kind = None, "none (1 GPU)"
    if world > 1:
        gather_kind = f"torch.distributed all_gather_into_tensor ({args.dist_backend})"
        if args.gather == "wh_comm":
            from whisperkit_amd import parallel
            try:
                def exchange(raw):
                    box = [raw]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
                rccl = args.dist_backend == "nccl" and not args.single_device      # RCCL refuses two ranks on one device: rehearsals use TCP
                port = int(os.environ.get("MASTER_PORT", "29500")) + 17
                comm = parallel.Comm(world, rank, transport="rccl" if rccl else "tcp", device=local_rank,
                                     tcp_address=f"{os.environ.get('MASTER_ADDR', '127.0.0.1')}:{port}", exchange_id=exchange if rccl else None)
                comm.barrier()
                gather_kind = "wh_comm_gather_records (C ABI): " + ("ncclAllGather over RCCL / xGMI" if rccl else "library TCP transport (one-GPU rehearsal)")
            except Exception as e:   # noqa: BLE001 - every rank takes the same branch only if the failure is symmetric; say what happened
                log(f"wh_comm unavailable ({e}); falling back to torch.distributed for the gather")
                comm = None
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and comm is not None:
                comm.close()
                comm = None
                gather_kind = f"torch.distributed all_gather_into_tensor ({args.dist_backend}); wh_comm failed on another rank"

    from whisperkit_amd import parallel as parallel_mod
    # what the result gather actually runs on: the world size and transport of the communicator wh_comm_create formed
    comm_world = int(comm.lib.wh_comm_world_size(comm.handle)) if comm is not None else (world if world > 1 else 1)
    if comm is not None:
        comm_transport = "rccl" if comm.lib.wh_comm_transport(comm.handle) == comm._L.COMM_RCCL else "tcp"
    else:
        comm_transport = "torch.distributed" if world > 1 else "none"
    dev_batch = args.device_batch if args.device_batch > 0 else (256 if args.batch >= 64 else args.batch)
    main_cfg = run_config(args, args.model, args.batch, args.inflight, args.steps, args.warmup, world, rank, local_rank, dev,
                          want_roofline=not args.no_roofline, want_cpu=(world == 1 and not args.no_cpu_baseline), comm=comm, scaling=args.scaling,
                          device_batch=dev_batch)
    other = {}
    headline = (args.model, args.batch) == ("large-v3", 64)
    extra = rank == 0 and world == 1 and not args.no_other_configs

    def brief_roofline(rf):
        """the dominant kernel's line + the whole-step HBM figure of a secondary configuration (every BASELINE config carries one)"""
        if not rf:
            return None
        keep = ("kernel", "cross_attention", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_us", "alg_per_launch", "share_of_step_time",
                "sum_kernel_ms_per_step", "workgroups", "cu_share", "whole_step_achieved", "whole_step_frac")
        r = {k: rf[k] for k in keep if k in rf}
        r["whole_step"] = rf.get("whole_step")
        # the three largest kernels of the step beside the dominant one
        top = sorted(rf["kernels"].items(), key=lambda kv: -(kv[1]["share_of_step"] or 0))[:4]
        r["top_kernels"] = {k: {"avg_us": v["avg_us"], "frac": v["frac"], "bound": v["bound"], "share_of_step": v["share_of_step"]} for k, v in top}
        return r

    def brief(o, n):
        return {"roofline": brief_roofline(o.get("roofline")),
                "value": round(o["value"], 2), "unit": "audio-sec/sec", "ms_per_step": round(o["elapsed"] / n * 1e3, 3),
                "median_ms_per_step": round(o["median_ms_per_step"], 3), "chunks_per_step": o["B"], "steps_in_flight": o["inflight"],
                "decoder_steps": o["dec_steps"], "cross_attention": o["cross_attention"], "stages": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in o["stages"].items()}}
    if extra and headline:
        o = run_config(args, "large-v3", 8, 3, 9, 3, 1, 0, local_rank, dev, want_roofline=True, want_cpu=False, whole_chip_leg=False)
        other["round-1 headline configuration: whisper-large-v3, 8 x 30 s chunks per step, 3 steps in flight, greedy, 1 GPU"] = brief(o, 9)
        saved = args.sample_length
        args.sample_length = 64
        o = run_config(args, "large-v3", 64, 3, 6, 3, 1, 0, local_rank, dev, want_roofline=True, want_cpu=False, whole_chip_leg=False)
        args.sample_length = saved
        other["whisper-large-v3, 64 chunks per step, 3 in flight, 64-token run (sampleLength 64, SURVEY 8d)"] = brief(o, 6)
        o = run_config(args, "large-v3", 128, 3, 3, 3, 1, 0, local_rank, dev, want_roofline=True, want_cpu=False, whole_chip_leg=False)
        other["whisper-large-v3, 128 chunks per step (four decoder batch tiles per session), 3 in flight = 384 chunks resident, greedy, 1 GPU"] = brief(o, 3)
        la = long_audio_config(args, local_rank)
        la["roofline"] = brief_roofline(la.pop("roofline_full", None))
        other["configs[4] whisper-large-v3, 10 min audio in 30 s VAD chunks, temperature ladder forced once, 1 GPU"] = la
        _MODELS.pop("large-v3")[0].close()
    if extra and (args.model, args.batch) != ("tiny.en", 1):
        o = run_config(args, "tiny.en", 1, 3, 9, 3, 1, 0, local_rank, dev, want_roofline=True, want_cpu=False, whole_chip_leg=False)
        other["configs[1] whisper-tiny.en, 1 x 30 s chunk per step, 3 in flight, greedy, 1 GPU"] = brief(o, 9)
    if extra and headline:
        o = run_config(args, "small", 8, 3, 6, 3, 1, 0, local_rank, dev, want_roofline=True, want_cpu=False, word_timestamps=True, whole_chip_leg=False)
        other["configs[2] whisper-small, 8 x 30 s chunks, greedy + word-timestamp alignment (DTW), 3 in flight, 1 GPU"] = brief(o, 6)
    if rank == 0:
        B = args.batch
        total, nl, G = main_cfg["total"], main_cfg["n_local"], main_cfg["steps_per_batch"]
        scaling = args.scaling if world > 1 else "strong"
        packed = (f"; each GPU packs its {nl}-chunk shares of {G} consecutive steps into one {main_cfg['slots']}-slot device batch" if G > 1 else "")
        out = {
            "metric": "audio-sec/sec (1/RTF), 30 s chunks: host PCM -> log-mel + encoder + greedy decode -> segments",
            "value": round(main_cfg["value"], 2), "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(main_cfg["elapsed"] / args.steps * 1e3, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"whisper-{args.model}, {total} x 30 s 16 kHz chunks per step" + (f" over {world} GPUs ({scaling} scaling, {nl} per GPU)" if world > 1 else "")
                                   + f", {main_cfg['inflight']} device batches in flight per GPU (= {main_cfg['slots'] * main_cfg['inflight']} chunks resident per GPU){packed}, "
                                   f"greedy (T=0), {main_cfg['dec_steps']} decoder steps/chunk, "
                                   "random-init weights, PCM handed over in host memory, segments built on the host",
                       "chunks_per_step": total, "chunks_per_gpu": nl, "steps_per_device_batch": G, "device_batch_slots": main_cfg["slots"],
                       "parallelism": f"chunk-dp{world}", "decoder_steps": main_cfg["dec_steps"], "cross_attention": main_cfg["cross_attention"],
                       "steps_in_flight": main_cfg["inflight"] * G, "device_batches_in_flight": main_cfg["inflight"], "result_gather": gather_kind,
                       "comm_world_size": comm_world, "comm_transport": comm_transport,
                       "chunk_ranges_per_rank": [list(parallel_mod.partition_chunks(total, world, r)) for r in range(world)],
                       "audio_sets": main_cfg["audio_sets"],
                       "serial_ms_per_step": round(main_cfg.get("serial_ms_per_step", 0.0), 3) or None,
                       "arith": "fp16 operands (decoder activations as f16 hi|lo pairs), fp32 accumulate/residual/softmax; mel fp32"},
            "rtf": round(main_cfg["elapsed"] / main_cfg["audio_s"], 6),
            "median_ms_per_step": round(main_cfg["median_ms_per_step"], 3), "median_step_latency_ms": round(main_cfg["median_step_latency_ms"], 3),
            "n_median": main_cfg["n_median"],
            "value_from_median_step": round(total * 30.0 / (main_cfg["median_ms_per_step"] * 1e-3), 2),
            "value_single_stream": (round(total * 30.0 / (main_cfg["serial_ms_per_step"] * 1e-3), 2)
                                    if main_cfg.get("serial_ms_per_step") else None),
            "encoder_ms_per_chunk": round(main_cfg["stages"]["encoder_ms_per_chunk"], 4),
            "stages": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in main_cfg["stages"].items()},
            "roofline": main_cfg.get("roofline"), "cpu_baseline": main_cfg.get("cpu_baseline"), "other_configs": other,
        }
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
