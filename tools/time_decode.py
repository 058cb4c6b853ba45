#!/usr/bin/env python3
"""GPU timing probe (development tool): decoder milliseconds per step and the per-kernel HIP-event table for one model at
several batch sizes.   python tools/time_decode.py large-v3 8,32 [inflight]"""
import ctypes, json, os, sys, threading, time
import numpy as np
if os.environ.get("WH_TOOL_NO_TORCH") != "1":
    import torch  # noqa: F401  (bench.py's process set-up: torch's HIP runtime is the one in the process; rocprofv3 bisect in profiles/r03j_*)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperkit_amd import api, weights
from whisperkit_amd.synth import synthetic_chunk

name = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
batches = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8").split(",")]
inflight = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dims = weights.MODEL_DIMS[name]
t0 = time.perf_counter()
model = api.Model(dims, weights.synthetic_state_dict(dims, seed=0))
print(f"# model {name} ready in {time.perf_counter() - t0:.1f}s", flush=True)
opts = api.DecodingOptions(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None,
                           noSpeechThreshold=None, temperatureFallbackCount=0)
for B in batches:
    sessions = [api.Session(model, B) for _ in range(inflight)]
    for s in sessions:
        for b in range(B):
            s.padOrTrim(synthetic_chunk(1234 + b), b)
        s.logMelSpectrogram(B); s.encodeFeatures(B); s.prepareDecoderInputs(B)
    prompt = sessions[0].prefillPrompt(opts)
    def warm():
        for s in sessions:
            s.decodeText(prompt, opts, batch=B)          # graph capture + warm-up
    if os.environ.get("WH_TOOL_MAIN_THREAD") == "1":
        warm()
    else:
        th = threading.Thread(target=warm); th.start(); th.join()     # bench.py captures on worker threads (profiles/r03j_*)
    def run(s, out, i):
        a = time.perf_counter(); r = s.decodeText(prompt, opts, batch=B); s.synchronize(); out[i] = (time.perf_counter() - a, r[0].steps)
    ts = []
    for rep in range(3):
        out = [None] * inflight
        ths = [threading.Thread(target=run, args=(s, out, i)) for i, s in enumerate(sessions)]
        a = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        ts.append(time.perf_counter() - a)
    wall = float(np.median(ts)); steps = out[0][1]
    rec = {"model": name, "B": B, "inflight": inflight, "steps": steps,
           "ms_per_step_wall": round(wall * 1e3 / steps, 4), "seq_steps_per_s": round(inflight * B * steps / wall, 1)}
    lib = sessions[0].lib
    nk = lib.wh_kernel_kind_count()
    avg = (ctypes.c_double * nk)(); cnt = (ctypes.c_int32 * nk)()
    api._check(lib.wh_measure_kernels(sessions[0].handle, B, 16, avg, cnt))
    rec["kernels_us"] = {lib.wh_kernel_kind_name(k).decode(): round(avg[k], 2) for k in range(nk) if cnt[k] and lib.wh_kernel_kind_name(k).decode().startswith(("dec_", "sampler"))}
    print(json.dumps(rec), flush=True)
    for s in sessions: s.close()
