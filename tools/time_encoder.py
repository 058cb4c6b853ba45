#!/usr/bin/env python3
"""GPU timing probe (development tool): encoder + cross-K/V milliseconds per chunk and the per-kernel HIP-event table.
    python tools/time_encoder.py large-v3 8,32      (WH_NO_GEMM256=1 forces the small-tile kernel)"""
import ctypes, hashlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperkit_amd import api, weights
from whisperkit_amd.synth import synthetic_chunk

name = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
batches = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8").split(",")]
dims = weights.MODEL_DIMS[name]
sd = weights.synthetic_state_dict(dims, seed=0)
if os.environ.get("WH_ZERO_WEIGHTS") == "1":       # DVFS probe: the same kernels, launches and bytes on all-zero operands (every GEMM multiplies zeros): a kernel that
    sd = {k: np.zeros_like(v) for k, v in sd.items()}       # gets faster is power-limited, not schedule-limited (MI355X_MICROARCH.md "DVFS give-back")
model = api.Model(dims, sd)
for B in batches:
    s = api.Session(model, B)
    for b in range(B):
        s.padOrTrim(synthetic_chunk(1234 + b), b)
    s.logMelSpectrogram(B)
    ts, digests = [], set()
    for _ in range(6):
        s.synchronize(); a = time.perf_counter(); s.encodeFeatures(B); s.synchronize(); b_ = time.perf_counter(); s.prepareDecoderInputs(B); s.synchronize()
        ts.append((b_ - a, time.perf_counter() - b_))
        # race screen: the encoder output of the first / middle / last slot must be the same bits on every repetition (and across
        # kernel variants that keep the K order: compare the digests of separate runs)
        h = hashlib.md5()
        for b in sorted({0, B // 2, B - 1}):
            h.update(s.getEncoderOutput(b).tobytes())
        digests.add(h.hexdigest())
    med = np.median(np.array(ts[1:]), axis=0)
    lib = s.lib
    nk = lib.wh_kernel_kind_count()
    avg = (ctypes.c_double * nk)(); cnt = (ctypes.c_int32 * nk)()
    api._check(lib.wh_measure_kernels(s.handle, B, 0, avg, cnt))
    print(json.dumps({"model": name, "B": B, "encoder_ms_per_chunk": round(med[0] * 1e3 / B, 4),
                      "cross_kv_ms_per_chunk": round(med[1] * 1e3 / B, 4), "encoder_output_md5": sorted(digests),
                      "kernels_us": {lib.wh_kernel_kind_name(k).decode(): round(avg[k], 1) for k in range(nk) if cnt[k]}}), flush=True)
    s.close()
