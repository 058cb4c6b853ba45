#!/usr/bin/env python3
"""GPU probe (development tool): in-kernel timeline of the decoder projection kernels from wall-clock stamps (100 MHz).
    WH_DBG=1 WH_NO_GRAPH=1 python tools/probe_dec32.py [model] [batch]
Stamps per workgroup: 0 entry, 1 first chunk requested (+ statistics in LDS), 2 first chunk's MFMAs done (= first data arrived),
3 all MFMAs done, 4 after the reduce barrier, 5 before the epilogue (after the split-K combine), 6 exit."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WH_DBG", "1"); os.environ.setdefault("WH_NO_GRAPH", "1")
from whisperkit_amd import api, weights
from whisperkit_amd.synth import synthetic_chunk

name = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dims = weights.MODEL_DIMS[name]
model = api.Model(dims, weights.synthetic_state_dict(dims, seed=0))
sess = api.Session(model, B)
for b in range(B):
    sess.padOrTrim(synthetic_chunk(1234 + b), b)
sess.logMelSpectrogram(B); sess.encodeFeatures(B); sess.prepareDecoderInputs(B)
opts = api.DecodingOptions(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None,
                           noSpeechThreshold=None, temperatureFallbackCount=0, sampleLength=9)
sess.decodeText(sess.prefillPrompt(opts), opts, batch=B)          # the probe buffer keeps the stamps of the LAST launch of every kind
lib = sess.lib
lib.wh_debug_dump.restype = ctypes.c_int; lib.wh_debug_dump.argtypes = [ctypes.c_char_p]
assert lib.wh_debug_dump(b"/tmp/wh_dbg.bin") == 0
nk = lib.wh_kernel_kind_count()
d = np.fromfile("/tmp/wh_dbg.bin", dtype=np.uint64).reshape(nk, 4096, 8).astype(np.int64)
for k in range(nk):
    nm = lib.wh_kernel_kind_name(k).decode()
    if not nm.startswith("dec_proj"):
        continue
    t = d[k]
    live = t[:, 0] > 0
    if not live.any():
        continue
    t = t[live] * 10.0 / 1000.0          # 100 MHz ticks -> us
    t0 = t[:, 0].min()
    fin = t[:, 6] > t[:, 4]              # finishers of THIS launch (or every workgroup when K is not split); older stamps are stale
    def stat(x): return f"{np.median(x):6.2f} (p10 {np.percentile(x, 10):5.2f}, p90 {np.percentile(x, 90):5.2f})"
    print(f"{nm}: {live.sum()} workgroups; entry spread {stat(t[:, 0] - t0)}; kernel span {max(t[fin][:, 6].max(), t[:, 4].max()) - t0:.2f} us")
    print(f"    entry->requested {stat(t[:, 1] - t[:, 0])}   requested->first data {stat(t[:, 2] - t[:, 1])}   ->all MFMAs {stat(t[:, 3] - t[:, 2])}")
    print(f"    reduce barrier {stat(t[:, 4] - t[:, 3])}   split-K combine (finishers) {stat(t[fin][:, 5] - t[fin][:, 4])}   epilogue {stat(t[fin][:, 6] - t[fin][:, 5])}")
