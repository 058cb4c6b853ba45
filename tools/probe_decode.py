#!/usr/bin/env python3
"""GPU probe (development tool): launch floor of the decoder kernel chain and the in-kernel timeline of the GEMV kernels.
    WH_DBG=1 python tools/probe_decode.py [model] [batch]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
                           noSpeechThreshold=None, temperatureFallbackCount=0)
prompt = sess.prefillPrompt(opts)
L = dims.n_text_layer
for label, kw in (("live", {}), ("all slots inactive (early-exit kernels)", {"active": [0] * B})):
    sess.decodeText(prompt, opts, batch=B, **kw)
    t0 = time.perf_counter(); r = sess.decodeText(prompt, opts, batch=B, **kw); t1 = time.perf_counter()
    steps = r[0].steps if not kw else 16
    print(f"{label}: {(t1 - t0) * 1e3:.2f} ms, {steps} steps -> {(t1 - t0) * 1e6 / steps:.1f} us/step, {(t1 - t0) * 1e6 / steps / (8 * L + 2):.2f} us/kernel")
os.environ["WH_NO_GRAPH"] = "1"
if os.environ.get("WH_DBG") == "1":
    lib = sess.lib
    lib.wh_debug_dump.restype = ctypes.c_int; lib.wh_debug_dump.argtypes = [ctypes.c_char_p]
    sess.decodeText(prompt, api.DecodingOptions(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None,
                                               noSpeechThreshold=None, temperatureFallbackCount=0, sampleLength=9), batch=B)
    path = "/tmp/whdbg.bin"
    assert lib.wh_debug_dump(path.encode()) == 0
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, 4096, 8)
    names = [lib.wh_kernel_kind_name(k).decode() for k in range(lib.wh_kernel_kind_count())]
    for k, nm in enumerate(names):
        if nm == 'dec_cross_attn':
            t = a[k]; t = t[t[:, 0] > 0].astype(np.int64)
            t0 = t[:, 0].min(); r = (t - t0) * 10
            last = t[:, 6] == 1
            print(f'dec_cross_attn blocks {len(t)} (probe keeps <= 4096): start spread {r[:,0].max()} ns | med loads issued {np.median(r[:,1]):.0f}  K scored {np.median(r[:,2]):.0f}  PV done {np.median(r[:,3]):.0f}  ticket done {np.median(r[:,4]):.0f}  end {np.median(r[:,5]):.0f} | p10/p90 K scored {np.percentile(r[:,2],10):.0f}/{np.percentile(r[:,2],90):.0f} | last-arriver blocks end med {np.median(r[last][:,5]) if last.any() else -1:.0f} max {r[:,5].max()} ns')
            continue
        t = a[k]; used = t[:, 0] > 0
        if not used.any():
            continue
        t = t[used].astype(np.int64)
        w = lambda a_, b_: np.median((t[:, a_] - t[:, b_]) * 10)          # wall_clock64 = 100 MHz -> ns
        ln = nm in ("dec_gemv_qkv", "dec_gemv_cq", "dec_gemv_fc1", "dec_gemv_logits")
        ghz = np.median((t[:, 7] - t[:, 6]) / np.maximum((t[:, 5] - t[:, 1]) * 10, 1))
        span = (t[:, 5].max() - t[:, 1].min()) * 10
        print(f"{nm:18s} blocks {used.sum():4d} | issue->x arrived {w(0, 1) if ln else float('nan'):6.0f}  ->barrier {w(2, 0) if ln else w(2, 1):6.0f}  "
              f"fma(+weight wait) {w(3, 2):6.0f}  reduce+store {w(4, 3):5.0f} | block total {w(5, 1):6.0f} | first issue -> last end {span:6d} ns | clock ~{ghz:.2f} GHz")
