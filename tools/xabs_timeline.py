#!/usr/bin/env python3
"""GPU probe (development tool): per-tile timeline of xabs_attn from shader-clock stamps (tiles 10 and 11, waves 0 and 5 of the first 64 workgroups).
    WH_DBG=1 WH_NO_GRAPH=1 WH_XABS=1 python tools/xabs_timeline.py [model] [batch]
Stamps: 0 loop top, 1 after the S phase of tile i + 1, 2 after softmax(i), 3 after barrier C, 4 partials(i + 1) written, 5 after P V(i),
6 after the wait for tile i + 2, 7 after barrier D, 8 after the LDS-DMA requests."""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WH_DBG", "1"); os.environ.setdefault("WH_NO_GRAPH", "1"); os.environ.setdefault("WH_XABS", "1")
from whisperkit_amd import api, weights
from whisperkit_amd.synth import synthetic_chunk

name = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dims = weights.MODEL_DIMS[name]
model = api.Model(dims, weights.synthetic_state_dict(dims, seed=0))
sess = api.Session(model, B)
for b in range(B):
    sess.padOrTrim(synthetic_chunk(1234 + b), b)
sess.logMelSpectrogram(B); sess.encodeFeatures(B); sess.prepareDecoderInputs(B)
opts = api.DecodingOptions(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None,
                           noSpeechThreshold=None, temperatureFallbackCount=0, sampleLength=5)
sess.decodeText(sess.prefillPrompt(opts), opts, batch=B)
lib = sess.lib
lib.wh_debug_dump.restype = ctypes.c_int; lib.wh_debug_dump.argtypes = [ctypes.c_char_p]
assert lib.wh_debug_dump(b"/tmp/wh_dbg.bin") == 0
nk = lib.wh_kernel_kind_count()
names = [lib.wh_kernel_kind_name(k).decode() for k in range(nk)]
d = np.fromfile("/tmp/wh_dbg.bin", dtype=np.uint64).reshape(nk, 4096 * 8).astype(np.int64)[names.index("dec_cross_attn")]
raw = d[:64 * 2 * 2 * 16].reshape(64, 2, 2, 16)
t = raw[..., :9]                                                  # [workgroup][wave 0 / 5][tile 10 / 11][stamp]
ok = (t[..., 0] > 0).all(axis=(1, 2))
t = t[ok]
seg = np.diff(t, axis=-1)                                         # 8 segments per tile
lab = ["S(i+1)", "softmax(i)", "barrier C", "partials->LDS", "PV(i)", "wait tile i+2", "barrier D", "DMA issue"]
out = {"model": name, "B": B, "workgroups": int(ok.sum()), "tile_period_cycles": float(np.median(t[:, :, 1, 0] - t[:, :, 0, 0]))}
for w, wn in enumerate(("wave0", "wave5")):
    out[wn] = {lab[k]: [float(np.median(seg[:, w, :, k])), float(np.percentile(seg[:, w, :, k], 90))] for k in range(8)}
ph = raw[ok][:, :, 0, 9:14]                                       # entry, state known, loop entry, loop exit, partials stored
out["phases_cycles"] = {"entry->state": float(np.median(ph[..., 1] - ph[..., 0])), "state->loop": float(np.median(ph[..., 2] - ph[..., 1])),
                        "loop": float(np.median(ph[..., 3] - ph[..., 2])), "partials stored": float(np.median(ph[..., 4] - ph[..., 3])),
                        "first entry -> last exit (64 workgroups)": float(ph[..., 4].max() - ph[..., 0].min())}
print(json.dumps(out))
