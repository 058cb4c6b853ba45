#!/usr/bin/env python3
"""Development tool: BASELINE configs[4] (10 min audio, 20 VAD chunks, ladder forced once) greedy vs beam = 5 in both cross-attention modes.

    python tools/beam_ab.py "mode:splits" ...        (mode 0 K / V rows, 1 absorbed; splits 0 = the library's choice)

One JSON line per configuration: audio-s/s of the greedy line and of the beam = 5 line (NO REFERENCE BEHAVIOUR) and their ratio."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

from whisperkit_amd import api, weights  # noqa: E402
from whisperkit_amd.synth import synthetic_chunk  # noqa: E402

dims = weights.MODEL_DIMS["large-v3"]
model = api.Model(dims, weights.synthetic_state_dict(dims, seed=0))
audio = np.concatenate([synthetic_chunk(5000 + i) for i in range(20)]).astype(np.float32)
kw = dict(firstTokenLogProbThreshold=None, compressionRatioThreshold=None, noSpeechThreshold=None, logProbThreshold=-1.0,
          temperatureFallbackCount=1, temperatureIncrementOnFallback=0.2, sampleLength=224, seed=7)


def run(slots, mode, splits, **extra):
    sess = api.Session(model, slots, crossAttentionMode=mode, crossAttentionSplits=splits or None)
    opts = api.DecodingOptions(**kw, **extra)
    sess.transcribeChunked(audio, opts)
    t0 = time.perf_counter()
    got = sess.transcribeChunked(audio, opts)
    el = time.perf_counter() - t0
    toks = [r.tokens for _, r in got]
    sess.close()
    return 600.0 / el, toks


greedy, _ = run(20, None, 0)
print(json.dumps({"greedy_audio_s_per_s": round(greedy, 1)}), flush=True)
ref = None
for spec in sys.argv[1:]:
    mode, splits = (int(x) for x in spec.split(":"))
    v, toks = run(100, mode, splits, beamSize=5)
    if ref is None:
        ref = toks
    print(json.dumps({"beam5_mode": mode, "splits": splits, "audio_s_per_s": round(v, 1), "ratio_to_greedy": round(v / greedy, 3),
                      "tokens_equal_first_config": toks == ref}), flush=True)
