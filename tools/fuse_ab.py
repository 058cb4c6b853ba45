#!/usr/bin/env python3
"""A/B of decoder launch variants (the WH_* knobs are read once per process, so each variant is its own process): prints an MD5 of the
greedy tokens + log-probs of every slot and the single-stream decode time.   WH_XATT_PERSIST=2 python tools/fuse_ab.py large-v3 64"""
import hashlib, json, os, sys, time
import numpy as np
import torch  # noqa: F401  (bench.py's process set-up, see tools/time_decode.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperkit_amd import api, weights
from whisperkit_amd.synth import synthetic_chunk

name, B = sys.argv[1], int(sys.argv[2])
dims = weights.MODEL_DIMS[name]
model = api.Model(dims, weights.synthetic_state_dict(dims, seed=0))
s = api.Session(model, B)
for b in range(B):
    s.padOrTrim(synthetic_chunk(1234 + b), b)
s.logMelSpectrogram(B); s.encodeFeatures(B); s.prepareDecoderInputs(B)
opts = api.DecodingOptions(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None, noSpeechThreshold=None,
                           temperatureFallbackCount=0)
prompt = s.prefillPrompt(opts)
res = s.decodeText(prompt, opts, batch=B)
ts = []
for _ in range(3):
    s.prepareDecoderInputs(B); s.synchronize()
    a = time.perf_counter(); res = s.decodeText(prompt, opts, batch=B); ts.append(time.perf_counter() - a)
h = hashlib.md5()
for r in res:
    h.update(np.asarray(r.tokens, np.int32).tobytes()); h.update(np.asarray(r.tokenLogProbs, np.float32).tobytes())
print(json.dumps({"model": name, "B": B, "knobs": {k: v for k, v in os.environ.items() if k.startswith("WH_")},
                  "ms_per_decoder_step": round(float(np.median(ts)) * 1e3 / res[0].steps, 4), "steps": res[0].steps, "md5_tokens_logprobs": h.hexdigest()}), flush=True)
