#!/usr/bin/env python3
"""Race screen: create a session and use it immediately, many times; every run must give the same transcription."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperkit_amd import api, weights
from whisperkit_amd.synth import synthetic_chunk

dims = weights.MODEL_DIMS["test-micro"]
model = api.Model(dims, weights.synthetic_state_dict(dims, seed=0))
audio = np.concatenate([synthetic_chunk(61), synthetic_chunk(62), synthetic_chunk(63)[:240000]])
kw = dict(sampleLength=12, firstTokenLogProbThreshold=None, compressionRatioThreshold=None, logProbThreshold=-1.0,
          temperatureFallbackCount=1, temperatureIncrementOnFallback=0.2, seed=5)
first, bad = None, 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for rep in range(n):
    junk = [api.Session(model, 3) for _ in range(2)]     # allocation / teardown churn around the session under test
    sess = api.Session(model, 1)
    res = sess.transcribe([audio], api.DecodingOptions(**kw))[0]
    sig = (res.seeks, res.tokens)
    if first is None:
        first = sig
    if sig != first:
        bad += 1
        print("run", rep, "differs:", res.seeks)
    del junk, sess
print(f"{n} runs, {bad} differing; seeks {first[0]}")
