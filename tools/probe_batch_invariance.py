#!/usr/bin/env python3
"""Development probe: where does a slot's result depend on the batch it is decoded in?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperkit_amd import api, weights
from whisperkit_amd.synth import synthetic_chunk

dims = weights.MODEL_DIMS["test-micro"]
model = api.Model(dims, weights.synthetic_state_dict(dims, seed=0))
for B in (3, 10):
    xs = [synthetic_chunk(500 + b) for b in range(B)]
    sb = api.Session(model, B)
    for b, x in enumerate(xs):
        sb.padOrTrim(x, b)
    sb.logMelSpectrogram(B); sb.encodeFeatures(B); sb.prepareDecoderInputs(B)
    s1 = api.Session(model, 1)
    b = B - 1
    s1.padOrTrim(xs[b]); s1.logMelSpectrogram(1); s1.encodeFeatures(1); s1.prepareDecoderInputs(1)
    print(f"B={B} slot {b}: mel diff {np.abs(sb.getMel(b) - s1.getMel(0)).max():.3e}  encoder diff {np.abs(sb.getEncoderOutput(b) - s1.getEncoderOutput(0)).max():.3e}")
    # decoder on identical encoder output: feed the single-session encoder output into the batch session
    enc = s1.getEncoderOutput(0)
    for bb in range(B):
        sb.setEncoderOutput(enc, bb)
    sb.prepareDecoderInputs(B)
    s1.setEncoderOutput(enc, 0); s1.prepareDecoderInputs(1)
    for pos, t in enumerate([50257, 50363, 400]):
        lb = sb.predictLogits([t] * B, [pos] * B)
        l1 = s1.predictLogits([t], [pos])
        print(f"   step {pos}: logits diff slot0 {np.abs(lb[0] - l1[0]).max():.3e} slot{b} {np.abs(lb[b] - l1[0]).max():.3e}")
