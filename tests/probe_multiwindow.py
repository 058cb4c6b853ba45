#!/usr/bin/env python3
"""Development probe: GPU vs oracle on the multi-window fallback test input; where is the first differing token and how
close was the sampling decision?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import decode as OD
from oracle import mel as omel
from oracle.model import OracleWhisper
from whisperkit_amd import api, weights
from whisperkit_amd.synth import synthetic_chunk

dims = weights.MODEL_DIMS["test-micro"]
sd = weights.synthetic_state_dict(dims, seed=0)
model = api.Model(dims, sd); om = OracleWhisper(dims, sd)
audio = np.concatenate([synthetic_chunk(61), synthetic_chunk(62), synthetic_chunk(63)[:240000]])
kw = dict(sampleLength=12, firstTokenLogProbThreshold=None, compressionRatioThreshold=None, logProbThreshold=-1.0,
          temperatureFallbackCount=1, temperatureIncrementOnFallback=0.2, seed=5)
for rep in range(2):
    sess = api.Session(model, 1)
    res = sess.transcribe([audio], api.DecodingOptions(**kw))[0]
    print("gpu run", rep, res.seeks, [g.tokens for g in res.segments][:3])
st, langs = OD.special_tokens_for_vocab(dims.n_vocab)
okw = dict(kw); seed = okw.pop("seed")
ores = OD.transcribe_task_run(audio, OD.DecodingOptions(**okw), st, False, langs, dims.n_vocab,
                              lambda pcm: om.encode(omel.log_mel_spectrogram(pcm, dims.n_mels).astype(np.float32)),
                              lambda enc: (lambda state: (lambda t, p: state.step(t, p)))(om.new_state(enc)), seed=seed)
print("oracle", ores.seeks, [g.tokens for g in ores.segments][:3])
# window 0, T = 0.2 decode on both sides with the SAME (GPU) encoder output
sess = api.Session(model, 1)
sess.padOrTrim(audio[:480000]); sess.logMelSpectrogram(1); sess.encodeFeatures(1); sess.prepareDecoderInputs(1)
enc = sess.getEncoderOutput(0)
opts = api.DecodingOptions(**{**kw, "temperature": 0.2, "temperatureFallbackCount": 0})
prompt = sess.prefillPrompt(opts)
g = sess.decodeText(prompt, opts, seed=5 + 1)[0]
rec = []
state = om.new_state(enc.astype(np.float16).astype(np.float32))
oo = OD.DecodingOptions(**{**okw, "temperature": 0.2, "temperatureFallbackCount": 0})
o = OD.decode_text(lambda t, p: state.step(t, p), prompt, OD.GreedyTokenSampler(0.2, st.endToken, oo, seed=6), oo, st, False, langs, record_logits=rec)
print("same-encoder T=0.2: gpu", g.tokens, "oracle", o.tokens)
