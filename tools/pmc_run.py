#!/usr/bin/env python3
"""Short eager run of the hot path for counter collection (rocprofv3 --pmc serialises dispatches):
large-v3, 8 chunks: mel + encoder + cross-K/V + N decoder steps, no hipGraph.

    cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $REPO/gpurun_out/pmc_fetch -o fetch -- python $REPO/tools/pmc_run.py
"""
import os, sys
os.environ["WH_NO_GRAPH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperkit_amd import api, weights
from whisperkit_amd.synth import synthetic_chunk

name = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dims = weights.MODEL_DIMS[name]
model = api.Model(dims, weights.synthetic_state_dict(dims, seed=0))
sess = api.Session(model, B)
for b in range(B):
    sess.padOrTrim(synthetic_chunk(1234 + b), b)
opts = api.DecodingOptions(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None,
                           noSpeechThreshold=None, temperatureFallbackCount=0, sampleLength=steps + 1)
prompt = sess.prefillPrompt(opts)
for _ in range(2):
    sess.logMelSpectrogram(B); sess.encodeFeatures(B); sess.prepareDecoderInputs(B)
    r = sess.decodeText(prompt, opts, batch=B)
print("done", r[0].steps, "steps")
