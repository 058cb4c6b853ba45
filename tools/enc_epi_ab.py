#!/usr/bin/env python3
"""Development tool: MD5 of the encoder output for several widths / slot counts, to be run once per WH_GEMM_EPI_MODE (0 direct epilogues,
1 LDS-staged, 2 direct with the bias fetched in one batch - csrc/gemm.hip).  The three modes must print identical lines: the staged
epilogues change which store instruction carries a value, not the value.

    for m in 0 1 2; do WH_GEMM_EPI_MODE=$m python tools/enc_epi_ab.py > gpurun_out/epi_md5_$m.json; done
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

from whisperkit_amd import api, weights  # noqa: E402
from whisperkit_amd.synth import synthetic_chunk  # noqa: E402

CASES = [("test-large-v3-l2", 64), ("test-large-v3-l2", 3), ("test-small-l2", 8), ("tiny.en", 8), ("base", 9), ("test-micro", 8), ("tiny.en", 1)]
if os.environ.get("WH_EPI_AB_QUICK"):        # tests/test_gpu_round5.py: seconds per mode (ragged M, a partial 256-column tile at width 384, every staged epilogue)
    CASES = [("test-large-v3-l2", 3), ("test-small-l2", 8), ("tiny.en", 8), ("test-micro", 8)]
out = {"mode": os.environ.get("WH_GEMM_EPI_MODE", "default")}
for name, slots in CASES:
    dims = weights.MODEL_DIMS[name]
    model = api.Model(dims, weights.synthetic_state_dict(dims, seed=3))
    sess = api.Session(model, slots)
    for b in range(slots):
        sess.padOrTrim(synthetic_chunk(900 + 7 * b), b)
    sess.logMelSpectrogram(slots); sess.encodeFeatures(slots)
    h = hashlib.md5()
    for b in range(slots):
        e = np.ascontiguousarray(sess.getEncoderOutput(b))
        assert np.isfinite(e).all()
        h.update(e.tobytes())
    out[f"{name}@{slots}"] = h.hexdigest()
    sess.close(); model.close()
print(json.dumps(out, sort_keys=True))
