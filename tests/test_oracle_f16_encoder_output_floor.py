"""Why the end-to-end logits bound on the realistic-statistics fixtures is not 1e-3 sigma (reading (3)) - shown on the CPU with the
oracle alone.  The decoder's input is a Float16 tensor (the reference's AudioEncoderOutput is an MLMultiArray of FloatType = Float16,
ArgmaxCore/FloatType.swift:9-13, Core/AudioEncoder.swift:50-63).  ONE Float16 rounding of the fp32 oracle's own encoder output - no GPU
arithmetic involved - already moves the oracle's logits by 1e-2 .. 2e-2 sigma when the cross-attention is sharp; the device's measured
end-to-end error (profiles/r05_realistic_errors.json, written by tests/test_gpu_realistic.py on the GPU, which since round 5 also measures the floor on the very positions it checks: 1.0 - 2.1 x) sits within 1.3 x of the floor measured here (asserted at 2 x: the maxima are taken over different position sets)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import mel as omel
from oracle.model import OracleWhisper
from realistic import realistic_state_dict
from whisperkit_amd import weights
from whisperkit_amd.synth import synthetic_chunk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,fixture,n_pos", [("small", "small-absorbed", 48), ("large-v3", "large-v3", 48)])
def test_one_float16_rounding_of_the_encoder_output_is_the_end_to_end_floor(name, fixture, n_pos):
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    dims = weights.MODEL_DIMS[name]
    om = OracleWhisper(dims, realistic_state_dict(dims, seed=0))
    enc = om.encode(omel.log_mel_spectrogram(synthetic_chunk(1234), dims.n_mels).astype(np.float32))
    rng = np.random.default_rng(3)
    toks = [50258, 50259, 50359] + [int(t) for t in rng.integers(0, 50000, n_pos - 3)]
    exact = om.new_state(enc).forward_full(toks, want_alignment=False)
    rounded = om.new_state(enc.astype(np.float16).astype(np.float32)).forward_full(toks, want_alignment=False)
    sigma = float(np.std(np.stack([exact[p] for p in range(0, n_pos, 4)])))
    floor = max(float(np.abs(exact[p] - rounded[p]).max()) for p in range(n_pos)) / sigma
    assert 3e-3 <= floor <= 5e-2, (name, floor)                 # one rounding of the decoder's input type alone breaks 1e-3 sigma (measured 9.6e-3 / 2.0e-2)
    assert all(int(np.argmax(exact[p])) == int(np.argmax(rounded[p])) for p in range(n_pos))
    dev = json.load(open(os.path.join(ROOT, "profiles", "r05_realistic_errors.json")))[fixture]["end_to_end"]
    assert abs(dev["logits_sigma"] - sigma) <= 0.1 * sigma       # the same fixture
    # (the device figure is a maximum over 2 slots x 97 positions, the floor over 48 positions of one chunk: measured 1.15e-2 vs 9.6e-3, 2.4e-2 vs 2.0e-2)
    assert dev["logits_rel_err"] <= 2.0 * floor, (name, dev["logits_rel_err"], floor)
