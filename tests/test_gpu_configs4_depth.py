"""BASELINE configs[4] at full depth (VERDICT r03 "next round" 1c): whisper-large-v3, 32 + 32 layers, a 42.5 s audio = two windows of
TranscribeTask.run (Core/TranscribeTask.swift:57-296), the temperature ladder forced once per window (T = 0 -> 0.2, seeded sampling,
decodeWithFallback :316-411) - with the greedy T = 0 pass (the reference's behaviour) and with beam = 5 at T = 0 (no reference behaviour:
Core/Text/TokenSampler.swift:254-290 is a fatalError stub; openai/whisper's semantics, tests/test_gpu_beam.py).  Window seeks, every
result token, and the segments (tokens, times, temperature) must equal the oracle's restated loop.

Weights: tests/realistic.py (decisive next-token distributions - with the benign N(0, 0.02) weights the beam rankings are a few 1e-4
apart and nothing could be compared).  Stage isolation: the oracle decodes from the GPU's encoder output, stepped per token with a KV
cache (the reference's call pattern), with the device's storage types: Float16 self-attention cache; Float16 cross keys / values for
the K / V-row cross-attention, none for the absorbed path.  sampleLength 12 keeps the stepped 32-layer oracle within a minute.
"""
import numpy as np
import pytest
import torch

from oracle import decode as OD
from oracle.model import OracleWhisper
from realistic import realistic_state_dict
from whisperkit_amd import api, weights
from whisperkit_amd.synth import synthetic_chunk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rig():
    torch.set_num_threads(32)
    dims = weights.MODEL_DIMS["large-v3"]
    sd = realistic_state_dict(dims, seed=0)
    model = api.Model(dims, sd)
    om = OracleWhisper(dims, sd)
    st, langs = OD.special_tokens_for_vocab(dims.n_vocab)
    yield dims, model, om, st, langs
    model.close()


# The two windows' audio.  The beam cases need a fixture whose candidate rankings are decisive (minimum margin > 1e-2 in the oracle's own records): the margins follow the
# encoder output's Float16 roundings, so a change of the encoder's arithmetic at the last-bit level (round 6: the vectorised LayerNorm) can turn a seed pair into a near-tie -
# (77, 78) went from > 1e-2 to 3.3e-3 on the second window that way.  tests/tools/configs4_fixture_search.py lists the margins of candidate pairs
# (profiles/r06ao_configs4_fixture_search.jsonl: (79, 80) 0.018 / 0.020, (81, 82) 0.039 / 0.085, (83, 84) 0.068 / 0.052, (90, 91) 0.099 / 0.390; every other check passes for all of them).
AUDIO_SEEDS = (90, 91)


@pytest.mark.parametrize("beam,mode", [(0, 0), (0, 1), (5, 1), (5, 0)], ids=["greedy-kv-rows", "greedy-absorbed", "beam5-absorbed", "beam5-kv-rows"])
def test_configs4_two_windows_ladder_at_depth(rig, beam, mode):
    margins = run_case(rig, beam, mode, AUDIO_SEEDS)
    if beam:
        assert min(margins) > 1e-2, margins                 # the rankings compared were decisive


def run_case(rig, beam, mode, seeds):
    """every check of the scenario except the fixture-quality guard; returns the oracle's minimum ranking margins of the T = 0 beam passes (empty for greedy)"""
    dims, model, om, st, langs = rig
    audio = np.concatenate([synthetic_chunk(seeds[0]), synthetic_chunk(seeds[1])[:200000]])
    kw = dict(sampleLength=12, firstTokenLogProbThreshold=None, compressionRatioThreshold=None, logProbThreshold=1000.0,      # +1000: every T = 0 result is rejected (the reference's own way to force a fallback, UnitTests.swift:768-814)
              temperatureFallbackCount=1, temperatureIncrementOnFallback=0.2, seed=11, detectLanguage=False)
    if beam:
        kw["beamSize"] = beam
    sess = api.Session(model, max(1, beam), crossAttentionMode=mode)
    assert sess.crossAttentionMode == mode
    res = sess.transcribe([audio], api.DecodingOptions(**kw))[0]
    s_enc = api.Session(model, 1, crossAttentionMode=0)

    def encode_window(pcm):
        s_enc.padOrTrim(pcm); s_enc.logMelSpectrogram(1); s_enc.encodeFeatures(1)
        return s_enc.getEncoderOutput(0).astype(np.float16).astype(np.float32)
    templates = {}

    def make_state(enc):
        key = enc.ctypes.data
        if key not in templates:
            templates[key] = (enc, om.new_state(enc, kvFloat16=True, crossFloat16=False))       # (enc kept alive: the key is its address)
        return templates[key][1].fresh_like()

    def make_step(enc):
        state = make_state(enc)
        return lambda t, p: state.step(t, p, want_alignment=False)
    okw = dict(kw); seed = okw.pop("seed")
    records = []
    ores = OD.transcribe_task_run(audio, OD.DecodingOptions(**okw), st, True, langs, dims.n_vocab, encode_window, make_step, seed=seed,
                                  records=records, make_state=make_state)
    assert len(res.seeks) == 2 and res.seeks == ores.seeks
    info = ([(g.seek, g.temperature, round(g.avgLogprob, 4), g.tokens) for g in res.segments], [(g.seek, g.temperature, round(g.avgLogprob, 4), g.tokens) for g in ores.segments],
            ores.temperatures, res.timings["total_decoding_fallbacks"])
    assert res.tokens == ores.tokens, info
    n_fb = sum(1 for t in ores.temperatures if t > 0)                   # windows whose T = 0 result the thresholds rejected
    assert n_fb >= 1 and res.timings["total_decoding_fallbacks"] == n_fb, info
    if not beam:
        assert n_fb == 2                                                 # greedy: both windows go down the ladder once
    assert len(res.segments) == len(ores.segments)
    for a, b in zip(res.segments, ores.segments):
        assert (a.seek, a.tokens) == (b.seek, b.tokens)
        assert a.start == pytest.approx(b.start, abs=1e-5) and a.end == pytest.approx(b.end, abs=1e-5)
        assert a.temperature == pytest.approx(b.temperature) and a.avgLogprob == pytest.approx(b.avgLogprob, abs=5e-3)
    margins = [r["record"][0].minMargin for r in records if r["temperature"] == 0.0] if beam else []
    sess.close(); s_enc.close()
    return margins
