"""Beam search on the GPU (wh_decode_text_beam, beam_size in the transcribe flow) against the CPU oracle's restatement of
openai/whisper's BeamSearchDecoder.  NO REFERENCE BEHAVIOUR: the reference's BeamSearchTokenSampler (Core/Text/TokenSampler.swift:254-290)
is fatalError; BASELINE configs[4] asks for "beam=5" all the same.  The host-side ranking is pinned exactly in tests/test_beam_search.py;
these tests pin the device side (filtered log-softmax + top-k per beam, cache replication / rearrangement, the pre-fill hand-over)."""
import numpy as np
import pytest

from oracle import decode as OD
from oracle.model import OracleWhisper
from whisperkit_amd import api, weights
from whisperkit_amd.synth import synthetic_chunk

pytestmark = pytest.mark.gpu

NOFALLBACK = dict(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None, temperatureFallbackCount=0)
AUDIOS = (77, 78, 81)


def _peaky_with_eot(name, seed, boost):
    """Random weights give near-uniform next-token distributions (candidate scores a few 1e-4 apart: nothing a 1e-3-accurate device
    path could be compared on) and never emit EOT.  Scaling the tied token embedding by 32 makes the distributions decisive;
    sharpening the cross-attention (as in test_gpu_round2._audio_sensitive) makes them depend on the audio; copying a token the
    greedy decode emits (scaled by `boost`, rounded to fp16) into the EOT row makes EOT compete with it, so beams finish.
    Powers of two keep every weight exactly representable in fp16.  Margins of the oracle's rankings on these models: >= 5e-3."""
    dims = weights.MODEL_DIMS[name]
    sd = dict(weights.synthetic_state_dict(dims, seed=seed))
    sd["decoder.token_embedding.weight"] = sd["decoder.token_embedding.weight"] * np.float32(32)
    for i in range(dims.n_text_layer):
        for w, f in ((".cross_attn.out.weight", 32), (".cross_attn.query.weight", 16), (".cross_attn.key.weight", 16)):
            k = f"decoder.blocks.{i}" + w
            sd[k] = sd[k] * np.float32(f)
    st, langs = OD.special_tokens_for_vocab(dims.n_vocab)
    ml = dims.n_vocab >= 51865
    model = api.Model(dims, sd)
    sess = api.Session(model, 1)
    sess.padOrTrim(synthetic_chunk(77)); sess.logMelSpectrogram(1); sess.encodeFeatures(1); sess.prepareDecoderInputs(1)
    opts = api.DecodingOptions(**NOFALLBACK, sampleLength=40)
    prompt = sess.prefillPrompt(opts)
    g = sess.decodeText(prompt, opts)[0]
    k = g.tokens[len(prompt) + 6]
    E = sd["decoder.token_embedding.weight"].copy()
    E[st.endToken] = (E[k] * np.float32(boost)).astype(np.float16).astype(np.float32)
    sd["decoder.token_embedding.weight"] = E
    sess.close(); model.close()
    return dims, sd, api.Model(dims, sd), OracleWhisper(dims, sd), st, langs, ml


@pytest.fixture(scope="module", params=[("test-micro", 0, 0.9921875), ("test-micro-ml", 1, 1.0078125), ("test-micro-ml", 1, 0.9921875)],
                ids=["micro-eot-below", "micro-ml-eot-above", "micro-ml-eot-below"])
def peaky(request):
    return _peaky_with_eot(*request.param)


def _encode(sess, seeds):
    for b, sd_ in enumerate(seeds):
        sess.padOrTrim(synthetic_chunk(sd_), b)
    n = len(seeds)
    sess.logMelSpectrogram(n); sess.encodeFeatures(n); sess.prepareDecoderInputs(n)
    return [sess.getEncoderOutput(b).astype(np.float16).astype(np.float32) for b in range(n)]     # the cross-K/V GEMM reads fp16 operands


def _oopts(opts_kw):
    return OD.DecodingOptions(**opts_kw)


def _lps(r):
    return [list(d.values())[0] if isinstance(d, dict) else d for d in r.tokenLogProbs]


STATS = dict(compared=0, near_tie=0, early=0, differs_from_greedy=0)
DECISIVE = 3e-3      # an oracle ranking margin below this is a near-tie for a device path that is 1e-3-accurate on the logits


def _compare(got, ores, sampler, what):
    """Identical tokens - unless the oracle's own smallest ranking margin says a ranking was a near-tie (then nothing about this
    audio is comparable and it is counted; test_fixture_coverage bounds how often that may happen).  No skip, no xfail.
    The oracle keeps keys and values in Float16 like the device (and the reference) does: with the fixture's sharpened
    cross-attention the fp32-vs-fp16 storage difference alone moves a log-probability by 1e-2."""
    if sampler.minMargin < DECISIVE:
        STATS["near_tie"] += 1
        return False
    assert got.tokens == ores.tokens, f"{what}: tokens differ although the oracle's smallest ranking margin is {sampler.minMargin:.3e}"
    np.testing.assert_allclose(_lps(got), _lps(ores), atol=5e-3, err_msg=what)
    assert got.avgLogProb == pytest.approx(ores.avgLogProb, abs=2e-3), what
    assert got.steps == ores.steps, what
    assert got.temperature == 0.0 and got.compressionRatio == pytest.approx(ores.compressionRatio, abs=1e-6)
    STATS["compared"] += 1
    return True


@pytest.mark.parametrize("beam", [5, 2])
def test_decode_text_beam_vs_oracle(peaky, beam):
    dims, _, model, om, st, langs, ml = peaky
    kw = dict(**NOFALLBACK, sampleLength=36)
    opts = api.DecodingOptions(**kw)
    n = len(AUDIOS)
    sess = api.Session(model, n * beam)
    encs = _encode(sess, AUDIOS)
    prompt = sess.prefillPrompt(opts)
    greedy = sess.decodeText(prompt, opts, batch=n)
    sess.prepareDecoderInputs(n)
    got = sess.decodeTextBeam(prompt, opts, nAudio=n, beamSize=beam)
    for a in range(n):
        so = []
        ores = OD.decode_text_beam(lambda: om.new_state(encs[a], kvFloat16=True, crossFloat16=False), prompt, beam, 1.0, _oopts(kw), st, ml, langs, sampler_out=so)
        if _compare(got[a], ores, so[0], f"audio {AUDIOS[a]} beam {beam}"):
            STATS["early"] += len(ores.tokens) < len(prompt) + 34          # ended through finished (EOT) sequences, not the length cap
            STATS["differs_from_greedy"] += got[a].tokens != greedy[a].tokens
        assert got[a].tokens[-1] == st.endToken and got[a].tokens[0] == prompt[0]
    # exactness: every audio decodes the same alone (1 x beam slots) as in the batch (slots a * beam ...)
    s1 = api.Session(model, beam)
    for a in range(n):
        s1.padOrTrim(synthetic_chunk(AUDIOS[a])); s1.logMelSpectrogram(1); s1.encodeFeatures(1); s1.prepareDecoderInputs(1)
        alone = s1.decodeTextBeam(prompt, opts, nAudio=1, beamSize=beam)[0]
        assert alone.tokens == got[a].tokens and alone.tokenLogProbs == got[a].tokenLogProbs and alone.steps == got[a].steps


def test_beam_of_one_is_the_greedy_decode(peaky):
    """Beam size 1 follows the greedy path token for token.  One convention differs on purpose: a beam that ends on a sampled EOT
    keeps that EOT's log-probability (openai: it is part of the sequence's score), decodeText's finalize appends EOT with 0."""
    dims, _, model, om, st, langs, ml = peaky
    opts = api.DecodingOptions(**NOFALLBACK, sampleLength=24)
    sess = api.Session(model, 2)
    _encode(sess, AUDIOS[:2])
    prompt = sess.prefillPrompt(opts)
    g = sess.decodeText(prompt, opts, batch=2)
    sess.prepareDecoderInputs(2)
    b = sess.decodeTextBeam(prompt, opts, nAudio=2, beamSize=1)
    for x, y in zip(g, b):
        assert x.tokens == y.tokens and x.steps == y.steps
        np.testing.assert_allclose(_lps(x)[:-1], _lps(y)[:-1], atol=1e-5)
        assert _lps(x)[-1] == 0.0 and _lps(y)[-1] <= 0.0


def test_beam_with_patience_and_short_sample_length(peaky):
    """patience 2 -> maxCandidates 4 for beam 2 (more finished sequences are collected before stopping); sampleLength shorter than
    the prompt leaves the pre-filled prompt as the result, like decodeText."""
    dims, _, model, om, st, langs, ml = peaky
    kw = dict(**NOFALLBACK, sampleLength=36)
    opts = api.DecodingOptions(**kw)
    sess = api.Session(model, 4)
    encs = _encode(sess, AUDIOS[:2])
    prompt = sess.prefillPrompt(opts)
    got = sess.decodeTextBeam(prompt, opts, nAudio=2, beamSize=2, patience=2.0)
    for a in range(2):
        so = []
        ores = OD.decode_text_beam(lambda: om.new_state(encs[a], kvFloat16=True, crossFloat16=False), prompt, 2, 2.0, _oopts(kw), st, ml, langs, sampler_out=so)
        assert so[0].maxCandidates == 4
        _compare(got[a], ores, so[0], f"patience audio {a}")
    kw2 = dict(**NOFALLBACK, sampleLength=len(prompt) - 1)
    sess.prepareDecoderInputs(2)
    got = sess.decodeTextBeam(prompt, api.DecodingOptions(**kw2), nAudio=2, beamSize=2)
    for a in range(2):
        ores = OD.decode_text_beam(lambda: om.new_state(encs[a], kvFloat16=True, crossFloat16=False), prompt, 2, 1.0, _oopts(kw2), st, ml, langs)
        assert got[a].tokens == ores.tokens and got[a].steps == ores.steps


def test_beam_argument_errors(peaky):
    dims, _, model, om, st, langs, ml = peaky
    sess = api.Session(model, 4)
    _encode(sess, AUDIOS[:1])
    opts = api.DecodingOptions(**NOFALLBACK, sampleLength=8)
    prompt = sess.prefillPrompt(opts)
    with pytest.raises(api.WhisperError):
        sess.decodeTextBeam(prompt, opts, nAudio=1, beamSize=5)              # 5 slots needed, 4 there
    with pytest.raises(api.WhisperError):
        sess.decodeTextBeam(prompt, opts, nAudio=1, beamSize=2, patience=0.1)     # maxCandidates 0
    with pytest.raises(api.WhisperError):
        sess.decodeTextBeam(prompt, api.DecodingOptions(**NOFALLBACK, sampleLength=8, wordTimestamps=True), nAudio=1, beamSize=2)
    with pytest.raises(api.WhisperError):
        sess.transcribe([synthetic_chunk(3)], api.DecodingOptions(**NOFALLBACK, sampleLength=8, wordTimestamps=True, beamSize=2))
    assert sess.decodeTextBeam(prompt, opts, nAudio=1, beamSize=2)[0].tokens[-1] == st.endToken     # the session is still usable


def test_transcribe_with_beam_and_temperature_fallback_vs_oracle(peaky):
    """configs[4]'s shape in small: a 2-window audio, beam 5 at T = 0, thresholds that force the fallback to T = 0.2 (sampled, seeded)
    after the beam pass - the window's cross K/V, overwritten by the beam slots, is prepared again for the fallback decode."""
    dims, _, model, om, st, langs, ml = peaky
    audio = np.concatenate([synthetic_chunk(77), synthetic_chunk(78)[:200000]])
    for kw in (dict(**NOFALLBACK, sampleLength=30, beamSize=5),
               dict(sampleLength=30, beamSize=5, firstTokenLogProbThreshold=None, compressionRatioThreshold=None, logProbThreshold=0.0,
                    temperatureFallbackCount=1, temperatureIncrementOnFallback=0.2, seed=11)):
        sess = api.Session(model, 5)
        res = sess.transcribe([audio], api.DecodingOptions(**kw, detectLanguage=False))[0]
        s_enc = api.Session(model, 1)

        def encode_window(pcm):
            s_enc.padOrTrim(pcm); s_enc.logMelSpectrogram(1); s_enc.encodeFeatures(1)
            return s_enc.getEncoderOutput(0).astype(np.float16).astype(np.float32)

        def make_step(enc):
            state = om.new_state(enc, kvFloat16=True, crossFloat16=False)
            return lambda t, p: state.step(t, p)
        okw = dict(kw); seed = okw.pop("seed", 0)
        records = []
        ores = OD.transcribe_task_run(audio, OD.DecodingOptions(**okw, detectLanguage=False), st, ml, langs, dims.n_vocab, encode_window, make_step,
                                      seed=seed, records=records, make_state=lambda enc: om.new_state(enc, kvFloat16=True, crossFloat16=False))
        margins = [r["record"][0].minMargin for r in records if r["temperature"] == 0.0]
        if min(margins) < DECISIVE:
            STATS["near_tie"] += 1
            continue
        STATS["compared"] += 1
        assert res.seeks == ores.seeks
        assert res.tokens == ores.tokens, (res.tokens, ores.tokens)
        if kw.get("logProbThreshold") == 0.0:       # every window fell back once: beam pass, then the sampled pass at T = 0.2
            assert res.timings["total_decoding_fallbacks"] == len(res.seeks) and any(t > 0 for t in ores.temperatures)
        else:
            assert res.timings["total_decoding_fallbacks"] == 0


def test_fixture_coverage():
    """Runs last: how much of the above was decisive.  Most comparisons must have been made on decisive rankings, some decodes must
    have ended through finished (EOT) sequences, and beam search must have left the greedy path at least once."""
    print(STATS)
    if STATS["compared"] + STATS["near_tie"] == 0:
        return          # selected on its own (-k): there is nothing to take stock of
    assert STATS["compared"] >= 3 * STATS["near_tie"] and STATS["compared"] >= 12
    assert STATS["early"] >= 2 and STATS["differs_from_greedy"] >= 1
