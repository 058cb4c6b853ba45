"""Beam search, host side (no GPU): the C++ BeamSearchTokenSampler behind wh_beam_sampler_* against the CPU oracle's restatement of
openai/whisper's BeamSearchDecoder, step by step on synthetic log-probability rows with planted ties, duplicate beams and EOTs.
NO REFERENCE BEHAVIOUR: Core/Text/TokenSampler.swift:254-290 is fatalError; these tests pin the product to the oracle, and the
oracle to properties openai's algorithm has (beam 1 == greedy; scores are prefix sums; finished lists never exceed maxCandidates)."""
import numpy as np
import pytest

from oracle import decode as OD
from whisperkit_amd import api

EOT = 7


def _rows(rng, n_beams, V, k, tie=False, eot_boost=0.0):
    x = rng.normal(size=(n_beams, V)).astype(np.float32) * 3
    if tie:
        x[:, 3] = x[:, 11]                       # equal logits inside a row: top-k must prefer the lower id
        x[1 % n_beams] = x[0]                    # equal rows: equal candidate scores across beams with equal sums
    x[:, EOT] += eot_boost
    m = x.max(axis=1, keepdims=True)
    lp = (x - (m + np.log(np.exp(x - m).sum(axis=1, keepdims=True)))).astype(np.float32)
    order = np.argsort(-lp, axis=1, kind="stable")[:, :k]
    return lp, np.take_along_axis(lp, order, axis=1), order.astype(np.int32)


@pytest.mark.parametrize("beam,patience,seed", [(5, 1.0, 0), (5, 1.0, 1), (2, 1.0, 2), (3, 2.0, 3), (1, 1.0, 4), (4, 1.6, 5)])
def test_sampler_update_and_finalize_equal_the_oracle(beam, patience, seed):
    rng = np.random.default_rng(seed)
    V, P = 40, 3
    osamp = OD.BeamSearchTokenSampler(beam, EOT, patience)
    dsamp = api.BeamSearchTokenSampler(beam, EOT, patience)
    assert dsamp.maxCandidates == osamp.maxCandidates
    prompt = [1, 2, 3]
    obeams = [(list(prompt), [0.0] * P, 0.0) for _ in range(beam)]
    for step in range(12):
        n = len(obeams)
        lp, klp, ktok = _rows(rng, n, V, beam + 1, tie=(step % 3 == 1), eot_boost=(4.0 if step in (3, 4, 7) else 0.0))
        toks = np.array([b[0] for b in obeams], dtype=np.int32)
        lps = np.array([b[1] for b in obeams], dtype=np.float32)
        sums = np.array([b[2] for b in obeams], dtype=np.float32)
        nbeams, src, done = osamp.update(obeams, list(lp))
        nt, nl, ns, dsrc, ddone = dsamp.update(toks, lps, sums, klp, ktok)
        assert ddone == done, step
        assert [list(t) for t in nt] == [b[0] for b in nbeams], step
        assert list(dsrc) == src, step
        np.testing.assert_array_equal(ns, np.array([b[2] for b in nbeams], dtype=np.float32))
        np.testing.assert_array_equal(nl, np.array([b[1] for b in nbeams], dtype=np.float32).reshape(nl.shape))
        assert dsamp.finishedCount == len(osamp.finishedSequences) <= osamp.maxCandidates
        obeams = nbeams
        if done:
            break
    toks = np.array([b[0] for b in obeams], dtype=np.int32)
    lps = np.array([b[1] for b in obeams], dtype=np.float32)
    sums = np.array([b[2] for b in obeams], dtype=np.float32)
    cands = osamp.finalize(obeams)
    best = cands[osamp.rank(cands, P)]
    bt, bl, bs, nf = dsamp.finalize(toks, lps, sums, P)
    assert bt == best[0] and nf == len(cands)
    np.testing.assert_array_equal(np.float32(bl), np.float32(best[1]))
    assert np.float32(bs) == np.float32(best[2])
    # properties of the algorithm itself: a finished sequence ends in EOT, its score is the sum of its token log-probs
    for toks_, lps_, sm in cands:
        assert toks_[-1] == EOT and toks_[:P] == prompt
    assert best[0][-1] == EOT
    dsamp.close()


def test_first_step_collapses_duplicate_beams_like_the_python_dict():
    """All beams equal the prompt at the first expansion: openai keys candidates by the whole sequence, so the 5 x 6 candidates
    collapse to 6 and the new beams are 5 DIFFERENT tokens (not 5 copies of the best one)."""
    rng = np.random.default_rng(9)
    beam, V = 5, 30
    lp1, _, _ = _rows(rng, 1, V, beam + 1)
    lp = np.repeat(lp1, beam, axis=0)
    order = np.argsort(-lp, axis=1, kind="stable")[:, : beam + 1]
    d = api.BeamSearchTokenSampler(beam, EOT, 1.0)
    toks = np.tile(np.array([[1, 2]], dtype=np.int32), (beam, 1))
    nt, nl, ns, src, done = d.update(toks, np.zeros_like(toks, dtype=np.float32), np.zeros(beam, dtype=np.float32),
                                     np.take_along_axis(lp, order, axis=1), order.astype(np.int32))
    last = [int(t[-1]) for t in nt]
    expect = [int(t) for t in order[0] if t != EOT][:beam]
    assert last == expect and len(set(last)) == beam
    assert list(src) == [beam - 1] * beam          # a later assignment to an existing key overwrites its source (dict semantics)
    d.close()


def test_invalid_construction_is_an_error_not_a_crash():
    with pytest.raises(api.WhisperError):
        api.BeamSearchTokenSampler(0, EOT, 1.0)
    with pytest.raises(api.WhisperError):
        api.BeamSearchTokenSampler(2, EOT, 0.1)        # maxCandidates = Int(0.2) = 0: fatalError in the reference
    with pytest.raises(ValueError):
        OD.BeamSearchTokenSampler(2, EOT, 0.1)


def test_oracle_beam_of_one_is_the_greedy_decode():
    """Beam size 1 keeps the best non-EOT continuation and finishes on the first EOT that ranks first: the greedy path."""
    from oracle import mel as omel
    from oracle.model import OracleWhisper
    from whisperkit_amd import weights
    from whisperkit_amd.synth import synthetic_chunk
    dims = weights.MODEL_DIMS["test-micro"]
    sd = dict(weights.synthetic_state_dict(dims, seed=0))
    sd["decoder.token_embedding.weight"] = sd["decoder.token_embedding.weight"] * np.float32(32)
    om = OracleWhisper(dims, sd)
    st, langs = OD.special_tokens_for_vocab(dims.n_vocab)
    enc = om.encode(omel.log_mel_spectrogram(synthetic_chunk(5), dims.n_mels).astype(np.float32))
    oo = OD.DecodingOptions(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None, temperatureFallbackCount=0,
                            sampleLength=14)
    prompt = OD.prefill_prompt(oo, st, False)
    state = om.new_state(enc)
    g = OD.decode_text(lambda t, p: state.step(t, p), prompt, OD.GreedyTokenSampler(0.0, st.endToken, oo), oo, st, False, langs)
    b = OD.decode_text_beam(lambda: om.new_state(enc), prompt, 1, 1.0, oo, st, False, langs)
    assert b.tokens == g.tokens and b.steps == g.steps
    assert abs(b.avgLogProb - g.avgLogProb) < 1e-6
