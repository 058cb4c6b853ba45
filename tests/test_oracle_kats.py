"""Pin the oracle's restated host logic against the reference's own known-answer tests.

Every test names the XCTest it ports (Tests/WhisperKitTests/UnitTests.swift under /root/reference).
CPU only; no HIP involved.
"""
import numpy as np
import pytest

from oracle import decode as D

INF = np.inf


def L(*v):
    # reference test helper MLMultiArray.logits([...]) builds Float16 logits (TestUtils.swift:113-121)
    return np.array(v, dtype=np.float16).astype(np.float32)


def st(**kw):
    # SpecialTokens.default(...) in TestUtils.swift:327-355: every id defaults to 0
    base = dict(endToken=0, englishToken=0, noSpeechToken=0, noTimestampsToken=0, specialTokenBegin=0,
                startOfPreviousToken=0, startOfTranscriptToken=0, timeTokenBegin=0, transcribeToken=0,
                translateToken=0, whitespaceToken=0)
    base.update(kw)
    return D.SpecialTokens(**base)


BASE = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7]


def eq(a, b):
    np.testing.assert_array_equal(np.asarray(a, np.float32), L(*b))


def test_suppress_tokens_filter():  # UnitTests.swift:1982-1997
    eq(D.SuppressTokensFilter([]).filterLogits(L(*BASE), []), BASE)
    eq(D.SuppressTokensFilter([0]).filterLogits(L(*BASE), []), [-INF] + BASE[1:])
    eq(D.SuppressTokensFilter([0, 2, 5, 6]).filterLogits(L(*BASE), []), [-INF, 0.2, -INF, 0.4, 0.5, -INF, -INF])


def test_suppress_blank_filter():  # UnitTests.swift:1999-2031
    eq(D.SuppressBlankFilter(st(), 0).filterLogits(L(*BASE), []), [-INF] + BASE[1:])
    f = D.SuppressBlankFilter(st(endToken=0, whitespaceToken=2), 0)
    eq(f.filterLogits(L(*BASE), []), [-INF, 0.2, -INF, 0.4, 0.5, 0.6, 0.7])
    f = D.SuppressBlankFilter(st(endToken=0, whitespaceToken=2), 3)
    eq(f.filterLogits(L(*BASE), [1, 2, 3]), [-INF, 0.2, -INF, 0.4, 0.5, 0.6, 0.7])
    f = D.SuppressBlankFilter(st(endToken=0, whitespaceToken=2), 5)
    eq(f.filterLogits(L(*BASE), [1, 2, 3]), BASE)


def test_language_logits_filter():  # UnitTests.swift:2033-2043
    eq(D.LanguageLogitsFilter([2, 4, 6], 7, 0).filterLogits(L(*BASE), []), [-INF, -INF, 0.3, -INF, 0.5, -INF, 0.7])
    eq(D.LanguageLogitsFilter([2, 4, 6], 7, 2).filterLogits(L(*BASE), [1]), BASE)


TS = [1.1, 5.2, 0.3, 0.4, 0.2, 0.1, 0.2, 0.1, 0.1]
TS_SPECIAL = dict(endToken=3, noTimestampsToken=2, timeTokenBegin=6, transcribeToken=4, translateToken=5)


def test_timestamp_rules_filter():  # UnitTests.swift:2045-2079
    f = D.TimestampRulesFilter(st(**TS_SPECIAL), 0, None, False)
    eq(f.filterLogits(L(*TS), [4]), [1.1, 5.2, -INF, 0.4, 0.2, 0.1, 0.2, 0.1, 0.1])
    eq(f.filterLogits(L(*TS), [0, 6, 7, 3]), [1.1, 5.2, -INF, 0.4, 0.2, 0.1, -INF, -INF, 0.1])
    eq(f.filterLogits(L(*TS), [0, 6, 7]), [1.1, 5.2, -INF, 0.4, 0.2, 0.1, -INF, -INF, -INF])
    eq(f.filterLogits(L(*TS), [0, 4, 7]), [-INF] * 7 + [0.1, 0.1])


def test_timestamp_rules_filter_multilingual():  # UnitTests.swift:2081-2115
    f = D.TimestampRulesFilter(st(**TS_SPECIAL), 0, None, True)
    eq(f.filterLogits(L(*TS), [0, 1, 2]), TS)
    eq(f.filterLogits(L(*TS), [0, 4, 6, 7, 3]), [1.1, 5.2, -INF, 0.4, 0.2, 0.1, -INF, -INF, 0.1])
    eq(f.filterLogits(L(*TS), [0, 5, 6, 7]), [1.1, 5.2, -INF, 0.4, 0.2, 0.1, -INF, -INF, -INF])
    eq(f.filterLogits(L(*TS), [0, 4, 0, 7]), [-INF] * 7 + [0.1, 0.1])


def test_create_logits_filters_composition():  # UnitTests.swift:3129-3247 (order + special-token filtering)
    s = D.special_tokens_for_vocab(51865)[0]
    o = D.DecodingOptions(suppressBlank=True, suppressTokens=[1, 2, s.specialTokenBegin + 5], withoutTimestamps=False)
    fs = D.create_logits_filters(o, 0, 4, s, True)
    assert [type(f).__name__ for f in fs] == ["SuppressBlankFilter", "SuppressTokensFilter", "TimestampRulesFilter"]
    assert fs[1].suppressTokens == [1, 2]          # ids >= specialTokenBegin are dropped (:877)
    assert fs[0].sampleBegin == 0 and fs[2].sampleBegin == 4
    assert D.create_logits_filters(D.DecodingOptions(withoutTimestamps=True), 0, 4, s, True) == []


def test_decoding_fallback_init():  # UnitTests.swift:816-878
    O = D.DecodingOptions
    f = D.decoding_fallback(O(compressionRatioThreshold=-1.0, logProbThreshold=-1.0, noSpeechThreshold=-1.0), True, 0, 0, -2.0)
    assert (f.fallbackReason, f.needsFallback) == ("firstTokenLogProbThreshold", True)
    f = D.decoding_fallback(O(compressionRatioThreshold=-1.0, logProbThreshold=-1.0, noSpeechThreshold=-1.0), False, 0, 0, -2.0)
    assert (f.fallbackReason, f.needsFallback) == ("silence", False)
    f = D.decoding_fallback(O(compressionRatioThreshold=-1.0, logProbThreshold=-1.0, noSpeechThreshold=0.0), False, 0, 0, -2.0)
    assert (f.fallbackReason, f.needsFallback) == ("compressionRatioThreshold", True)
    f = D.decoding_fallback(O(compressionRatioThreshold=0.0, logProbThreshold=-1.0, noSpeechThreshold=0.0), False, 0, 0, -2.0)
    assert (f.fallbackReason, f.needsFallback) == ("logProbThreshold", True)
    assert D.decoding_fallback(O(compressionRatioThreshold=0.0, logProbThreshold=0.0, noSpeechThreshold=0.0), False, 0, 0, 0) is None


def test_compression_ratio_ordering():  # UnitTests.swift:695-705
    u = D.compression_ratio(list(range(1, 11)))
    r = D.compression_ratio([1] * 10)
    rl = D.compression_ratio([1] * 20)
    assert u < r < rl
    assert D.compression_ratio([]) == float("inf")


def test_dtw_simple_matrix():  # UnitTests.swift:2337-2367
    ti, tj = D.dynamic_time_warping(np.array([[1.0, 1.0, 1.0], [5.0, 2.0, 1.0], [1.0, 5.0, 2.0]]))
    assert ti == [0, 1, 1, 2, 2]
    assert tj == [0, 0, 1, 1, 2]


def test_dtw_large_matrix_properties():  # UnitTests.swift:2369-2418 (smaller rows: pure-python DP)
    rng = np.random.default_rng(0)
    n, m = 48, 1500
    ti, tj = D.dynamic_time_warping(rng.random((n, m)).astype(np.float16))
    assert (ti[0], tj[0]) == (0, 0) and (ti[-1], tj[-1]) == (n - 1, m - 1)
    for a in range(1, len(ti)):
        dr, dc = ti[a] - ti[a - 1], tj[a] - tj[a - 1]
        assert dr in (0, 1) and dc in (0, 1) and (dr == 1 or dc == 1)


def test_word_probability_is_exp_mean_logprob():  # UnitTests.swift:2420-2482 (findAlignment probability rule)
    n_tok = 4
    align = np.zeros((n_tok, 20), np.float32)
    for i in range(n_tok):
        align[i, i * 5:(i + 1) * 5] = 1.0
    lps = [-0.1, -0.2, -0.3, -0.4]
    split = lambda ids: (["a", "b"], [ids[:2], ids[2:]])
    w = D.find_alignment([10, 11, 12, 13], align, lps, split)
    assert len(w) == 2
    assert w[0].probability == pytest.approx(np.exp(np.float32(-0.15)), rel=1e-6)
    assert w[1].probability == pytest.approx(np.exp(np.float32(-0.35)), rel=1e-6)
    assert w[0].start == 0.0 and w[1].end == pytest.approx(19 * 0.02)


def test_vad_on_jfk(jfk_pcm):  # UnitTests.swift:2119-2189 testVoiceActivity
    vad = D.EnergyVAD()
    assert vad.voiceActivity([]) == []
    va = vad.voiceActivity(jfk_pcm)
    assert vad.findLongestSilence(va) == (43, 54)
    assert vad.voiceActivityIndexToAudioSampleIndex(43) == 68800
    assert vad.voiceActivityIndexToAudioSampleIndex(54) == 86400
    assert vad.voiceActivityIndexToSeconds(43) == pytest.approx(4.3)
    assert vad.voiceActivityIndexToSeconds(54) == pytest.approx(5.4)
    v = D.EnergyVAD(frameLengthSamples=320)
    z, o = np.zeros(1600, np.float32), np.ones(1600, np.float32)
    assert v.calculateActiveChunks(np.zeros(0, np.float32)) == []
    assert v.calculateActiveChunks(z) == []
    assert v.calculateActiveChunks(o) == [(0, 1600)]
    assert v.calculateActiveChunks(np.concatenate([z, o])) == [(1600, 3200)]
    assert v.calculateActiveChunks(np.ones(1601, np.float32)) == [(0, 1601)]
    assert v.calculateActiveChunks(np.ones(1599, np.float32)) == [(0, 1599)]
    assert v.calculateActiveChunks(np.concatenate([np.ones(1599, np.float32), z])) == [(0, 1600)]
    vo = D.EnergyVAD(frameLengthSamples=320, frameOverlapSamples=80)
    assert vo.calculateActiveChunks(np.concatenate([z, o])) == [(1280, 3200)]
    big = D.EnergyVAD(frameLength=0.2, frameOverlap=0.1)
    clips = big.calculateNonSilentSeekClips(jfk_pcm)
    assert [c[0] for c in clips] == [3200, 51200, 83200, 128000, 169600]
    assert [c[1] for c in clips] == [35200, 70400, 121600, 166400, 176000]
    np.testing.assert_allclose(big.voiceActivityClipTimestamps(jfk_pcm), [0.2, 2.2, 3.2, 4.4, 5.2, 7.6, 8.0, 10.4, 10.6, 11.0], rtol=1e-6)


def test_find_longest_silence():  # UnitTests.swift:2210-2241
    f = D.EnergyVAD.findLongestSilence
    T, F = True, False
    for v in ([], [T], [T, T], [T] * 5):
        assert f(v) is None
    assert f([F]) == (0, 1)
    assert f([F, F]) == (0, 2)
    assert f([T, F, F]) == (1, 3)
    assert f([F, F, T]) == (0, 2)
    assert f([T, F, F, T]) == (1, 3)
    assert f([F, F, T, T, T, F, T, F, F, F, F, T, T]) == (7, 11)


def test_vad_chunker_single_chunk(jfk_pcm):  # UnitTests.swift:2243-2262 (first half; ted_60.m4a is not decodable here)
    chunks = D.vad_chunk_all(jfk_pcm, 480000)
    assert len(chunks) == 1 and chunks[0][0] == 0 and len(chunks[0][1]) == 176000
    long = np.concatenate([jfk_pcm] * 4)                     # 44 s: must split on a silence in the 2nd half
    chunks = D.vad_chunk_all(long, 480000)
    assert len(chunks) >= 2
    assert chunks[0][0] == 0 and all(len(c[1]) <= 480000 for c in chunks)
    assert sum(len(c[1]) for c in chunks) <= len(long)
    for (s0, a0), (s1, _) in zip(chunks, chunks[1:]):
        assert s0 + len(a0) == s1


def test_prepare_seek_clips():  # Extensions+Internal.swift:112-130
    assert D.DecodingOptions().prepareSeekClips(1000) == [(0, 1000)]
    assert D.DecodingOptions(clipTimestamps=[1.0]).prepareSeekClips(48000) == [(16000, 48000)]
    assert D.DecodingOptions(clipTimestamps=[0.5, 1.0, 2.0]).prepareSeekClips(48000) == [(8000, 16000), (32000, 48000)]


def test_special_tokens_match_reference_defaults():  # Core/Models.swift:1309-1322
    s, langs = D.special_tokens_for_vocab(51865)
    assert s == D.SpecialTokens()
    assert len(langs) == 99 and langs[0] == 50259
    s3, langs3 = D.special_tokens_for_vocab(51866)
    assert s3.timeTokenBegin == 50365 and len(langs3) == 100
    se, _ = D.special_tokens_for_vocab(51864)
    assert (se.endToken, se.startOfTranscriptToken, se.noTimestampsToken, se.timeTokenBegin) == (50256, 50257, 50362, 50363)


def test_segments_from_reference_token_sequence():
    # JFK tiny tokens pinned in UnitTests.swift:1289-1297: <|0.00|> ... <|10.50|>; then EOT appended by finalize
    toks = [50364, 400, 370, 452, 7177, 6280, 1029, 406, 437, 428, 1941, 393, 360, 337, 291, 1029, 437, 291, 393,
            360, 337, 428, 1941, 13, 50889]
    s = D.SpecialTokens()
    full = [s.startOfTranscriptToken, 50259, 50359] + toks + [s.endToken]
    res = D.DecodingResult("en", full, [{t: -0.1} for t in full], -0.1, 0.0, 0.0, 1.0, None)
    seek, segs = D.find_seek_point_and_segments(res, D.DecodingOptions(), 0, 0, 176000, s)
    # single timestamp ending ([..., text, ts, EOT]) and no consecutive pair -> one segment, seek += segmentSize
    assert len(segs) == 1 and seek == 176000
    assert segs[0].start == 0.0 and segs[0].end == pytest.approx(10.5)
    # with a consecutive timestamp pair the window is split and seek moves to the last timestamp
    full2 = [s.startOfTranscriptToken, 50364, 400, 50464, 50464, 370, 50564, s.endToken]
    res2 = D.DecodingResult("en", full2, [{t: -0.1} for t in full2], -0.1, 0.0, 0.0, 1.0, None)
    seek2, segs2 = D.find_seek_point_and_segments(res2, D.DecodingOptions(), 3, 16000, 480000, s)
    assert [g.id for g in segs2] == [3, 4]
    assert segs2[0].tokens == [s.startOfTranscriptToken, 50364, 400, 50464]
    assert (segs2[0].start, segs2[0].end) == (pytest.approx(1.0), pytest.approx(3.0))
    assert (segs2[1].start, segs2[1].end) == (pytest.approx(3.0), pytest.approx(5.0))
    assert seek2 == 16000 + int(np.float32(4.0) * 16000)


def test_update_segment_timings_and_vad_chunked_flow():
    """TranscriptionUtilities.updateSegmentTimings (Utilities/TranscriptionUtilities.swift:55-69) and the .vad branch of
    WhisperKit.transcribe(audioArray:) (Core/WhisperKit.swift:878-906) with a stub transcriber."""
    from oracle import decode as OD
    import numpy as np
    mk = lambda **kw: OD.TranscriptionSegment(text="", temperature=0.0, avgLogprob=0.0, compressionRatio=1.0, noSpeechProb=0.0, **kw)
    seg = mk(id=0, seek=160, start=1.0, end=2.5, tokens=[1, 2], tokenLogProbs=[{1: -0.1}, {2: -0.2}],
             words=[OD.WordTiming("a", [1], 1.0, 1.5, 0.9)])
    up = OD.update_segment_timings(seg, 22.9)
    assert up.seek == 160 + int(np.float32(22.9) * np.float32(16000)) and up.start == pytest.approx(23.9) and up.end == pytest.approx(25.4)
    assert up.words[0].start == pytest.approx(23.9) and up.words[0].end == pytest.approx(24.4)
    assert seg.seek == 160 and seg.start == 1.0 and seg.words[0].start == 1.0          # input untouched

    calls = []

    def one(samples, opts):
        calls.append((len(samples), list(opts.clipTimestamps)))
        return OD.TranscriptionResult([mk(id=0, seek=0, start=0.5, end=1.0, tokens=[7], tokenLogProbs=[{7: 0.0}])], [7], "en")
    rng = np.random.default_rng(0)
    loud = lambda n: (0.5 * rng.standard_normal(n)).astype(np.float32)
    audio = np.concatenate([loud(400000), np.zeros(32000, np.float32), loud(300000), np.zeros(32000, np.float32), loud(200000)])
    out = OD.transcribe_vad_chunked(audio, OD.DecodingOptions(clipTimestamps=[0.0]), one)
    chunks = OD.vad_chunk_all(audio, options=OD.DecodingOptions(clipTimestamps=[0.0]))
    assert len(out) == len(chunks) == len(calls) >= 2
    assert all(ct == [] for _, ct in calls)              # clipTimestamps reset for the chunks (:889-891)
    for (t, r), (off, samples) in zip(out, chunks):
        assert t == pytest.approx(off / 16000) and r.segments[0].start == pytest.approx(0.5 + off / 16000, abs=1e-4)
        assert r.segments[0].seek == int(np.float32(np.float32(off) / np.float32(16000)) * np.float32(16000))
    assert sum(len(c[1]) for c in chunks) == len(audio) or chunks[-1][0] + len(chunks[-1][1]) <= len(audio)
    short = OD.transcribe_vad_chunked(audio[:1000], None, one)
    assert len(short) == 1 and short[0][0] == 0.0


# ----------------------------------------------------------------------------- mergePunctuations KATs (UnitTests.swift:2484-2667)
def _wt(word, tokens, start, end, prob=1.0):
    from oracle import decode as OD
    return OD.WordTiming(word, list(tokens), float(start), float(end), float(prob))


def _assert_words(got, expected):
    assert len(got) == len(expected)
    for g, e in zip(got, expected):
        assert (g.word, g.tokens, g.start, g.end, g.probability) == (e.word, e.tokens, e.start, e.end, e.probability)


def test_merge_punctuations_empty_and_english():  # UnitTests.swift:2484-2539
    from oracle import decode as OD
    assert OD.merge_punctuations([]) == []
    words = [_wt("<|0.00|>", [50364], 0, 1), _wt(" Hello", [2425], 1, 2), _wt(",", [11], 2, 3), _wt(" world", [1002], 3, 4),
             _wt("!", [0], 4, 5), _wt("<|1.00|>", [50414], 5, 6), _wt("<|1.00|>", [50414], 6, 7), _wt(" This", [639], 7, 8),
             _wt(" is", [307], 8, 9), _wt(" a", [257], 9, 10), _wt(" test", [220, 31636], 10, 11), _wt(",", [11], 11, 12),
             _wt(" isn't", [1943, 380], 12, 13), _wt(" it", [309], 13, 14), _wt("?", [30], 14, 15), _wt("<|endoftext|>", [50257], 15, 16)]
    got = OD.merge_punctuations(words, prepended="\"'“¿([{-", appended="\"'.。,，!！?？:：”)]}、")
    _assert_words(got, [_wt("<|0.00|>", [50364], 0, 1), _wt(" Hello,", [2425, 11], 1, 2), _wt(" world!", [1002, 0], 3, 4),
                        _wt("<|1.00|>", [50414], 5, 6), _wt("<|1.00|>", [50414], 6, 7), _wt(" This", [639], 7, 8), _wt(" is", [307], 8, 9),
                        _wt(" a", [257], 9, 10), _wt(" test,", [220, 31636, 11], 10, 11), _wt(" isn't", [1943, 380], 12, 13),
                        _wt(" it?", [309, 30], 13, 14), _wt("<|endoftext|>", [50257], 15, 16)])


def test_merge_punctuations_spanish():  # UnitTests.swift:2541-2587
    from oracle import decode as OD
    words = [_wt("<|notimestamps|>", [50363], 0, 1), _wt(" ¡", [24364], 0, 1), _wt("Hola", [48529], 1, 2), _wt(" Mundo", [376, 6043], 2, 3),
             _wt("!", [0], 3, 4), _wt(" Esta", [20547], 4, 5), _wt(" es", [785], 5, 6), _wt(" una", [2002], 6, 7),
             _wt(" prueba", [48241], 7, 8), _wt(",", [11], 8, 9), _wt(" ¿", [3841], 9, 10), _wt("no", [1771], 10, 11), _wt("?", [30], 11, 12),
             _wt("<|endoftext|>", [50257], 12, 13)]
    _assert_words(OD.merge_punctuations(words),
                  [_wt("<|notimestamps|>", [50363], 0, 1), _wt(" ¡Hola", [24364, 48529], 1, 2), _wt(" Mundo!", [376, 6043, 0], 2, 3),
                   _wt(" Esta", [20547], 4, 5), _wt(" es", [785], 5, 6), _wt(" una", [2002], 6, 7), _wt(" prueba,", [48241, 11], 7, 8),
                   _wt(" ¿no?", [3841, 1771, 30], 10, 11), _wt("<|endoftext|>", [50257], 12, 13)])


def test_merge_punctuations_spanish_start_with_prepend():  # UnitTests.swift:2589-2622
    from oracle import decode as OD
    words = [_wt(" ¿", [1201], 0, 1), _wt("Que", [1202], 1, 2, 0.9), _wt(" pasa", [1203], 2, 3), _wt(" mundo", [1204], 3, 4, 0.6),
             _wt("?", [1205], 4, 5, 0.4)]
    _assert_words(OD.merge_punctuations(words),
                  [_wt(" ¿Que", [1201, 1202], 1, 2, 0.9), _wt(" pasa", [1203], 2, 3), _wt(" mundo?", [1204, 1205], 3, 4, 0.6)])


def test_merge_punctuations_japanese():  # UnitTests.swift:2624-2667
    from oracle import decode as OD
    words = [_wt("<|0.00|>", [50364], 0, 1), _wt("こんにちは", [38088], 1, 2), _wt("、", [1231], 2, 3), _wt("世界", [24486], 3, 4),
             _wt("！", [171, 120, 223], 4, 5), _wt("これは", [25212], 5, 6), _wt("テ", [22985], 6, 7), _wt("スト", [40498], 7, 8),
             _wt("です", [4767], 8, 9), _wt("よね", [30346], 9, 10), _wt("？", [171, 120, 253], 10, 11), _wt("<|endoftext|>", [50257], 11, 12)]
    _assert_words(OD.merge_punctuations(words),
                  [_wt("<|0.00|>", [50364], 0, 1), _wt("こんにちは、", [38088, 1231], 1, 2), _wt("世界！", [24486, 171, 120, 223], 3, 4),
                   _wt("これは", [25212], 5, 6), _wt("テ", [22985], 6, 7), _wt("スト", [40498], 7, 8), _wt("です", [4767], 8, 9),
                   _wt("よね？", [30346, 171, 120, 253], 9, 10), _wt("<|endoftext|>", [50257], 11, 12)])


# ----------------------------------------------------------------------------- word-duration KATs (UnitTests.swift:2754-2937)
def _seg(i, start, end, tokens):
    from oracle import decode as OD
    return OD.TranscriptionSegment(id=i, seek=0, start=start, end=end, text="", tokens=list(tokens), tokenLogProbs=[{0: 0.0}] * len(tokens),
                                   temperature=0.0, avgLogprob=0.0, compressionRatio=1.0, noSpeechProb=0.0)


def test_long_word_durations():  # UnitTests.swift:2754-2867
    from oracle import decode as OD
    words = [_wt(" The", [264], 0.5, 1.0), _wt(" first", [4589], 1.0, 2.0), _wt(" segment", [234], 2.0, 3.0), _wt(" with", [567], 3.0, 4.0),
             _wt(" a", [257], 4.0, 5.0), _wt(" long", [890], 5.0, 6.0), _wt(" ending", [123], 6.0, 35.0), _wt(".", [13], 35.0, 35.0)]
    segments = [_seg(0, 0.0, 6.0, [264, 4589, 234, 567, 257, 890]), _seg(1, 6.5, 30.0, [123, 13])]
    med, mx = OD.calculate_word_duration_constraints(words)
    assert med == pytest.approx(0.7, abs=1e-7) and mx == pytest.approx(1.4, abs=1e-6)
    merged = OD.merge_punctuations(OD.truncate_long_words_at_sentence_boundaries(words, mx))
    upd = OD.update_segments_with_word_timings(segments, merged, 0, 0.0, med, mx, specialTokenBegin=50257)
    allw = [w for g in upd for w in g.words]
    assert len(upd) == 2
    assert allw[-1].duration == pytest.approx(mx, abs=1e-4)
    assert upd[-1].end - upd[-1].start <= 19.5
    k = next(i for i, w in enumerate(allw) if w.word == " ending.")
    assert allw[k].duration == pytest.approx(mx, abs=1e-4)
    assert allw[k].start == pytest.approx(33.6, abs=1e-5)
    assert all(a.end <= b.start for a, b in zip(allw, allw[1:]))


def test_single_token_segment_word_duration():  # UnitTests.swift:2869-2937
    from oracle import decode as OD
    words = [_wt("<|notimestamps|>", [50363], 0, 0.5), _wt(" Hello", [314], 0.5, 20.5), _wt("<|endoftext|>", [50257], 20.5, 30)]
    segments = [_seg(0, 0.0, 30.0, [314])]
    med, mx = OD.calculate_word_duration_constraints(words)
    assert med == pytest.approx(0.7, abs=1e-7) and mx == pytest.approx(1.4, abs=1e-6)
    merged = OD.merge_punctuations(OD.truncate_long_words_at_sentence_boundaries(words, mx))
    upd = OD.update_segments_with_word_timings(segments, merged, 0, 0.0, med, mx, specialTokenBegin=50257)
    ws = upd[0].words
    hello = next(w for w in ws if w.word == " Hello")
    assert hello.duration <= mx + 1e-6
    prev_end = 0.0
    for w in ws:
        assert w.start >= prev_end and w.duration <= mx + 1e-6
        prev_end = w.end


def test_float16_timestamp_rule_emulation_differs_from_fp32_exactly_where_float16_ties():
    """Reference-numerics switch (wh_decoding_options.float16_logits, oracle DecodingOptions.float16Logits): the reference compares
    logSumExp(timestamp log-probs) > max(text log-probs) on FloatType = Float16 values (Core/Text/LogitsFilter.swift:144-242).
    Two log-probabilities 0.002 apart near -6.9 are distinct in fp32 and equal in Float16 (spacing 2^-8 there)."""
    from oracle import decode as OD
    f = OD.TimestampRulesFilter._sumOfProbabilityOverTimestampsIsAboveAnyOtherToken
    x = np.zeros(20000, dtype=np.float32)        # log-sum-exp ~ 9.9
    tb = 15000
    x[10] = 3.0                                   # best text logit
    x[tb + 5] = 3.002                             # a single dominant timestamp logit, everything else negligible
    x[tb:tb + 5] = -30.0; x[tb + 6:] = -30.0
    assert f(x, tb, False) is True                # fp32: -6.905 > -6.907
    assert f(x, tb, True) is False                # Float16: both round to the same value -> not strictly greater
    x[tb + 5] = 3.02
    assert f(x, tb, True) is True


def test_fill_indexes_with_value():
    """UnitTests.swift:1903-1920 (testFillIndexesWithValue): the two in-place writers the logits filters are built from."""
    base = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7]
    mk = lambda: np.array(base, np.float32).reshape(1, 1, -1)
    ninf = -np.inf
    np.testing.assert_array_equal(D.fill_indexes(mk(), [], ninf)[0, 0], np.array(base, np.float32))
    np.testing.assert_array_equal(D.fill_indexes(mk(), [[0, 0, 0], [0, 0, 1], [0, 0, 5]], ninf)[0, 0], np.array([ninf, ninf, 0.3, 0.4, 0.5, ninf, 0.7], np.float32))
    np.testing.assert_array_equal(D.fill_last_dimension(mk(), range(0, 1), ninf)[0, 0], np.array([ninf, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7], np.float32))
    np.testing.assert_array_equal(D.fill_last_dimension(mk(), range(2, 5), ninf)[0, 0], np.array([0.1, 0.2, ninf, ninf, ninf, 0.6, 0.7], np.float32))
    with pytest.raises(AssertionError):
        D.fill_last_dimension(np.zeros((2, 1, 7), np.float32), range(0, 1), ninf)          # the reference's precondition: [1, 1, n]


def test_batched_array():
    """UnitTests.swift:1922-1931 (testBatchedArray): Array.batched(into:), the grouping of WhisperKit.transcribeWithOptions (Core/WhisperKit.swift:739)."""
    assert D.batched([], 1) == [] and D.batched([1, 2, 3, 4], 1) == [[1], [2], [3], [4]]
    assert D.batched([], 10) == [] and D.batched([1, 2, 3, 4], 10) == [[1, 2, 3, 4]]
    assert D.batched([], 3) == [] and D.batched([1, 2, 3, 4], 3) == [[1, 2, 3], [4]]
