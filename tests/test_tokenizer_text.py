"""Tokenizer text side of the path: oracle restatement pinned to the reference's own known-answer
tests, cross-checked against the HF `tokenizers` library, and the native implementation behind the C ABI compared with the oracle.
All host logic: runs without a GPU."""
import os
import random

import numpy as np
import pytest

from oracle import tokenizer as otok
from whisperkit_amd import synth


@pytest.fixture(scope="module")
def tok_dirs(tmp_path_factory):
    out = {}
    for n in (51864, 51865, 51866):
        d = tmp_path_factory.mktemp(f"tok{n}")
        synth.write_kat_tokenizer(str(d), n)
        out[n] = str(d)
    return out


@pytest.fixture(scope="module")
def otoks(tok_dirs):
    return {n: otok.Tokenizer(os.path.join(d, "tokenizer.json")) for n, d in tok_dirs.items()}


# ---------------------------------------------------------------------------------------------- reference KATs on the oracle
def test_kat_tokenizer_output(otoks):
    """UnitTests.swift:1288-1297 testTokenizerOutput (large-v3 vocabulary)."""
    ids = [50364, 400, 370, 452, 7177, 6280, 1029, 406, 437, 428, 1941, 393, 360, 337, 291, 1029, 437, 291, 393, 360, 337, 428,
           1941, 13, 50889]
    assert otoks[51866].decode(ids) == ("<|notimestamps|> And so my fellow Americans ask not what your country can do for you "
                                        "ask what you can do for your country.<|10.48|>")


def test_kat_split_to_word_tokens_english(otoks):
    """UnitTests.swift:1326-1341 testSplitToWordTokens."""
    ids = [50364, 2425, 11, 1002, 0, 50414, 50414, 639, 307, 257, 220, 31636, 11, 1943, 380, 309, 30, 50257]
    words, wt = otoks[51865].splitToWordTokens(ids, "en")
    assert words == ["<|0.00|>", " Hello", ",", " world", "!", "<|1.00|>", "<|1.00|>", " This", " is", " a", " test", ",", " isn't",
                     " it", "?", "<|endoftext|>"]
    assert wt == [[50364], [2425], [11], [1002], [0], [50414], [50414], [639], [307], [257], [220, 31636], [11], [1943, 380], [309],
                  [30], [50257]]
    assert words != [otoks[51865].convertIdToToken(i) for i in ids]


def test_kat_split_to_word_tokens_spanish(otoks):
    """UnitTests.swift:1343-1358 testSplitToWordTokensSpanish."""
    ids = [50363, 24364, 48529, 376, 6043, 0, 20547, 785, 2002, 48241, 11, 3841, 1771, 30, 50257]
    words, wt = otoks[51865].splitToWordTokens(ids, "es")
    assert words == ["<|notimestamps|>", "¡Hola", " Mundo", "!", " Esta", " es", " una", " prueba", ",", " ¿no", "?", "<|endoftext|>"]
    assert wt == [[50363], [24364, 48529], [376, 6043], [0], [20547], [785], [2002], [48241], [11], [3841, 1771], [30], [50257]]


def test_kat_split_to_word_tokens_japanese(otoks):
    """UnitTests.swift:1360-1375 testSplitToWordTokensJapanese (unicode splitter: raw byte tokens regroup into characters)."""
    ids = [50364, 38088, 1231, 24486, 171, 120, 223, 25212, 22985, 40498, 4767, 30346, 171, 120, 253, 50257]
    words, wt = otoks[51865].splitToWordTokens(ids, "ja")
    assert words == ["<|0.00|>", "こんにちは", "、", "世界", "！", "これは", "テ", "スト", "です", "よね", "？", "<|endoftext|>"]
    assert wt == [[50364], [38088], [1231], [24486], [171, 120, 223], [25212], [22985], [40498], [4767], [30346], [171, 120, 253],
                  [50257]]


def test_special_tokens_from_vocabulary(otoks):
    """WhisperTokenizerWrapper.init (Core/Models.swift:1198-1224): ids looked up by token text equal the ids the decoder oracle
    derives from the vocabulary size; multilingual ids equal the reference defaults (:1309-1322)."""
    from oracle import decode as od
    for n, t in otoks.items():
        st, langs = od.special_tokens_for_vocab(n)
        assert t.specialTokens() == st
        assert t.allLanguageTokens(synth.LANGUAGE_CODES) == set(langs)
    assert otoks[51865].specialTokens() == od.SpecialTokens()
    assert otok.trimming_special_token_characters(otoks[51865].decode([50357])) == "su"


def test_oracle_decode_matches_hf_tokenizers(tok_dirs, otoks):
    """The HF `tokenizers` library (what swift-transformers mirrors) decodes the same ids to the same text; it has no cleanUp
    step (that lives in transformers / Tokenizer.swift:433-449), so compare against the oracle with cleanUp disabled, and it
    treats a run across an added token as one byte string - equal unless an added token splits a broken UTF-8 sequence."""
    hf = pytest.importorskip("tokenizers")
    for n in (51865, 51864):
        t = hf.Tokenizer.from_file(os.path.join(tok_dirs[n], "tokenizer.json"))
        o = otoks[n]
        rng = random.Random(n)
        keep = o.clean_up
        o.clean_up = False
        try:
            for trial in range(300):
                ids = [rng.randrange(0, n) if rng.random() < 0.9 else rng.randrange(0, 256) for _ in range(rng.randrange(1, 40))]
                want = t.decode(ids, skip_special_tokens=False)
                got = o.decode(ids)
                if got != want:      # only allowed around an added token that cuts an incomplete multi-byte sequence
                    assert any(i in o.special_ids for i in ids) and otok.REPLACEMENT in got, (ids, got, want)
                assert o.decode(ids, skipSpecialTokens=True) == \
                    (lambda s: s)(o.decode([i for i in ids if i not in o.special_ids]))
        finally:
            o.clean_up = keep


def test_clean_up_tokenization_spaces(otoks):
    """Tokenizer.swift:433-449: the ten literal replacements, applied in order, only when the tokenizer config asks for it."""
    o = otoks[51865]
    assert o.cleanUp("we 're here , ok ? yes ! it 's . i 'm , they 've , is n't") == "we're here, ok? yes! it's. i'm, they've, isn't"
    assert o.cleanUp("a ' b") == "a'b"
    ids = [2425, 220, 11, 1002, 220, 0]                       # " Hello" " " "," " world" " " "!"
    assert o.decode(ids) == " Hello, world!"
    o.clean_up = False
    try:
        assert o.decode(ids) == " Hello , world !"
    finally:
        o.clean_up = True


# ---------------------------------------------------------------------------------------------- native (C ABI) vs oracle
from whisperkit_amd import api   # noqa: E402  (host-only entry points: no GPU needed)
from oracle import decode as od  # noqa: E402


@pytest.fixture(scope="module")
def ntoks(tok_dirs):
    return {n: api.Tokenizer(os.path.join(d, "tokenizer.json")) for n, d in tok_dirs.items()}


def _random_ids(rng, n_vocab, length):
    out = []
    for _ in range(length):
        r = rng.random()
        if r < 0.15:
            out.append(rng.randrange(0, 256))                 # raw byte tokens: broken UTF-8 runs
        elif r < 0.25:
            out.append(rng.randrange(n_vocab - 1600, n_vocab))  # specials / timestamps
        elif r < 0.3:
            out.append(rng.choice([11, 13, 0, 30, 220, 1231]))  # punctuation, space
        else:
            out.append(rng.randrange(256, n_vocab - 1600))
    return out


def test_native_reference_kats(ntoks):
    """The same four reference KATs through the C ABI."""
    t = ntoks[51866]
    assert t.decode([50364, 400, 370, 452, 7177, 6280, 1029, 406, 437, 428, 1941, 393, 360, 337, 291, 1029, 437, 291, 393, 360, 337, 428,
                     1941, 13, 50889]) == ("<|notimestamps|> And so my fellow Americans ask not what your country can do for you "
                                           "ask what you can do for your country.<|10.48|>")
    t = ntoks[51865]
    w, wt = t.splitToWordTokens([50364, 2425, 11, 1002, 0, 50414, 50414, 639, 307, 257, 220, 31636, 11, 1943, 380, 309, 30, 50257], "en")
    assert w == ["<|0.00|>", " Hello", ",", " world", "!", "<|1.00|>", "<|1.00|>", " This", " is", " a", " test", ",", " isn't", " it", "?",
                 "<|endoftext|>"]
    assert wt[10] == [220, 31636] and wt[12] == [1943, 380]
    w, wt = t.splitToWordTokens([50363, 24364, 48529, 376, 6043, 0, 20547, 785, 2002, 48241, 11, 3841, 1771, 30, 50257], "es")
    assert w == ["<|notimestamps|>", "¡Hola", " Mundo", "!", " Esta", " es", " una", " prueba", ",", " ¿no", "?", "<|endoftext|>"]
    w, wt = t.splitToWordTokens([50364, 38088, 1231, 24486, 171, 120, 223, 25212, 22985, 40498, 4767, 30346, 171, 120, 253, 50257], "ja")
    assert w == ["<|0.00|>", "こんにちは", "、", "世界", "！", "これは", "テ", "スト", "です", "よね", "？", "<|endoftext|>"]
    assert wt[4] == [171, 120, 223]


@pytest.mark.parametrize("n_vocab", [51864, 51865, 51866])
def test_native_decode_and_special_tokens_equal_oracle(ntoks, otoks, n_vocab):
    n, o = ntoks[n_vocab], otoks[n_vocab]
    st = o.specialTokens()
    c = n.specialTokens
    assert (c.end_token, c.english_token, c.no_speech_token, c.no_timestamps_token, c.special_token_begin, c.start_of_previous_token,
            c.start_of_transcript_token, c.time_token_begin, c.transcribe_token, c.translate_token, c.whitespace_token) == \
        (st.endToken, st.englishToken, st.noSpeechToken, st.noTimestampsToken, st.specialTokenBegin, st.startOfPreviousToken,
         st.startOfTranscriptToken, st.timeTokenBegin, st.transcribeToken, st.translateToken, st.whitespaceToken)
    langs = sorted(o.allLanguageTokens(synth.LANGUAGE_CODES))
    assert (c.language_token_begin, c.n_language_tokens) == (langs[0], len(langs))
    assert n.vocabSize == n_vocab
    assert n.convertTokenToId("<|startoftranscript|>") == st.startOfTranscriptToken and n.convertTokenToId("nope-not-a-token") is None
    assert n.convertIdToToken(st.endToken) == "<|endoftext|>" and n.convertIdToToken(n_vocab + 5) is None
    assert n.languageToken("en") == st.englishToken and n.languageToken("xx") is None
    rng = random.Random(7 + n_vocab)
    for trial in range(400):
        ids = _random_ids(rng, n_vocab, rng.randrange(0, 48)) + ([n_vocab + 3] if trial % 50 == 0 else [])   # unknown id is dropped
        assert n.decode(ids) == o.decode(ids), ids
        assert n.decode(ids, skipSpecialTokens=True) == o.decode(ids, skipSpecialTokens=True), ids


@pytest.mark.parametrize("language", ["en", "ja", "zh", "de"])
def test_native_split_to_word_tokens_equals_oracle(ntoks, otoks, language):
    n, o = ntoks[51865], otoks[51865]
    rng = random.Random(sum(map(ord, language)))
    for trial in range(150):
        ids = _random_ids(rng, 51865, rng.randrange(0, 60))
        assert n.splitToWordTokens(ids, language) == tuple(o.splitToWordTokens(ids, language)), ids


def _window(rng, st, n_text=40, n_segments=3):
    """A decoded window the way findSeekPointAndSegments slices it: [<|t0|> text.. <|t1|>][<|t1|> text.. <|t2|>]..."""
    tb = st.timeTokenBegin
    times = sorted(rng.sample(range(0, 1400, 2), n_segments + 1))
    segs, tokens, lps = [], [], []
    for s in range(n_segments):
        toks = [tb + times[s]]
        for _ in range(rng.randrange(1, n_text // n_segments + 2)):
            r = rng.random()
            toks.append(rng.choice([11, 13, 0, 30, 220, 1231, 6, 7, 1]) if r < 0.25 else rng.randrange(256, 50000))
        toks.append(tb + times[s + 1])
        lp = [-rng.random() * 2 for _ in toks]
        segs.append(od.TranscriptionSegment(id=s, seek=0, start=times[s] * 0.02, end=times[s + 1] * 0.02, text="", tokens=toks,
                                            tokenLogProbs=[{t: l} for t, l in zip(toks, lp)], temperature=0.0, avgLogprob=-0.3,
                                            compressionRatio=1.2, noSpeechProb=0.0))
        tokens += toks
        lps += lp
    return segs, tokens, lps


def _alignment(rng_np, n_rows):
    """Cross-attention-like matrix: noise plus a monotone ridge with random speed (so words get uneven durations)."""
    a = rng_np.random((n_rows, 1500)).astype(np.float32) * 0.2
    pos = np.sort(rng_np.integers(0, 1500, n_rows))
    for r in range(n_rows):
        lo, hi = max(0, pos[r] - 3), min(1500, pos[r] + 4)
        a[r, lo:hi] += 1.0
    return a


@pytest.mark.parametrize("language", ["en", "ja"])
def test_native_add_word_timestamps_equals_oracle(ntoks, otoks, language):
    """SegmentSeeker.addWordTimestamps end to end (DTW, word grouping, duration constraints, punctuation merge,
    updateSegmentsWithWordTimings) on random windows: native C++ against the oracle, Float for Float."""
    n, o = ntoks[51865], otoks[51865]
    st = o.specialTokens()
    rng = random.Random(11)
    rng_np = np.random.default_rng(5)
    compared = 0
    for trial in range(60):
        osegs, tokens, lps = _window(rng, st, n_text=rng.randrange(6, 60), n_segments=rng.randrange(1, 5))
        seek = rng.choice([0, 160000, 480000 * 3 + 3200])
        last = float(np.float32(seek) / np.float32(16000))
        align = _alignment(rng_np, len(tokens))
        want = od.add_word_timestamps(osegs, align, lambda ids: o.splitToWordTokens(ids, language), o.decode, seek, last, st.specialTokenBegin)
        asegs = [api.TranscriptionSegment(g.id, g.seek, g.start, g.end, g.tokens, [lp[t] for lp, t in zip(g.tokenLogProbs, g.tokens)],
                                          g.temperature, g.avgLogprob, g.compressionRatio, g.noSpeechProb, []) for g in osegs]
        got = api.addWordTimestamps(asegs, np.vstack([align, np.zeros((224 - len(align), 1500), np.float32)]) if len(align) < 224 else align,
                                    n, seek, last, language)
        assert len(got.segments) == len(want)
        for g, w in zip(got.segments, want):
            assert np.float32(g.start) == np.float32(w.start) and np.float32(g.end) == np.float32(w.end), (trial, g, w)
            assert [x.word for x in g.words] == [x.word for x in w.words], trial
            assert [x.tokens for x in g.words] == [x.tokens for x in w.words], trial
            for x, y in zip(g.words, w.words):
                assert np.float32(x.start) == np.float32(y.start) and np.float32(x.end) == np.float32(y.end), (trial, x, y)
                assert abs(x.probability - y.probability) <= 0.0101, (trial, x, y)      # expf vs np.exp at a .5 rounding edge
                compared += 1
            assert g.text == o.decode(w.tokens)
    assert compared > 250


def _decoded_window(rng, st, kind):
    """Result tokens SOT..EOT of one window in the shapes findSeekPointAndSegments distinguishes."""
    tb = st.timeTokenBegin
    prompt = [st.startOfTranscriptToken, st.englishToken, st.transcribeToken]
    text = lambda k: [rng.choice([11, 13, 0, 30, 220, 6, 1]) if rng.random() < 0.25 else rng.randrange(256, 50000) for _ in range(k)]
    t = sorted(rng.sample(range(2, 1400, 2), 4))
    if kind == "pairs":          # two complete segments, consecutive timestamp pairs
        body = [tb] + text(rng.randrange(2, 9)) + [tb + t[0], tb + t[0]] + text(rng.randrange(2, 9)) + [tb + t[1], tb + t[1]] + text(3) + [tb + t[2]]
    elif kind == "single":       # single timestamp ending
        body = [tb] + text(rng.randrange(2, 9)) + [tb + t[0], tb + t[0]] + text(rng.randrange(2, 6)) + [tb + t[1]]
    elif kind == "none":         # no timestamp at the end
        body = [tb] + text(rng.randrange(2, 9)) + [tb + t[0], tb + t[0]] + text(rng.randrange(2, 6))
    else:                        # no consecutive timestamps at all
        body = [tb] + text(rng.randrange(3, 12)) + ([tb + t[3]] if rng.random() < 0.5 else [])
    toks = prompt + body + [st.endToken]
    return toks, [-rng.random() * 1.5 for _ in toks]


@pytest.mark.parametrize("word_timestamps,skip_special", [(False, False), (True, False), (True, True)])
def test_native_window_assembly_equals_oracle(ntoks, otoks, word_timestamps, skip_special):
    """TranscribeTask.run windowing over a sequence of decoded windows (Core/TranscribeTask.swift:175-312): segments, texts, seek
    chain (never backward, refined by word timings, maxWindowSeek), zero-length filter, result text and language."""
    n, o = ntoks[51865], otoks[51865]
    st = o.specialTokens()
    rng = random.Random(31 + word_timestamps + 2 * skip_special)
    rng_np = np.random.default_rng(8)
    for trial in range(12):
        mws = rng.choice([None, None, 200000])
        oopt = od.DecodingOptions(wordTimestamps=word_timestamps, skipSpecialTokens=skip_special, maxWindowSeek=mws,
                                  noSpeechThreshold=rng.choice([None, 0.6]))
        aopt = api.DecodingOptions(wordTimestamps=word_timestamps, skipSpecialTokens=skip_special, maxWindowSeek=mws,
                                   noSpeechThreshold=oopt.noSpeechThreshold)
        asm = api.WindowAssembler(aopt, n)
        seek_o = seek_n = 0
        all_segments, all_tokens = [], []
        for w in range(rng.randrange(1, 5)):
            toks, lps = _decoded_window(rng, st, rng.choice(["pairs", "single", "none", "lump"]))
            nsp = rng.choice([0.0, 0.0, 0.9])
            avg = -0.4 if nsp == 0.0 else rng.choice([-0.4, -2.0])       # a silent window is skipped unless avgLogProb is high
            align = _alignment(rng_np, len(toks)) if word_timestamps else None
            ores = od.DecodingResult(language="en", tokens=toks, tokenLogProbs=[{t: l} for t, l in zip(toks, lps)], avgLogProb=avg, noSpeechProb=nsp,
                                     temperature=0.0, compressionRatio=1.3, fallback=None, alignment=align)
            ares = api.DecodingResult(toks, lps, avg, nsp, 0.0, 1.3, st.englishToken, None, False, False, len(toks))
            seg_size = rng.choice([480000, 300000])
            seek_o, cur = od.windowing(ores, oopt, len(all_segments), seek_o, seg_size, st, o, "en")
            seek_n = asm.addWindow(ares, seek_n, seg_size, align)
            assert seek_n == seek_o, (trial, w)
            if cur is not None:
                all_segments += cur
                for g in cur:
                    all_tokens += g.tokens
        got = asm.result()
        assert [g.tokens for g in got.segments] == [g.tokens for g in all_segments]
        assert [g.text for g in got.segments] == [g.text for g in all_segments]
        assert [g.id for g in got.segments] == [g.id for g in all_segments]
        assert [g.seek for g in got.segments] == [g.seek for g in all_segments]
        for g, w in zip(got.segments, all_segments):
            assert np.float32(g.start) == np.float32(w.start) and np.float32(g.end) == np.float32(w.end)
            ww = w.words or []
            assert [x.word for x in g.words] == [x.word for x in ww] and [x.tokens for x in g.words] == [x.tokens for x in ww]
            assert all(np.float32(x.start) == np.float32(y.start) and np.float32(x.end) == np.float32(y.end) for x, y in zip(g.words, ww))
        assert got.tokens == all_tokens
        assert got.text == od.trim_whitespaces(o.decode([t for t in all_tokens if t < st.specialTokenBegin]))
        assert got.language == "en"


# ---------------------------------------------------------------------------------------------- reference word KATs on the native code
def _nw(word, tokens, start, end, prob=1.0):
    return api.WordTiming(list(tokens), float(start), float(end), float(prob), word)


def _same_words(got, expected):
    assert [(g.word, g.tokens, np.float32(g.start), np.float32(g.end), np.float32(g.probability)) for g in got] == \
        [(e.word, e.tokens, np.float32(e.start), np.float32(e.end), np.float32(e.probability)) for e in expected]


def test_native_kat_merge_punctuations():
    """UnitTests.swift:2484-2667 testMergePunctuations{,Spanish,SpanishStartWithPrepend,Japanese} through wh_merge_punctuations."""
    assert api.mergePunctuations([]) == []
    words = [_nw("<|0.00|>", [50364], 0, 1), _nw(" Hello", [2425], 1, 2), _nw(",", [11], 2, 3), _nw(" world", [1002], 3, 4),
             _nw("!", [0], 4, 5), _nw("<|1.00|>", [50414], 5, 6), _nw("<|1.00|>", [50414], 6, 7), _nw(" This", [639], 7, 8),
             _nw(" is", [307], 8, 9), _nw(" a", [257], 9, 10), _nw(" test", [220, 31636], 10, 11), _nw(",", [11], 11, 12),
             _nw(" isn't", [1943, 380], 12, 13), _nw(" it", [309], 13, 14), _nw("?", [30], 14, 15), _nw("<|endoftext|>", [50257], 15, 16)]
    _same_words(api.mergePunctuations(words, prepended="\"'“¿([{-", appended="\"'.。,，!！?？:：”)]}、"),
                [_nw("<|0.00|>", [50364], 0, 1), _nw(" Hello,", [2425, 11], 1, 2), _nw(" world!", [1002, 0], 3, 4),
                 _nw("<|1.00|>", [50414], 5, 6), _nw("<|1.00|>", [50414], 6, 7), _nw(" This", [639], 7, 8), _nw(" is", [307], 8, 9),
                 _nw(" a", [257], 9, 10), _nw(" test,", [220, 31636, 11], 10, 11), _nw(" isn't", [1943, 380], 12, 13),
                 _nw(" it?", [309, 30], 13, 14), _nw("<|endoftext|>", [50257], 15, 16)])
    words = [_nw("<|notimestamps|>", [50363], 0, 1), _nw(" ¡", [24364], 0, 1), _nw("Hola", [48529], 1, 2), _nw(" Mundo", [376, 6043], 2, 3),
             _nw("!", [0], 3, 4), _nw(" Esta", [20547], 4, 5), _nw(" es", [785], 5, 6), _nw(" una", [2002], 6, 7),
             _nw(" prueba", [48241], 7, 8), _nw(",", [11], 8, 9), _nw(" ¿", [3841], 9, 10), _nw("no", [1771], 10, 11), _nw("?", [30], 11, 12),
             _nw("<|endoftext|>", [50257], 12, 13)]
    _same_words(api.mergePunctuations(words),
                [_nw("<|notimestamps|>", [50363], 0, 1), _nw(" ¡Hola", [24364, 48529], 1, 2), _nw(" Mundo!", [376, 6043, 0], 2, 3),
                 _nw(" Esta", [20547], 4, 5), _nw(" es", [785], 5, 6), _nw(" una", [2002], 6, 7), _nw(" prueba,", [48241, 11], 7, 8),
                 _nw(" ¿no?", [3841, 1771, 30], 10, 11), _nw("<|endoftext|>", [50257], 12, 13)])
    words = [_nw(" ¿", [1201], 0, 1), _nw("Que", [1202], 1, 2, 0.9), _nw(" pasa", [1203], 2, 3), _nw(" mundo", [1204], 3, 4, 0.6),
             _nw("?", [1205], 4, 5, 0.4)]
    _same_words(api.mergePunctuations(words), [_nw(" ¿Que", [1201, 1202], 1, 2, 0.9), _nw(" pasa", [1203], 2, 3), _nw(" mundo?", [1204, 1205], 3, 4, 0.6)])
    words = [_nw("<|0.00|>", [50364], 0, 1), _nw("こんにちは", [38088], 1, 2), _nw("、", [1231], 2, 3), _nw("世界", [24486], 3, 4),
             _nw("！", [171, 120, 223], 4, 5), _nw("これは", [25212], 5, 6), _nw("テ", [22985], 6, 7), _nw("スト", [40498], 7, 8),
             _nw("です", [4767], 8, 9), _nw("よね", [30346], 9, 10), _nw("？", [171, 120, 253], 10, 11), _nw("<|endoftext|>", [50257], 11, 12)]
    _same_words(api.mergePunctuations(words),
                [_nw("<|0.00|>", [50364], 0, 1), _nw("こんにちは、", [38088, 1231], 1, 2), _nw("世界！", [24486, 171, 120, 223], 3, 4),
                 _nw("これは", [25212], 5, 6), _nw("テ", [22985], 6, 7), _nw("スト", [40498], 7, 8), _nw("です", [4767], 8, 9),
                 _nw("よね？", [30346, 171, 120, 253], 9, 10), _nw("<|endoftext|>", [50257], 11, 12)])


def _nseg(i, start, end, tokens):
    return api.TranscriptionSegment(i, 0, start, end, list(tokens), [0.0] * len(tokens), 0.0, 0.0, 1.0, 0.0, [])


def test_native_kat_long_word_durations():
    """UnitTests.swift:2754-2867 testLongWordDurations through wh_update_segments_with_word_timings, and equal to the oracle."""
    words = [_nw(" The", [264], 0.5, 1.0), _nw(" first", [4589], 1.0, 2.0), _nw(" segment", [234], 2.0, 3.0), _nw(" with", [567], 3.0, 4.0),
             _nw(" a", [257], 4.0, 5.0), _nw(" long", [890], 5.0, 6.0), _nw(" ending", [123], 6.0, 35.0), _nw(".", [13], 35.0, 35.0)]
    upd, med, mx = api.updateSegmentsWithWordTimings([_nseg(0, 0.0, 6.0, [264, 4589, 234, 567, 257, 890]), _nseg(1, 6.5, 30.0, [123, 13])],
                                                     words, 0, 0.0, 50257)
    assert med == pytest.approx(0.7, abs=1e-7) and mx == pytest.approx(1.4, abs=1e-6)
    allw = [w for g in upd for w in g.words]
    assert len(upd) == 2
    assert allw[-1].end - allw[-1].start == pytest.approx(mx, abs=1e-4)
    assert upd[-1].end - upd[-1].start <= 19.5
    k = next(i for i, w in enumerate(allw) if w.word == " ending.")
    assert allw[k].end - allw[k].start == pytest.approx(mx, abs=1e-4) and allw[k].start == pytest.approx(33.6, abs=1e-5)
    assert all(a.end <= b.start for a, b in zip(allw, allw[1:]))
    ow = [od.WordTiming(w.word, w.tokens, w.start, w.end, w.probability) for w in words]
    osegs = [od.TranscriptionSegment(0, 0, 0.0, 6.0, "", [264, 4589, 234, 567, 257, 890], [], 0, 0, 1, 0),
             od.TranscriptionSegment(1, 0, 6.5, 30.0, "", [123, 13], [], 0, 0, 1, 0)]
    omed, omx = od.calculate_word_duration_constraints(ow)
    want = od.update_segments_with_word_timings(osegs, od.merge_punctuations(od.truncate_long_words_at_sentence_boundaries(ow, omx)), 0, 0.0,
                                                omed, omx, 50257)
    for g, w in zip(upd, want):
        assert np.float32(g.start) == np.float32(w.start) and np.float32(g.end) == np.float32(w.end)
        _same_words(g.words, [_nw(x.word, x.tokens, x.start, x.end, x.probability) for x in w.words])


def test_native_kat_single_token_segment_word_duration(ntoks):
    """UnitTests.swift:2869-2937 testSingleTokenSegmentWordDuration."""
    words = [_nw("<|notimestamps|>", [50363], 0, 0.5), _nw(" Hello", [314], 0.5, 20.5), _nw("<|endoftext|>", [50257], 20.5, 30)]
    upd, med, mx = api.updateSegmentsWithWordTimings([_nseg(0, 0.0, 30.0, [314])], words, 0, 0.0, 50257, ntoks[51865])
    assert med == pytest.approx(0.7, abs=1e-7) and mx == pytest.approx(1.4, abs=1e-6)
    ws = upd[0].words
    hello = next(w for w in ws if w.word == " Hello")
    assert hello.end - hello.start <= mx + 1e-6
    prev_end = 0.0
    for w in ws:
        assert w.start >= prev_end and w.end - w.start <= mx + 1e-6
        prev_end = w.end


def test_native_text_entry_points_reject_bad_arguments(tmp_path, ntoks):
    """Error behaviour of the host-only entry points: status codes mirror WhisperError, nothing aborts."""
    with pytest.raises(api.WhisperError) as e:
        api.Tokenizer(str(tmp_path / "missing.json"))
    assert e.value.code == 1                                             # tokenizerUnavailable
    bad = tmp_path / "tokenizer.json"
    bad.write_text('{"model": {"type": "WordPiece", "vocab": {}}, "decoder": {"type": "WordPiece"}}')
    with pytest.raises(api.WhisperError) as e:
        api.Tokenizer(str(bad))
    assert e.value.code == 1
    bad.write_text('{"model": ')
    with pytest.raises(api.WhisperError):
        api.Tokenizer(str(bad))
    seg = _nseg(0, 0.0, 1.0, [50363, 314])
    with pytest.raises(api.WhisperError) as e:                           # mixed special + text word needs a tokenizer to re-decode
        api.updateSegmentsWithWordTimings([seg], [_nw("<|notimestamps|> Hello", [50363, 314], 0, 1), _nw(" x", [400], 1, 2)], 0, 0.0, 50257)
    assert e.value.code == 1
    plain = api.makeTranscriptionResult([seg], None, specialTokens=ntoks[51865].specialTokens)
    assert plain.text is None
    with pytest.raises(api.WhisperError):
        plain.writeSRT(str(tmp_path / "x.srt"))                          # no text without a tokenizer
