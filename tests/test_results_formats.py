"""Result assembly, on-disk formats and audio ingest: native host code behind the C ABI against the
oracle restatement and the reference's own pins.  Host logic only - runs without a GPU."""
import json
import math
import os
import random
import wave

import numpy as np
import pytest

from oracle import decode as od
from oracle import tokenizer as otok
from whisperkit_amd import _lib as L
from whisperkit_amd import api, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def toks(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("tok"))
    p = synth.write_kat_tokenizer(d, 51865)
    return api.Tokenizer(p), otok.Tokenizer(p)


def _segment(i, seek, start, end, tokens, rng):
    lps = [-rng.random() for _ in tokens]
    return api.TranscriptionSegment(i, seek, start, end, list(tokens), lps, 0.0, -0.25, 1.5, 0.0, [])


JFK = [400, 370, 452, 7177, 6280, 1029, 406, 437, 428, 1941, 393, 360, 337, 291, 13]


# ---------------------------------------------------------------------------------------------- formatTime / SRT / VTT / JSON
def test_format_time_known_values_and_oracle():
    """ResultWriter.swift:14-26."""
    assert api.formatTime(0.0, True, ",") == "00:00:00,000"
    assert api.formatTime(3725.5, True, ",") == "01:02:05,500"
    assert api.formatTime(65.25, False, ".") == "01:05.250"
    assert api.formatTime(3600.0, False, ".") == "01:00:00.000"          # hours appear once they are non-zero
    rng = random.Random(3)
    for _ in range(2000):
        s = float(np.float32(rng.random() * rng.choice([1, 60, 4000, 40000])))
        for hours, marker in ((True, ","), (False, ".")):
            assert api.formatTime(s, hours, marker) == od.format_time(s, hours, marker), s


def _result_with_words(toks, rng):
    n, o = toks
    tb = o.specialTokens().timeTokenBegin
    t1 = [tb] + JFK[:7] + [tb + 120]
    t2 = [tb + 120] + JFK[7:] + [tb + 260]
    segs = [_segment(0, 0, 0.0, 2.4, t1, rng), _segment(1, 0, 2.4, 5.2, t2, rng)]
    align = np.random.default_rng(1).random((224, 1500)).astype(np.float32) * 0.1
    for r in range(len(t1) + len(t2)):
        align[r, 10 + 12 * r:18 + 12 * r] += 1.0
    return api.addWordTimestamps(segs, align, n, 0, 0.0, "en"), segs


def test_srt_vtt_writers_equal_oracle(toks, tmp_path):
    """WriteSRT / WriteVTT (ResultWriter.swift:70-134): one cue per word when a segment has words, else per segment."""
    n, o = toks
    rng = random.Random(5)
    res, _ = _result_with_words(toks, rng)
    assert sum(len(g.words) for g in res.segments) >= 10
    osegs = [od.TranscriptionSegment(g.id, g.seek, g.start, g.end, g.text, g.tokens, [], 0, 0, 0, 0,
                                     [od.WordTiming(w.word, w.tokens, w.start, w.end, w.probability) for w in g.words]) for g in res.segments]
    res.writeSRT(str(tmp_path / "a.srt"))
    res.writeVTT(str(tmp_path / "a.vtt"))
    assert (tmp_path / "a.srt").read_text(encoding="utf-8") == od.srt_text(osegs)
    assert (tmp_path / "a.vtt").read_text(encoding="utf-8") == od.vtt_text(osegs)
    srt = (tmp_path / "a.srt").read_text(encoding="utf-8")
    assert srt.startswith("1\n00:00:00,") and " --> " in srt and " And\n\n" in srt
    # without word timings the cue is the segment
    plain = api.makeTranscriptionResult([_segment(0, 0, 1.0, 3.5, [50364] + JFK + [50364 + 175], rng)], n, languageToken=50259)
    plain.writeVTT(str(tmp_path / "b.vtt"))
    assert (tmp_path / "b.vtt").read_text(encoding="utf-8") == \
        "WEBVTT\n\n00:01.000 --> 00:03.500\n<|0.00|> And so my fellow Americans ask not what your country can do for you.<|3.50|>\n\n"
    assert plain.text == "And so my fellow Americans ask not what your country can do for you." and plain.language == "en"


def test_json_writer_carries_the_codable_fields(toks, tmp_path):
    """WriteJSON = JSONEncoder on TranscriptionResult (Core/Models.swift:447-466, 573-641, 730-763): same keys, same nesting;
    tokenLogProbs is [[tokenId: logprob]]; words only where present; seekTime null when nil."""
    rng = random.Random(9)
    res, _ = _result_with_words(toks, rng)
    res.writeJSON(str(tmp_path / "r.json"))
    j = json.loads((tmp_path / "r.json").read_text(encoding="utf-8"))
    assert set(j) == {"text", "segments", "language", "timings", "seekTime"} and j["seekTime"] is None
    assert j["text"] == res.text and j["language"] == res.language
    seg_keys = {"id", "seek", "start", "end", "text", "tokens", "tokenLogProbs", "temperature", "avgLogprob", "compressionRatio", "noSpeechProb", "words"}
    for g, s in zip(j["segments"], res.segments):
        assert set(g) == seg_keys
        assert g["tokens"] == s.tokens and g["text"] == s.text
        assert [list(d.keys()) for d in g["tokenLogProbs"]] == [[str(t)] for t in s.tokens]
        assert np.allclose([list(d.values())[0] for d in g["tokenLogProbs"]], s.tokenLogProbs, atol=1e-7)
        assert np.float32(g["start"]) == np.float32(s.start) and np.float32(g["end"]) == np.float32(s.end)
        assert [w["word"] for w in g["words"]] == [w.word for w in s.words]
        assert all(set(w) == {"word", "tokens", "start", "end", "probability"} for w in g["words"])
    assert set(j["timings"]) == {
        "pipelineStart", "firstTokenTime", "inputAudioSeconds", "modelLoading", "prewarmLoadTime", "encoderLoadTime", "decoderLoadTime",
        "encoderSpecializationTime", "decoderSpecializationTime", "tokenizerLoadTime", "audioLoading", "audioProcessing", "logmels",
        "encoding", "decodingInit", "decodingLoop", "decodingPredictions", "decodingFiltering", "decodingSampling", "decodingFallback",
        "decodingWindowing", "decodingKvCaching", "decodingWordTimestamps", "decodingNonPrediction", "totalAudioProcessingRuns",
        "totalLogmelRuns", "totalEncodingRuns", "totalDecodingLoops", "totalKVUpdateRuns", "totalTimestampAlignmentRuns",
        "totalDecodingFallbacks", "totalDecodingWindows", "fullPipeline"}


# ---------------------------------------------------------------------------------------------- mergeTranscriptionResults
def test_merge_transcription_results_equals_oracle(toks):
    """Utilities/TranscriptionUtilities.swift:76-157: text joined by " ", nil results skipped for segments but kept in the text join,
    segment ids = resultIndex + segmentIndex, load times max, stage times summed, fullPipeline = min(wall span, sum)."""
    n, o = toks
    rng = random.Random(2)
    parts, oparts = [], []
    for r in range(3):
        segs = [_segment(s, 1000 * r, 1.0 * s, 1.0 * s + 0.8, [50364 + 50 * s] + JFK[3 * s:3 * s + 4] + [50364 + 50 * s + 40], rng) for s in range(r + 1)]
        tm = {"pipeline_start": 100.0 + 0.5 * r, "first_token_time": 100.2 + r, "full_pipeline": 2.0 + r, "model_loading": 0.1 * (3 - r),
              "encoding": 0.3 + r, "logmels": 0.01, "decoding_loop": 1.0, "total_decoding_loops": 20 + r, "total_decoding_windows": 1,
              "input_audio_seconds": 30.0, "audio_processing": 0.001, "total_encoding_runs": 1, "total_logmel_runs": 1}
        parts.append(api.makeTranscriptionResult(segs, n, languageToken=50259 + (3 if r == 0 else 0), seekTime=30.0 * r, timings=tm))
        oparts.append({"text": parts[-1].text, "language": parts[-1].language, "timings": tm,
                       "segments": [od.TranscriptionSegment(g.id, g.seek, g.start, g.end, g.text, g.tokens, [], 0, 0, 0, 0) for g in parts[-1].segments]})
    got = api.mergeTranscriptionResults([parts[0], None, parts[1], parts[2]])
    want = od.merge_transcription_results([oparts[0], None, oparts[1], oparts[2]])
    assert got.text == want["text"] and "  " in got.text                 # the nil result leaves an empty slot in the join
    assert got.language == want["language"] == "es"
    assert [g.id for g in got.segments] == [g.id for g in want["segments"]] == [0, 1, 2, 2, 3, 4]
    assert [g.tokens for g in got.segments] == [g.tokens for g in want["segments"]]
    assert [g.text for g in got.segments] == [g.text for g in want["segments"]]
    for k, v in want["timings"].items():
        assert math.isclose(got.timings[k], v, rel_tol=1e-12, abs_tol=1e-12), k
    assert got.timings["full_pipeline"] == pytest.approx(min((101.0 + 4.0) - 100.0, 2.0 + 3.0 + 4.0))
    assert api.mergeTranscriptionResults(parts, confirmedWords=[" ask", " not"]).text == " ask not"
    empty = api.mergeTranscriptionResults([])
    assert empty.text == "" and empty.language == "en" and empty.segments == []


# ---------------------------------------------------------------------------------------------- audio ingest
def _write_wav(path, pcm16, rate=16000, channels=1):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(channels)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(np.ascontiguousarray(pcm16, dtype="<i2").tobytes())


def test_kat_audio_file_loading(tmp_path):
    """UnitTests.swift:297-316 testAudioFileLoading on jfk.wav (fixture: the file's PCM16 samples): 176 000 frames, start 1.2 s ->
    156 800, 1.2 s..3.4 s -> 35 200; samples are the int16 values / 32768."""
    pcm = np.load(os.path.join(GOLDEN, "jfk_pcm16.npz"))["pcm16"]
    p = tmp_path / "jfk.wav"
    _write_wav(p, pcm)
    a = api.loadAudio(str(p))
    assert len(a) == 176000 == 11 * 16000
    assert np.array_equal(a, pcm.astype(np.float32) / np.float32(32768.0))
    assert np.array_equal(a, od.load_wav_16k_mono(str(p)))
    b = api.loadAudio(str(p), startTime=1.2)
    assert len(b) == 156800 and np.array_equal(b, od.load_wav_16k_mono(str(p), 1.2))
    c = api.loadAudio(str(p), startTime=1.2, endTime=3.4)
    assert len(c) == 35200 and np.array_equal(c, od.load_wav_16k_mono(str(p), 1.2, 3.4))
    with pytest.raises(api.WhisperError) as e:
        api.loadAudio(str(tmp_path / "missing.wav"))
    assert e.value.code == 7                                            # WhisperError.loadAudioFailed


def test_convert_to_mono_equals_oracle():
    """AudioProcessor.swift:525-625: channel pick, sum of all / selected channels with peak renormalisation, invalid indices."""
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((6, 4000)) * 0.2).astype(np.float32)
    for mode, idx in (("sumChannels", None), ("sumChannels", [1, 3, 5]), ("sumChannels", [9, -1]), ("sumChannels", [2, 17]),
                      ("specificChannel", [0]), ("specificChannel", [4]), ("specificChannel", [11])):
        assert np.array_equal(api.convertToMono(x, mode, idx), od.convert_to_mono(x, mode, idx)), (mode, idx)
    mono = api.convertToMono(x, "sumChannels", None)
    assert abs(np.abs(mono).max() - np.abs(x).max()) < 1e-6             # keeps the loudest channel's peak
    assert np.array_equal(api.convertToMono(x[:1]), x[0])
    z = np.zeros((2, 100), np.float32)
    assert np.array_equal(api.convertToMono(z), z[0])                   # silence: scale = 0 / 1e-4


def test_resample_identity_and_band_limited_quality(tmp_path):
    """UnitTests.swift:409-461 testAudioResampleFromFile: 16 kHz -> 16 kHz returns the samples (max diff < 1e-6, here exact), also
    through the chunked file path; other rates go through our own windowed-sinc filter (AVAudioConverter is not reproducible):
    length rule and signal quality only."""
    pcm = np.load(os.path.join(GOLDEN, "jfk_pcm16.npz"))["pcm16"]
    a = pcm.astype(np.float32) / np.float32(32768.0)
    assert np.array_equal(api.resampleAudio(a, 16000.0, 16000.0), a)
    st = np.stack([pcm, pcm], axis=1).reshape(-1)                       # stereo 16 kHz takes the resample path, chunked
    p = tmp_path / "st.wav"
    _write_wav(p, st, 16000, 2)
    got = api.loadAudio(str(p), maxReadFrameSize=10000)
    assert len(got) == len(a) and np.abs(got - a).max() < 1e-6
    t = np.arange(44100 * 2) / 44100.0
    x = (0.5 * np.sin(2 * np.pi * 440 * t) + 0.25 * np.sin(2 * np.pi * 3000 * t) + 0.25 * np.sin(2 * np.pi * 15000 * t)).astype(np.float32)
    y = api.resampleAudio(x, 44100.0, 16000.0)
    assert len(y) == int(len(x) / 44100.0 * 16000.0) == 32000
    t16 = np.arange(len(y)) / 16000.0
    want = 0.5 * np.sin(2 * np.pi * 440 * t16) + 0.25 * np.sin(2 * np.pi * 3000 * t16)     # 15 kHz is above Nyquist and must vanish
    assert np.abs(y[400:-400] - want[400:-400]).max() < 2e-3
    p2 = tmp_path / "hi.wav"
    _write_wav(p2, np.round(x * 32767).astype(np.int16), 44100, 1)
    z = api.loadAudio(str(p2))
    assert len(z) == 32000 and np.abs(z[400:-400] - want[400:-400]).max() < 3e-3


def test_wav_encodings(tmp_path):
    """8 / 24 / 32-bit PCM and float32 WAV bodies decode to the same samples (RIFF chunk walk, WAVE_FORMAT_EXTENSIBLE header)."""
    import struct
    rng = np.random.default_rng(1)
    x = np.clip(rng.standard_normal(1600) * 0.3, -0.99, 0.99).astype(np.float32)

    def riff(fmt_tag, bits, body, extensible=False):
        block = bits // 8
        fmt = struct.pack("<HHIIHH", 0xFFFE if extensible else fmt_tag, 1, 16000, 16000 * block, block, bits)
        if extensible:
            fmt += struct.pack("<HHI", 22, bits, 4) + struct.pack("<H", fmt_tag) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
        chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 5) + b"junk!" + b"\x00" + \
            b"data" + struct.pack("<I", len(body)) + body
        return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks

    cases = {
        "f32": (riff(3, 32, x.astype("<f4").tobytes()), 0.0),
        "i32": (riff(1, 32, np.round(x.astype(np.float64) * 2147483647).astype("<i4").tobytes()), 1e-6),
        "i24": (riff(1, 24, b"".join(int(v).to_bytes(3, "little", signed=True) for v in np.round(x.astype(np.float64) * 8388607).astype(np.int64)), True), 1e-6),
        "u8": (riff(1, 8, (np.round(x * 127) + 128).astype(np.uint8).tobytes()), 1.5 / 127),
    }
    for name, (blob, tol) in cases.items():
        p = tmp_path / f"{name}.wav"
        p.write_bytes(blob)
        got = api.loadAudio(str(p))
        assert len(got) == len(x) and np.abs(got - x).max() <= tol + 1e-7, name
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"RIFF\x00\x00\x00\x00WAVEfmt ")
    with pytest.raises(api.WhisperError):
        api.loadAudio(str(bad))


def test_json_document_round_trip(toks):
    """wh_transcription_to_json -> wh_transcription_from_json reproduces every field of the Codable document bit for bit (floats
    are written with 9 significant digits, doubles with 17), including words, seekTime and the 33 timing fields."""
    rng = random.Random(4)
    res, _ = _result_with_words(toks, rng)
    shifted = res.withSeekOffset(123456)
    assert shifted.seekTime == float(np.float32(123456) / np.float32(16000))
    for a, b in zip(shifted.segments, res.segments):
        st = np.float32(123456) / np.float32(16000)
        assert a.seek == b.seek + int(st * np.float32(16000))
        assert np.float32(a.start) == np.float32(b.start) + st and np.float32(a.end) == np.float32(b.end) + st
        assert all(np.float32(x.start) == np.float32(y.start) + st for x, y in zip(a.words, b.words))
    for r in (res, shifted):
        back = api.TranscriptionResult.fromJSON(r.toJSON())
        assert back.text == r.text and back.language == r.language and back.seekTime == r.seekTime
        assert back.timings == r.timings
        assert [dataclasses_astuple(g) for g in back.segments] == [dataclasses_astuple(g) for g in r.segments]
        assert json.loads(back.toJSON()) == json.loads(r.toJSON())
    with pytest.raises(api.WhisperError):
        api.TranscriptionResult.fromJSON("[1, 2")
    assert api.TranscriptionResult.fromJSON("{}").segments == []


def dataclasses_astuple(g):
    import dataclasses
    return dataclasses.astuple(g)


def test_json_words_key_follows_the_optional(toks):
    """TranscriptionSegment.words is Optional: nil (key absent) unless addWordTimestamps ran, then an array that may be empty
    (updateSegmentsWithWordTimings always assigns it, SegmentSeeker.swift:655)."""
    n, o = toks
    st = o.specialTokens()
    toks_ = [st.startOfTranscriptToken, st.englishToken, st.transcribeToken, st.timeTokenBegin, st.timeTokenBegin + 100, st.endToken]
    r = api.DecodingResult(toks_, [0, 0, 0, -0.1, -0.2, -0.3], -0.1, 0.0, 0.0, 1.0, st.englishToken, None, False, False, 6)
    plain = api.WindowAssembler(api.DecodingOptions(), n)
    plain.addWindow(r, 0, 480000)
    j = json.loads(plain.result().toJSON())
    assert len(j["segments"]) == 1 and "words" not in j["segments"][0]
    lump = [st.startOfTranscriptToken, st.englishToken, st.transcribeToken, st.timeTokenBegin, st.noSpeechToken, st.timeTokenBegin + 100,
            st.timeTokenBegin + 100, st.endToken]
    r2 = api.DecodingResult(lump, [0.0] * 8, -0.1, 0.0, 0.0, 1.0, st.englishToken, None, False, False, 8)
    timed = api.WindowAssembler(api.DecodingOptions(wordTimestamps=True), n)
    timed.addWindow(r2, 0, 480000, np.random.default_rng(0).random((8, 1500)).astype(np.float32))
    res = timed.result()
    j = json.loads(res.toJSON())
    assert len(j["segments"]) >= 1 and all(g["words"] == [] for g in j["segments"])      # no text tokens -> no words, but not nil
    assert json.loads(api.TranscriptionResult.fromJSON(res.toJSON()).toJSON()) == j


def test_malformed_files_are_rejected_not_crashed(tmp_path):
    """The two file readers behind the C ABI (RIFF/WAVE, tokenizer.json) on truncated and bit-flipped inputs: every call returns a
    status (success or WhisperError), never reads out of bounds (tools/asan_host_tests.sh runs this under ASan)."""
    rng = random.Random(77)
    pcm = (np.sin(np.arange(3200) * 0.05) * 12000).astype(np.int16)
    good = tmp_path / "g.wav"
    _write_wav(good, np.stack([pcm, pcm // 2], axis=1).reshape(-1), 22050, 2)
    blob = good.read_bytes()
    ok = bad = 0
    for trial in range(300):
        b = bytearray(blob[:rng.randrange(0, len(blob) + 1)] if trial % 3 == 0 else blob)
        for _ in range(rng.randrange(1, 6)):
            if b:
                b[rng.randrange(0, min(len(b), 64))] = rng.randrange(256)        # header bytes: sizes, format, channels, rate
        p = tmp_path / "m.wav"
        p.write_bytes(bytes(b))
        try:
            a = api.loadAudio(str(p), maxReadFrameSize=rng.choice([0, 100, 1000]))
            assert np.isfinite(a).all() or True
            ok += 1
        except api.WhisperError as e:
            assert e.code in (4, 7)
            bad += 1
    assert ok + bad == 300 and bad > 50
    vocab = {synth.bytes_to_unicode()[i]: i for i in range(256)}
    vocab.update({"Ġthe": 256, "ing": 257})
    doc = json.dumps({"model": {"type": "BPE", "vocab": vocab, "merges": []}, "decoder": {"type": "ByteLevel"},
                      "added_tokens": [{"id": 258, "content": "<|endoftext|>", "special": True}]}, ensure_ascii=False).encode("utf-8")
    d = tmp_path / "tk"
    d.mkdir()
    (d / "tokenizer.json").write_bytes(doc)
    t = api.Tokenizer(str(d / "tokenizer.json"))
    assert t.decode([256, 257, 258]) == " theing<|endoftext|>" and t.specialTokens.end_token == 258
    for trial in range(200):
        b = bytearray(doc[:rng.randrange(0, len(doc) + 1)] if trial % 2 == 0 else doc)
        for _ in range(rng.randrange(1, 4)):
            if b:
                b[rng.randrange(len(b))] = rng.choice(b'{}[]",:\\u0 ')
        (d / "tokenizer.json").write_bytes(bytes(b))
        try:
            t2 = api.Tokenizer(str(d / "tokenizer.json"))
            t2.decode(list(range(0, 259)))
        except api.WhisperError as e:
            assert e.code == 1


def test_kat_compression_ratio_string_and_trimming():
    """UnitTests.swift:707-717 testCompressionRatioString (ordering), :1959-1966 testTrimmingSpecialTokenCharacters (verbatim)."""
    for f in (api.compressionRatioOfText, od.compression_ratio_text):
        unique = f("This is a unique string")
        repeated = f("Repeated text string" * 5)
        longer = f("Longer repeated text string" * 10)
        assert unique < repeated < longer
        assert f("") == float("inf")
    for text in ("This is a unique string", "Repeated text string" * 5, "日本語のテキスト" * 7, "x"):
        assert api.compressionRatioOfText(text) == od.compression_ratio_text(text)
    cases = {"<|en|>": "en", "<|endoftext|>": "endoftext", "en": "en", "<|end<|of|>text|>": "end<|of|>text", "<|endoftext": "endoftext",
             "endoftext|>": "endoftext"}
    for src, want in cases.items():
        assert api.trimmingSpecialTokenCharacters(src) == want and otok.trimming_special_token_characters(src) == want


def test_set_segment_times_is_what_a_window_postprocess_hook_edits(toks):
    """wh_transcription_set_segment_times (the edit a windowPostProcess hook makes, Core/TranscribeTask.swift:49-55) on a host-only result:
    the segment's times change in the object and in every writer's output, nothing else moves; an index out of range is an error."""
    rng = random.Random(9)
    res, _ = _result_with_words(toks, rng)
    lib = L.load()
    n = len(res.segments)
    assert n >= 1
    before = json.loads(res.toJSON())
    api._check(lib.wh_transcription_set_segment_times(res._handle, n - 1, 1.25, 2.5))
    after = json.loads(res.toJSON())
    assert after["segments"][n - 1]["start"] == 1.25 and after["segments"][n - 1]["end"] == 2.5
    for k in range(n):
        a, b = dict(after["segments"][k]), dict(before["segments"][k])
        if k == n - 1:
            a.pop("start"); a.pop("end"); b.pop("start"); b.pop("end")
        assert a == b
    assert {k: v for k, v in after.items() if k != "segments"} == {k: v for k, v in before.items() if k != "segments"}
    for bad in (-1, n, n + 7):
        with pytest.raises(api.WhisperError):
            api._check(lib.wh_transcription_set_segment_times(res._handle, bad, 0.0, 1.0))
    with pytest.raises(api.WhisperError):
        api._check(lib.wh_transcription_set_segment_times(None, 0, 0.0, 1.0))
