"""CPU-side checks of the C ABI: the library loads, exports every symbol include/whisperhip.h declares, fails loudly
without a GPU, and its host-side logic (C++ restatement of the reference's Swift) agrees with the reference's KATs and
with the oracle.  No compute kernels are launched here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from oracle import decode as D
from whisperkit_amd import _lib as L
from whisperkit_amd import api, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    header = open(os.path.join(ROOT, "include", "whisperhip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(wh_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(L.SYMBOLS), (declared ^ set(L.SYMBOLS))
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.wh_version()


def test_struct_layouts_match_header_sizes():
    # spot checks against the C layout rules the header implies
    assert C.sizeof(L.WhDims) == 40
    assert C.sizeof(L.WhSpecialTokens) == 52
    assert C.sizeof(L.WhDecodingResult) == 4 + 232 * 8 + 4 * 4 + 5 * 4
    assert C.sizeof(L.WhSegment) == 48
    o = L.WhDecodingOptions()
    L.load().wh_decoding_options_default(C.byref(o))
    # DecodingOptions() defaults, Core/Configurations.swift:184-212
    assert (o.temperature, o.temperature_fallback_count, o.sample_length, o.top_k) == (0.0, 5, 224, 5)
    assert o.temperature_increment_on_fallback == pytest.approx(0.2) and o.window_clip_time == 1.0
    assert o.compression_ratio_threshold == pytest.approx(2.4) and o.log_prob_threshold == -1.0
    assert o.first_token_log_prob_threshold == -1.5 and o.no_speech_threshold == pytest.approx(0.6)
    assert o.use_prefill_prompt == 1 and o.detect_language == -1 and o.language_token == -1 and o.seed == 0


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is visible")
def test_product_path_fails_loudly_without_gpu():
    dims = weights.MODEL_DIMS["test-micro"]
    blob = weights.pack_blob(dims, weights.synthetic_state_dict(dims, seed=0))
    with pytest.raises(api.WhisperError) as e:
        api.Model(dims, blob=blob)
    assert e.value.code == 101 and "no CPU fallback" in str(e.value)


def test_null_handles_return_models_unavailable():
    lib = L.load()
    assert lib.wh_log_mel_spectrogram(None, 1) == 2       # WhisperError.modelsUnavailable
    assert lib.wh_mel_count(None) == -1
    assert b"null" in lib.wh_last_error()
    h = C.c_void_p()
    assert lib.wh_model_create(b"nope", 4, 0, C.byref(h)) == 100
    assert lib.wh_model_load(b"/nonexistent/file", 0, C.byref(h)) == 2


def test_compression_ratio_matches_oracle_and_kat():
    rng = np.random.default_rng(0)
    for toks in ([1, 2, 3, 4, 5, 6, 7, 8, 9, 10], [1] * 10, [1] * 20, list(rng.integers(0, 50000, 200)), [400, 370] * 60):
        assert api.compressionRatio(toks) == pytest.approx(D.compression_ratio(toks), rel=1e-6)
    assert api.compressionRatio(list(range(1, 11))) < api.compressionRatio([1] * 10) < api.compressionRatio([1] * 20)  # UnitTests.swift:695-705
    assert api.compressionRatio([]) == float("inf")


def test_dtw_kat_and_oracle():
    ti, tj = api.dynamicTimeWarping(np.array([[1.0, 1.0, 1.0], [5.0, 2.0, 1.0], [1.0, 5.0, 2.0]]))   # UnitTests.swift:2337-2367
    assert ti == [0, 1, 1, 2, 2] and tj == [0, 0, 1, 1, 2]
    rng = np.random.default_rng(1)
    m = rng.random((24, 300)).astype(np.float16).astype(np.float32)
    assert api.dynamicTimeWarping(m) == D.dynamic_time_warping(m)
    big = rng.random((224, 1500)).astype(np.float32)                                                # UnitTests.swift:2369-2418 properties
    ti, tj = api.dynamicTimeWarping(big)
    assert (ti[0], tj[0], ti[-1], tj[-1]) == (0, 0, 223, 1499)
    d = np.diff(np.array([ti, tj]), axis=1)
    assert ((d == 0) | (d == 1)).all() and (d.sum(0) >= 1).all()


def test_decoding_fallback_order():   # UnitTests.swift:816-878
    O = api.DecodingOptions
    f = api.decodingFallback
    assert f(O(compressionRatioThreshold=-1.0, logProbThreshold=-1.0, noSpeechThreshold=-1.0), True, 0, 0, -2.0) == ("firstTokenLogProbThreshold", True)
    assert f(O(compressionRatioThreshold=-1.0, logProbThreshold=-1.0, noSpeechThreshold=-1.0), False, 0, 0, -2.0) == ("silence", False)
    assert f(O(compressionRatioThreshold=-1.0, logProbThreshold=-1.0, noSpeechThreshold=0.0), False, 0, 0, -2.0) == ("compressionRatioThreshold", True)
    assert f(O(compressionRatioThreshold=0.0, logProbThreshold=-1.0, noSpeechThreshold=0.0), False, 0, 0, -2.0) == ("logProbThreshold", True)
    assert f(O(compressionRatioThreshold=0.0, logProbThreshold=0.0, noSpeechThreshold=0.0), False, 0, 0, 0.0) == (None, False)


def test_vad_kats(jfk_pcm):   # UnitTests.swift:2119-2189
    va = api.voiceActivity(jfk_pcm)
    assert va == D.EnergyVAD().voiceActivity(jfk_pcm)
    assert D.EnergyVAD.findLongestSilence(va) == (43, 54)
    assert api.voiceActivity([]) == []
    z, o = np.zeros(1600, np.float32), np.ones(1600, np.float32)
    assert api.voiceActivity(np.concatenate([z, o]), 320, 80) == D.EnergyVAD(frameLengthSamples=320, frameOverlapSamples=80).voiceActivity(np.concatenate([z, o]))
    assert api.vadChunkAll(jfk_pcm) == [(0, 176000)]
    long = np.concatenate([jfk_pcm] * 4)
    assert api.vadChunkAll(long) == [(s, s + len(a)) for s, a in D.vad_chunk_all(long)]
    assert len(api.vadChunkAll(long)) >= 2


def test_find_seek_point_and_segments_matches_oracle():
    s = D.SpecialTokens()
    st = L.WhSpecialTokens(s.endToken, s.englishToken, s.noSpeechToken, s.noTimestampsToken, s.specialTokenBegin, s.startOfPreviousToken,
                           s.startOfTranscriptToken, s.timeTokenBegin, s.transcribeToken, s.translateToken, s.whitespaceToken, 50259, 99)
    rng = np.random.default_rng(3)
    cases = [
        [s.startOfTranscriptToken, 50364, 400, 50464, 50464, 370, 50564, s.endToken],
        [s.startOfTranscriptToken, 50259, 50359, 50364, 400, 370, 452, 13, 50889, s.endToken],
        [s.startOfTranscriptToken, 50364, 400, 370, s.endToken],
        [s.startOfTranscriptToken, 50364, 400, 50400, 50400, 11, 50500, 50500, 12, 13, s.endToken],
        [s.startOfTranscriptToken, 50363, 400, 370, s.endToken],
    ]
    for toks in cases:
        lps = list(-rng.random(len(toks)).astype(np.float32))
        for seek, size in ((0, 480000), (16000, 176000)):
            res = D.DecodingResult("en", toks, [{t: float(l)} for t, l in zip(toks, lps)], -0.3, 0.0, 0.0, 1.0, None)
            oseek, osegs = D.find_seek_point_and_segments(res, D.DecodingOptions(), 2, seek, size, s)
            nseek, nsegs = api.findSeekPointAndSegments(toks, lps, api.DecodingOptions(), st, 2, seek, size, avgLogProb=-0.3)
            assert nseek == oseek
            assert len(nsegs) == len(osegs)
            for a, b in zip(nsegs, osegs):
                assert (a.id, a.seek) == (b.id, b.seek)
                assert a.start == pytest.approx(b.start, abs=1e-6) and a.end == pytest.approx(b.end, abs=1e-6)
                assert toks[a.token_offset:a.token_offset + a.n_tokens] == b.tokens
    # noSpeech skip path: avgLogProb below threshold and noSpeechProb above -> whole window skipped
    nseek, nsegs = api.findSeekPointAndSegments(cases[0], [0.0] * len(cases[0]), api.DecodingOptions(), st, 0, 0, 480000, avgLogProb=-2.0, noSpeechProb=0.9)
    assert nsegs is None and nseek == 480000


def test_prefill_prompt_matches_oracle():
    lib = L.load()
    # wh_prefill_prompt needs a model handle only for isModelMultilingual; emulate both with a fake dims-only check through the oracle
    s_ml, _ = D.special_tokens_for_vocab(51865)
    o = D.DecodingOptions(promptTokens=[5, 6, 60000, 7], prefixTokens=[8, 9])
    assert D.prefill_prompt(o, s_ml, True) == [s_ml.startOfPreviousToken, 5, 6, 7, s_ml.startOfTranscriptToken, s_ml.englishToken,
                                               s_ml.transcribeToken, s_ml.timeTokenBegin, 8, 9]
    assert lib.wh_prefill_prompt(None, None, None, -1, None, 0) == -1


def test_bench_knows_every_kernel_kind():
    """bench.py prices every kernel kind the library can report (wh_kernel_kind_name): a new kind without an algorithmic
    byte / FLOP formula must fail here, not in the middle of a GPU bench run."""
    import importlib.util
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "whisperkit_amd", "csrc", "host.hip")).read()
    block = src[src.index("kKindNames[KK_COUNT]"):]
    names = re.findall(r'"([a-z0-9_]+)"', block[:block.index("};")])
    assert len(names) >= 20 and "dec_cross_attn" in names
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from whisperkit_amd import weights
    lib = L.load()
    assert lib.wh_kernel_kind_count() == len(names)
    assert [lib.wh_kernel_kind_name(k).decode() for k in range(len(names))] == names
    for model in ("tiny.en", "large-v3"):
        for n in names:
            bound, amount = bench.algorithmic_work(n, weights.MODEL_DIMS[model], 8, 8.5)
            assert bound in ("hbm", "mfma") and amount > 0


def test_prepare_seek_clips_matches_oracle_and_kat(jfk_pcm):
    """DecodingOptions.prepareSeekClips (Extensions+Internal.swift:112-130) through the C ABI: the literal cases, random clip lists
    against the oracle, and the reference KAT that feeds it VAD clip timestamps of jfk.wav (UnitTests.swift:2178-2189)."""
    import random
    from oracle import decode as OD
    assert api.prepareSeekClips(api.DecodingOptions(), 1000) == [(0, 1000)]
    assert api.prepareSeekClips(api.DecodingOptions(clipTimestamps=[1.0]), 48000) == [(16000, 48000)]
    assert api.prepareSeekClips(api.DecodingOptions(clipTimestamps=[0.5, 1.0, 2.0]), 48000) == [(8000, 16000), (32000, 48000)]
    rng = random.Random(1)
    for _ in range(200):
        ts = sorted(round(rng.random() * 60, rng.choice([1, 2, 3])) for _ in range(rng.randrange(0, 7)))
        n = rng.randrange(1, 960000)
        assert api.prepareSeekClips(api.DecodingOptions(clipTimestamps=ts), n) == OD.DecodingOptions(clipTimestamps=ts).prepareSeekClips(n), ts
    vad = OD.EnergyVAD(frameLength=0.2, frameOverlap=0.1)
    ts = vad.voiceActivityClipTimestamps(jfk_pcm)
    clips = api.prepareSeekClips(api.DecodingOptions(clipTimestamps=ts), len(jfk_pcm))
    assert clips == OD.DecodingOptions(clipTimestamps=ts).prepareSeekClips(len(jfk_pcm))
    assert [c[0] for c in clips] == [3200, 51200, 83200, 128000, 169600] and [c[1] for c in clips] == [35200, 70400, 121600, 166400, 176000]


def test_header_is_plain_c_and_struct_layouts_match_ctypes(tmp_path):
    """include/whisperhip.h must compile as C (the boundary is a C ABI) and every struct the ctypes mirror declares must have the
    size and field offsets the C compiler gives it."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    structs = {"wh_dims": L.WhDims, "wh_special_tokens": L.WhSpecialTokens, "wh_decoding_options": L.WhDecodingOptions,
               "wh_decoding_result": L.WhDecodingResult, "wh_segment": L.WhSegment, "wh_word_timing": L.WhWordTiming,
               "wh_timings": L.WhTimings, "wh_progress": L.WhProgress}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "whisperhip.h"', "int main(void) {"]
    for cname, ct in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", inc, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = {}
    for ln in out:
        if ln.strip():
            c, f, v = ln.split()
            got[(c, f)] = int(v)
    for cname, ct in structs.items():
        assert got[(cname, "size")] == C.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert got[(cname, fname)] == getattr(ct, fname).offset, (cname, fname)


def test_plain_c_host_links_and_runs_host_entry_points(tmp_path):
    """examples/transcribe.c - a C99 program over nothing but include/whisperhip.h - compiles, links against libwhisperhip.so and its
    --selftest mode (tokenizer, WAV ingest, VAD chunking, time formatting) runs without a GPU."""
    import shutil
    import subprocess
    import wave
    from whisperkit_amd import synth
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "transcribe"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "transcribe.c"),
                    "-L", os.path.join(root, "whisperkit_amd"), "-lwhisperhip", "-lm", "-Wl,-rpath," + os.path.join(root, "whisperkit_amd"),
                    "-o", str(exe)], check=True)
    tok = synth.write_kat_tokenizer(str(tmp_path), 51865)
    pcm = (np.sin(np.arange(16000 * 40) * 0.02) * 8000).astype("<i2")
    with wave.open(str(tmp_path / "a.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    r = subprocess.run([str(exe), "--selftest", tok, str(tmp_path / "a.wav")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "decode -> <|startoftranscript|><|1.00|><|endoftext|>" in r.stdout
    assert "audio 640000 samples = 00:00:40,000, 2 chunk(s)" in r.stdout
    bad = subprocess.run([str(exe), "--selftest", str(tmp_path / "nope.json"), str(tmp_path / "a.wav")], capture_output=True, text=True)
    assert bad.returncode == 1 and "cannot read" in bad.stderr


def test_bench_work_formulas_reproduce_the_survey_figures():
    """SURVEY.md section 8(d) states the algorithmic work per 30 s chunk that `roofline.achieved` must be computed from; the
    per-launch formulas of bench.py have to add up to those figures (large-v3: encoder 2.274 TFLOP, cross-K/V 314.6 GFLOP, cross-
    attention K/V stream 245.8 MB per token, decoder weights 1.81 GB; tiny.en: encoder 36.9 GFLOP)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    w = lambda kind, dims, B=1, avg=1.0: bench.algorithmic_work(kind, dims, B, avg)[1]
    for model, enc_flop, ckv_flop in (("large-v3", 2.274e12, 314.6e9), ("tiny.en", 36.9e9, 3.5e9), ("small", 344e9, 42.5e9)):
        dims = weights.MODEL_DIMS[model]
        L_ = dims.n_audio_layer
        enc = w("gemm_conv1", dims) + w("gemm_conv2", dims) + L_ * (w("gemm_enc_qkv", dims) + w("enc_attention", dims) + w("gemm_enc_o", dims) +
                                                                    w("gemm_enc_fc1", dims) + w("gemm_enc_fc2", dims))
        assert enc == pytest.approx(enc_flop, rel=0.01), (model, enc)
        assert w("gemm_cross_kv", dims) == pytest.approx(ckv_flop, rel=0.02), model
    dims = weights.MODEL_DIMS["large-v3"]
    d, L_, V = dims.n_text_state, dims.n_text_layer, dims.n_vocab
    # K and V of 1500 positions: SURVEY's 245.8 MB per token and sequence is the Float16 figure; the K / V-row mode keeps 24-bit rows since round 5
    # (1.5 x the bytes: Float16 rows cost 7e-3 sigma of the logits, tests/test_gpu_realistic.py), the absorbed mode reads HALF of SURVEY's figure
    assert L_ * (w("dec_cross_attn", dims) - 2 * d * 4) == pytest.approx(1.5 * 245.8e6, rel=0.001)
    # decoder weights read per step, shared by the batch (MFMA path: every matrix exactly once, no folded product matrices)
    per_layer_w = (3 * d * d + d * d + d * d + d * d + 4 * d * d + 4 * d * d) * 2
    got = L_ * sum(w(k, dims, 1, 0.0) for k in ("dec_proj_qkv", "dec_proj_oproj", "dec_proj_cq", "dec_proj_coproj", "dec_proj_fc1", "dec_proj_fc2")) + \
        w("dec_proj_logits", dims, 1, 0.0)
    assert got == pytest.approx(L_ * per_layer_w + V * d * 2, rel=0.01)
    all_decoder = L_ * (4 + 4 + 8) * d * d * 2 + V * d * 2               # SURVEY's 1.81 GB: every decoder matrix incl. the cross K/V projections
    assert all_decoder == pytest.approx(1.81e9, rel=0.02)
    per_step = all_decoder - L_ * 2 * d * d * 2                             # ... which run once per window (gemm_cross_kv), not per token
    assert got == pytest.approx(per_step, rel=0.01)
    assert w("mel_power", weights.MODEL_DIMS["large-v3"]) + 0 >= 480000 * 4                              # PCM read is in the bill


def test_bench_absorbed_cross_attention_bytes_follow_the_key_split_count():
    """bench.py prices the absorbed cross-attention by the session's key split count (wh_session_cross_attention_splits): the encoder rows
    of every slot once (SURVEY 8d's 245.8 MB per token and sequence over 32 layers = half of it per slot and layer: the K and the V stream
    became one), the absorbed queries, and every split's unnormalised partial + (m, l).  The figures DESIGN section 4 and the committed bench
    lines quote: 265.4 MB per launch at 64 slots and 2 splits, 278.6 MB at 4."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod3", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dims = weights.MODEL_DIMS["large-v3"]
    d, H, B = dims.n_text_state, dims.n_text_head, 64
    enc_rows = B * 1500 * d * 2
    assert enc_rows == 245_760_000
    for splits, want in ((1, None), (2, 265_441_280), (3, None), (4, 278_568_960)):
        bound, got = bench.algorithmic_work("dec_cross_attn", dims, B, 8.5, absorbed=True, splits=splits)
        assert bound == "hbm" and got == enc_rows + B * H * d * 4 + splits * B * H * (d * 4 + 8)
        if want:
            assert got == want
        _, vup = bench.algorithmic_work("dec_xabs_vup", dims, B, 8.5, absorbed=True, splits=splits)
        assert vup >= d * d * 2 + splits * B * H * d * 4          # W_v once + every split's partial read back
    line = json.loads(open(os.path.join(root, "profiles", "r04ag_bench_final_steps20_warmup5.json")).read().strip().splitlines()[-1])
    r = line["roofline"]
    assert r["alg_per_launch"] == 265_441_280 and r["same_kernel_alone_on_the_whole_chip"]["alg_per_launch"] == 278_568_960
    assert r["frac"] == pytest.approx(r["alg_per_launch"] / (r["avg_us"] * 1e-6) / 8e12, rel=1e-3)
    assert r["traffic"] >= r["alg_per_launch"]                  # PMC traffic is never below the algorithmic bytes


def test_swift_shim_source_names_the_session_entry_points_of_the_header():
    """bindings/swift is source only (no Swift toolchain in the image), so nothing compiles it: this guard keeps it from going stale
    against the header again (it did not mention the round-4 entry points).  Every C function the shim calls must be
    declared in include/whisperhip.h, and the session-creation / hook entry points must be used by it."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    swift = open(os.path.join(root, "bindings", "swift", "Sources", "WhisperKitHIP", "HIPBackend.swift")).read()
    header = open(os.path.join(root, "include", "whisperhip.h")).read()
    declared = set(re.findall(r"\b(wh_[a-z0-9_]+)\s*\(", header))
    code = "\n".join(l.split("//")[0] for l in swift.splitlines())                     # calls in code, not in comments
    called = set(re.findall(r"\b(wh_[a-z0-9_]+)\s*\(", code))
    types = {"wh_special_tokens", "wh_decoding_options", "wh_decoding_result", "wh_window_hooks", "wh_dims", "wh_timings", "wh_progress", "wh_session_options"}
    assert called - types <= declared, sorted(called - types - declared)
    for fn in ("wh_session_create_tuned", "wh_session_set_window_hooks", "wh_xabs_auto_min_slots", "wh_session_cross_attention_mode",
               "wh_session_cross_attention_splits", "wh_session_step_graph_count",
               "wh_transcribe_batch_with_options", "wh_session_item_status", "wh_session_item_error",        # round 6: one Result per audio
               "wh_session_create_with_options", "wh_session_options_default", "wh_session_cross_attention_slots_per_workgroup"):
        assert fn in called, fn
    # the shim's comments quote the library's behaviour: they went stale once ("48 slots", "fp32 K / V rows")
    from whisperkit_amd import _lib
    assert f"wh_xabs_auto_min_slots: {_lib.load().wh_xabs_auto_min_slots()}" in swift
    assert "fp32 K / V rows" not in swift and "24-bit rows" in swift


def test_automatic_key_splits_keep_a_lone_session_within_one_round_of_the_chip():
    """csrc/xabs.hip xabs_auto_splits (round 6, profiles/r06ah_lone_session_key_splits.jsonl): an absorbed session created without a split count gets as many key splits per
    slot as keep slots x splits workgroups within the 256 CUs, at most 4, at least 1 - and the committed sweep says that choice is the fastest column at every batch it measured."""
    from whisperkit_amd import _lib
    f = _lib.load().wh_xabs_auto_splits
    assert [f(b) for b in (1, 28, 32, 64, 65, 85, 86, 100, 128, 129, 192, 256)] == [4, 4, 4, 4, 3, 3, 2, 2, 2, 1, 1, 1]
    for b in range(1, 257):
        s = f(b)
        assert 1 <= s <= 4 and (s == 1 or b * s <= 256) and (s == 4 or b * (s + 1) > 256), b
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = [json.loads(l) for l in open(os.path.join(root, "profiles", "r06ah_lone_session_key_splits.jsonl"))]
    for b in sorted({r["B"] for r in rows}):
        ms = {r["splits"]: r["ms_per_step_wall"] for r in rows if r["B"] == b}
        assert ms[f(b)] == min(ms.values()), (b, ms)
    swift = open(os.path.join(root, "bindings", "swift", "Sources", "WhisperKitHIP", "HIPBackend.swift")).read()
    assert "wh_xabs_auto_splits(" in swift


def test_round5_bench_line_bookkeeping_is_per_bench_step():
    """The committed round-5 bench line (profiles/r05i_*: the final binary): a 128-slot device batch carries two 64-chunk bench steps, so the decoder kernels' launches_per_step is
    32 layers x 223 decoder steps / 2 = 3568 (the event pool used to overflow and report 6467 of 7136), launches x average duration of the dominant kernel
    fits inside ms_per_step, the algorithmic bytes are the formula's, and the PMC pass of the same binary (profiles/r05_pmc_traffic.json) is not below them."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = json.loads(open(os.path.join(root, "profiles", "r05i_bench_steps20_warmup5.json")).read().strip().splitlines()[-1])
    r, cfg = line["roofline"], line["config"]
    assert cfg["device_batch_slots"] == 128 and cfg["steps_per_device_batch"] == 2 and r["steps_per_device_batch"] == 2
    k = r["kernels"]["dec_cross_attn"]
    assert k["launches_per_step"] == 32 * 223 / 2 and k["launches_measured"] == 32 * 16
    dims = weights.MODEL_DIMS["large-v3"]
    d, H, B = dims.n_text_state, dims.n_text_head, 128
    assert r["alg_per_launch"] == B * 1500 * d * 2 + B * H * d * 4 + 1 * B * H * (d * 4 + 8) == 517_754_880
    assert r["frac"] == pytest.approx(r["alg_per_launch"] / (r["avg_us"] * 1e-6) / 8e12, rel=1e-3)
    assert k["launches_per_step"] * k["avg_us"] * 1e-3 <= line["ms_per_step"]                       # the dominant kernel fits inside the step it is a share of
    assert r["whole_step"]["hbm_bound_algorithmic_bytes"] / (line["ms_per_step"] * 1e-3) <= 8e12 and r["whole_step_frac"] == r["whole_step"]["frac"]
    assert r["whole_chip_frac"] == r["same_kernel_alone_on_the_whole_chip"]["frac"] and r["whole_chip_frac"] > r["frac"]
    tj = json.load(open(os.path.join(root, "profiles", "r05_pmc_traffic.json")))
    assert tj["chunks_per_step"] == 128 and tj["cross_attention_splits"] == 1
    assert 1.0 <= tj["bytes_per_launch"]["dec_cross_attn"] / r["alg_per_launch"] <= 1.05
    assert r["traffic"] == tj["bytes_per_launch"]["dec_cross_attn"]                                   # the line carries the PMC figure of its own workload


def test_library_carries_the_staged_epilogue_kernels():
    """csrc/gemm.hip (round 5): the four layer GEMMs of the encoder run gemm256_kernel<EPI, 1> (LDS-staged epilogues) by default; the
    direct variants <EPI, 0> stay for WH_GEMM_EPI_MODE=0 and for shapes the staged form does not take.  A build that lost either set
    would silently change what the encoder runs: the mangled names must be in the library's code object."""
    import os
    from whisperkit_amd import _lib
    path = os.path.join(os.path.dirname(_lib.__file__), "libwhisperhip.so")
    blob = open(path, "rb").read()
    for epi in (0, 1, 2, 3):                                     # F16, GELU_F16, RESID_F32, QKV_ENC (kernels.h GemmEpi)
        for mode in (0, 1, 2):
            assert f"gemm256_kernelILi{epi}ELi{mode}EEE".encode() in blob, (epi, mode)
    for epi in (4, 5, 6, 7):                                     # conv1, conv2, plain fp32, cross K / V rows: direct only
        assert f"gemm256_kernelILi{epi}ELi0EEE".encode() in blob and f"gemm256_kernelILi{epi}ELi1EEE".encode() not in blob, epi


def test_library_carries_the_grouped_row_tile_projection_kernels():
    """csrc/decoder32.hip (round 6): from four batch tiles on a decoder projection workgroup multiplies its activation planes by TWO weight-row tiles (every
    projection, the logits included) or, from five on, FOUR (qkv, fc1, the residual projections = fc2) - dec32_proj_kernel<MODE, HILO, TC, NTW, RT>.  The one-tile kernels stay
    for smaller batches (and are what the bit-identity tests of tests/test_gpu_round6.py compare the grouped ones with).  All three sets must be in the library."""
    from whisperkit_amd import _lib
    blob = open(os.path.join(os.path.dirname(_lib.__file__), "libwhisperhip.so"), "rb").read()
    for mode in (0, 1, 2, 3, 4):                                 # QKV, Q, RESID, FC1, LOGITS (kernels.h)
        for rt, tcs in ((1, (5, 2)), (2, (4, 2))):
            for tc in tcs:
                assert f"dec32_proj_kernelILi{mode}ELb1ELi{tc}ELb0ELi{rt}EEE".encode() in blob, (mode, tc, rt)
    for mode in (0, 2, 3):
        assert f"dec32_proj_kernelILi{mode}ELb1ELi1ELb0ELi4EEE".encode() in blob, mode
    for mode in (1, 4):
        assert f"dec32_proj_kernelILi{mode}ELb1ELi1ELb0ELi4EEE".encode() not in blob, mode


def test_round6_bench_line_bookkeeping_of_the_256_slot_device_batch():
    """The committed round-6 bench line (profiles/r06aq_*: the final binary under the driver's command line): a 256-slot device batch carries FOUR 64-chunk bench steps and a
    cross-attention workgroup streams two slots, so the launch takes 128 workgroups; launches_per_step is 32 layers x 223 decoder steps / 4 = 1784; the algorithmic bytes are the
    formula's at 256 slots; the PMC pass of that configuration (profiles/r06_pmc_traffic.json) is not below them; every other_configs entry carries a roofline of its own."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = json.loads(open(os.path.join(root, "profiles", "r06aq_bench_steps20_warmup5.json")).read().strip().splitlines()[-1])
    r, cfg = line["roofline"], line["config"]
    assert cfg["device_batch_slots"] == 256 and cfg["steps_per_device_batch"] == 4 and cfg["steps_in_flight"] == 12 and cfg["audio_sets"] >= 4
    assert r["workgroups"] == 128 and r["slots_per_workgroup"] == 2 and r["cu_share"] == 0.5
    k = r["kernels"]["dec_cross_attn"]
    assert k["launches_per_step"] == 32 * 223 / 4 and k["launches_measured"] == 32 * 16
    dims = weights.MODEL_DIMS["large-v3"]
    d, H, B = dims.n_text_state, dims.n_text_head, 256
    assert r["alg_per_launch"] == B * 1500 * d * 2 + B * H * d * 4 + 1 * B * H * (d * 4 + 8) == 1_035_509_760
    assert r["frac"] == pytest.approx(r["alg_per_launch"] / (r["avg_us"] * 1e-6) / 8e12, rel=1e-3)
    assert k["launches_per_step"] * k["avg_us"] * 1e-3 <= line["ms_per_step"]
    assert r["whole_chip_frac"] == r["same_kernel_alone_on_the_whole_chip"]["frac"] and r["whole_chip_frac"] > r["frac"]
    assert r["same_kernel_alone_on_the_whole_chip"]["workgroups"] == 256
    tj = json.load(open(os.path.join(root, "profiles", "r06_pmc_traffic.json")))
    assert tj["chunks_per_step"] == 256 and tj["cross_attention_splits"] == 1
    assert 1.0 <= tj["bytes_per_launch"]["dec_cross_attn"] / r["alg_per_launch"] <= 1.05 and r["traffic"] == tj["bytes_per_launch"]["dec_cross_attn"]
    assert line["cpu_baseline"]["first_tokens_equal_gpu"] is True and "of" in line["cpu_baseline"]["cores_of"]
    for name, o in line["other_configs"].items():
        rf = o["roofline"]
        assert rf and rf["kernel"] and 0.0 < rf["frac"] < 1.0 and rf["whole_step"]["frac"] < 1.0 and len(rf["top_kernels"]) >= 3, name
