"""Round-6 GPU tests.

  * the batch entry points with one Result and one DecodingOptions PER AUDIO (VERDICT r05 "what's missing" 1):
    WhisperKit.transcribeWithOptions(audioArrays:decodeOptionsArray:) returns [Result<[TranscriptionResult], Error>] and an audio that fails
    is `.failure(error)` beside its neighbours' `.success` (Core/WhisperKit.swift:716-812, the catch at :786-790);
    transcribe(audioArrays:) maps the failures to nil (:660-688); the VAD-chunked path skips a failed chunk (Core/Audio/AudioChunker.swift:14-39);
  * a Python exception raised inside the progress callback comes out of EVERY entry point that can run it.
"""
import numpy as np
import pytest

from whisperkit_amd import api, weights
from whisperkit_amd.synth import synthetic_chunk

pytestmark = pytest.mark.gpu

NOFALLBACK = dict(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None, temperatureFallbackCount=0)


@pytest.fixture(scope="module")
def micro():
    dims = weights.MODEL_DIMS["test-micro"]
    model = api.Model(dims, weights.synthetic_state_dict(dims, seed=0))
    yield dims, model
    model.close()


def _same(a, b):
    return a.tokens == b.tokens and a.seeks == b.seeks and [(g.start, g.end, g.tokens) for g in a.segments] == [(g.start, g.end, g.tokens) for g in b.segments]


def test_one_result_per_audio_a_failing_audio_does_not_fail_its_neighbours(micro):
    dims, model = micro
    sess = api.Session(model, 4)
    a0, a1, a2 = synthetic_chunk(71), synthetic_chunk(72), np.concatenate([synthetic_chunk(73), synthetic_chunk(74)[:200000]])
    base = api.DecodingOptions(**NOFALLBACK, sampleLength=12)
    alone = [sess.transcribe([a], base)[0] for a in (a0, a1, a2)]
    # audio 1 is asked for a clip that ends beyond its samples: its first window is fine, the second one starts at sample 480000 = the buffer
    # size, where the reference's slice is empty, AudioProcessor.padOrTrimAudio returns nil and the task throws transcriptionFailed("Audio
    # samples are nil") - audio 1 fails, after having shared a device batch with the others for a whole window
    bad = api.DecodingOptions(**NOFALLBACK, sampleLength=12, clipTimestamps=(0.0, 35.0))
    res = sess.transcribeWithOptions([a0, a1, a2], [base, bad, base])
    assert isinstance(res[1], api.WhisperError) and res[1].code == 9 and "Audio samples are nil" in str(res[1])
    assert _same(res[0], alone[0]) and _same(res[2], alone[2])                 # bit-identical to the audios transcribed alone
    assert len(res[2].seeks) == 2
    # the reference's transcribe(audioArrays:) -> [[TranscriptionResult]?]: the 30 s audios cannot hold a 35 s clip, the 42.5 s audio can
    opt = sess.transcribe([a0, a1, a2], bad, optional=True)
    assert opt[0] is None and opt[1] is None
    assert opt[2] is not None and _same(opt[2], sess.transcribe([a2], bad)[0]) and opt[2].seeks[0] == 0 and max(opt[2].seeks) < 560000
    with pytest.raises(api.WhisperError):
        sess.transcribe([a0, a1, a2], bad)                                     # the default Python convenience raises the first failure
    # every audio failing: the per-audio entry point still returns (one Result each), the shared-options one reports the failure
    allbad = sess.transcribeWithOptions([a0, a1], [bad, bad])
    assert all(isinstance(r, api.WhisperError) for r in allbad)
    # the session is usable afterwards
    assert _same(sess.transcribe([a0], base)[0], alone[0])
    sess.close()


def test_one_decoding_options_per_audio(micro):
    """decodeOptionsArray: audios with different options run as different TranscribeTasks in the reference; here they form groups that share
    device batches - a result never depends on its neighbours' options."""
    dims, model = micro
    sess = api.Session(model, 4)
    xs = [synthetic_chunk(81 + i) for i in range(4)]
    o_short = api.DecodingOptions(**NOFALLBACK, sampleLength=8)
    o_long = api.DecodingOptions(**NOFALLBACK, sampleLength=20, withoutTimestamps=True)
    o_clip = api.DecodingOptions(**NOFALLBACK, sampleLength=8, clipTimestamps=(5.0,))         # differs from o_short by its clip only: same group
    opts = [o_short, o_long, o_clip, None]
    alone = [sess.transcribe([x], o or api.DecodingOptions())[0] for x, o in zip(xs, opts)]
    got = sess.transcribeWithOptions(xs, opts)
    assert all(not isinstance(r, api.WhisperError) for r in got)
    for g, a in zip(got, alone):
        assert _same(g, a)
    assert got[2].seeks[0] == 80000 and got[0].seeks[0] == 0                  # the clip start belongs to audio 2 alone
    assert len(got[1].tokens) > len(got[0].tokens)
    with pytest.raises(api.WhisperError):
        sess.transcribeWithOptions(xs, [o_short])                              # "must be balanced" (WhisperKit.swift:724-726)
    # hooks see the caller's audio index whatever group the audio ran in
    seen = []
    sess.setWindowHooks(windowPreprocess=lambda ai, x, seek, size: seen.append((ai, seek)))
    sess.transcribeWithOptions(xs, opts)
    sess.setWindowHooks()
    assert sorted({(ai, seek) for ai, seek in seen if seek in (0, 80000)}) == [(0, 0), (1, 0), (2, 80000), (3, 0)]
    assert {ai: [s_ for a_, s_ in seen if a_ == ai] for ai in range(4)} == {ai: got[ai].seeks for ai in range(4)}
    sess.close()


def test_chunked_transcription_over_the_per_audio_batch_entry_point(micro):
    """WhisperKit.transcribe(audioArray:) with .vad chunking hands the chunks to transcribeWithOptions and updateSeekOffsetsForResults keeps the
    `.success` chunks (`case .failure`: logged and skipped - wh_transcribe_chunked compacts its output the same way).  The chunks of a healthy
    audio all succeed: order and offsets as before the batch entry point learned per-audio results."""
    dims, model = micro
    sess = api.Session(model, 4)
    audio = np.concatenate([synthetic_chunk(91), synthetic_chunk(92), synthetic_chunk(93)[:100000]])
    opts = api.DecodingOptions(**NOFALLBACK, sampleLength=10)
    chunks = sess.transcribeChunked(audio, opts)
    assert len(chunks) >= 2 and [o for o, _ in chunks] == sorted(o for o, _ in chunks) and chunks[0][0] == 0
    cuts = api.vadChunkAll(audio, options=opts)
    assert [o for o, _ in chunks] == [c0 for c0, _ in cuts]
    for (off, r), (c0, c1) in zip(chunks, cuts):
        alone = sess.transcribe([audio[c0:c1]], opts)[0]
        assert r.tokens == alone.tokens
    sess.close()


def test_callback_exception_comes_out_of_every_entry_point_that_runs_the_callback(micro):
    dims, model = micro
    sess = api.Session(model, 5)
    sess.padOrTrim(synthetic_chunk(95)); sess.logMelSpectrogram(1); sess.encodeFeatures(1); sess.prepareDecoderInputs(1)
    opts = api.DecodingOptions(**NOFALLBACK, sampleLength=24)
    prompt = sess.prefillPrompt(opts)

    class Boom(Exception):
        pass

    def cb(*_a):
        raise Boom("progress callback failed")
    sess.setProgressCallback(cb)
    with pytest.raises(Boom):
        sess.decodeText(prompt, opts)
    sess.prepareDecoderInputs(1)
    with pytest.raises(Boom):
        sess.decodeTextCustom(prompt, opts)
    with pytest.raises(Boom):
        sess.transcribe([synthetic_chunk(95)], opts)
    # nothing is left behind for a later, unrelated call
    sess.setProgressCallback(None)
    sess.padOrTrim(synthetic_chunk(95)); sess.logMelSpectrogram(1); sess.encodeFeatures(1); sess.prepareDecoderInputs(1)
    assert len(sess.decodeText(prompt, opts)[0].tokens) > 4
    sess.detectLanguage(1)
    sess.close()


@pytest.mark.parametrize("name,fname,modes", [("test-large-v3-l2", "hf_model_large_v3_l2.npz", (0, 1)), ("test-small-l2", "hf_model_small_l2.npz", (0, 1)),
                                              ("test-tiny-en-l2", "hf_model_tiny_en_l2.npz", (0,))], ids=["large-v3-width", "small-width", "tiny.en-width"])
def test_benchmarked_widths_against_the_hf_golden_logits_and_cross_attention_weights(jfk_pcm, name, fname, modes):
    """The device at the headline width (d = 1280, 20 heads, 128 mel, V = 51866; 2 + 2 layers) against the HF-transformers golden itself
    (tests/golden/hf_model_large_v3_l2.npz, written by tests/golden/make_golden.py; the oracle is pinned to the same file on the CPU):
    end to end from the PCM of jfk.wav - encoder rows, teacher-forced logits within BASELINE's 1e-3, and the alignment rows (mean of the
    two alignment heads' cross-attention weights: what DecodingCache.alignmentWeights carries, Core/TextDecoder.swift:272-296) within 1e-4.
    Both cross-attention modes of the library.  Round 6, last session: the same at the widths of BASELINE configs[2] (d = 768, 12 heads, 80 mel, V = 51865: both
    modes) and configs[1] (d = 384, 6 heads, V = 51864: K / V rows, the only mode at that width)."""
    from conftest import golden
    g = golden(fname)
    dims = weights.MODEL_DIMS[name]
    heads = [tuple(int(v) for v in h) for h in g["heads"]]
    model = api.Model(dims, weights.synthetic_state_dict(dims, seed=0), alignment_heads=heads)
    es, ls, xs = int(g["enc_stride"]), int(g["logit_stride"]), int(g["xatt_stride"])
    toks = [int(t) for t in g["tokens"]]
    report = {}
    for mode in modes:
        sess = api.Session(model, 2, crossAttentionMode=mode)
        assert sess.crossAttentionMode == mode
        for b in range(2):
            sess.padOrTrim(jfk_pcm, b)
        sess.logMelSpectrogram(2); sess.encodeFeatures(2); sess.prepareDecoderInputs(2)
        enc = sess.getEncoderOutput(1)
        e_enc = float(np.abs(enc[::es] - g["enc"]).max())
        e_log = 0.0
        for pos, tok in enumerate(toks):
            lg = sess.predictLogits([tok, tok], [pos, pos])
            np.testing.assert_array_equal(lg[0], lg[1])                       # the two slots carry the same audio
            e_log = max(e_log, float(np.abs(lg[1][::ls] - g["logits"][pos]).max()))
        al = sess.getAlignmentWeights(1)
        e_al = max(float(np.abs(al[pos + 1, ::xs] - g["xatt"][:, pos].mean(0)).max()) for pos in range(len(toks)))
        report[mode] = (e_enc, e_log, e_al)
        sess.close()
    model.close()
    print(name, "HF golden: mode -> (encoder rows, logits, alignment rows) max abs err", report)
    for mode, (e_enc, e_log, e_al) in report.items():
        assert e_enc <= 5e-3 and e_log <= 1e-3 and e_al <= 1e-4, report


def test_persistent_gemm_tile_loop_is_bit_identical_to_one_workgroup_per_tile():
    """csrc/gemm.hip, round 6: gemm256p_kernel (WH_GEMM_PERSIST=1: one workgroup per CU loops over tiles, the next tile's first K-tile is requested
    under the epilogue, the staged epilogues move to [64 KB, 160 KB) of the LDS) must produce the encoder output of gemm256_kernel bit for bit -
    ragged M, a partial 256-column tile at width 384, every staged epilogue (tools/enc_epi_ab.py, quick cases)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for persist in ("0", "1"):
        env = dict(os.environ, PYTHONPATH=root, WH_GEMM_PERSIST=persist, WH_EPI_AB_QUICK="1")
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "enc_epi_ab.py")], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        got[persist] = {k: v for k, v in json.loads(p.stdout.strip().splitlines()[-1]).items() if k != "mode"}
    assert got["0"] == got["1"] and len(got["0"]) >= 4, got


def test_cross_attention_slots_per_workgroup_do_not_change_a_bit():
    """wh_session_options.cross_attention_slots_per_workgroup (round 6): a workgroup of the absorbed cross-attention streams n slots one after the
    other, so the launch takes ceil(batch / n) x splits workgroups.  A slot is processed exactly as by a workgroup of its own: tokens, log-probs,
    teacher-forced logits and alignment rows of a ragged 70-slot batch (three batch tiles; 35 / 24 / 14 first-pass slots at n = 2 / 3 / 5) are the
    bits of n = 1, at the headline width (d = 1280, 20 heads; 2 + 2 layers) with 2 key splits per slot."""
    dims = weights.MODEL_DIMS["test-large-v3-l2"]
    heads = [(0, 3), (1, 17)]
    model = api.Model(dims, weights.synthetic_state_dict(dims, seed=5), alignment_heads=heads)
    B = 70
    xs = [synthetic_chunk(300 + b) for b in range(B)]
    opts = api.DecodingOptions(**NOFALLBACK, sampleLength=20, wordTimestamps=True)
    ref = None
    for n in (1, 2, 3, 5):
        sess = api.Session(model, B, crossAttentionMode=1, crossAttentionSplits=2, crossAttentionSlotsPerWorkgroup=n)
        assert sess.crossAttentionSlotsPerWorkgroup == n and sess.crossAttentionSplits == 2
        for b, x in enumerate(xs):
            sess.padOrTrim(x, b)
        sess.logMelSpectrogram(B); sess.encodeFeatures(B); sess.prepareDecoderInputs(B)
        prompt = sess.prefillPrompt(opts)
        res = sess.decodeText(prompt, opts, batch=B)
        got = ([r.tokens for r in res], [r.tokenLogProbs for r in res], [sess.getAlignmentWeights(b)[:20].copy() for b in (0, 34, 35, 69)])
        sess.resetDecoderInputs(B)
        lg = sess.predictLogits([res[b].tokens[0] for b in range(B)], [0] * B)
        got = got + (lg.copy(),)
        if ref is None:
            ref = got
        else:
            assert got[0] == ref[0] and got[1] == ref[1], n
            for a, b_ in zip(got[2], ref[2]):
                np.testing.assert_array_equal(a, b_)
            np.testing.assert_array_equal(got[3], ref[3])
        sess.close()
    with pytest.raises(api.WhisperError):
        api.Session(model, 4, crossAttentionMode=1, crossAttentionSlotsPerWorkgroup=17)
    model.close()


@pytest.mark.parametrize("slots,spw", [(224, 2), (112, 1), (192, 2), (256, 2), (160, 2)])
def test_device_batch_shapes_of_the_multi_gpu_runs_decode_every_slot_like_a_lone_session(slots, spw):
    """bench.py at 2 / 4 GPUs packs 7 steps of 32 / 16 chunks into device batches of 224 / 112 slots (1 key split; 2 / 1 slots per cross-attention
    workgroup), at one GPU a run's shorter batches have 192 slots, the full ones 256 x 2.  None of these has a full-depth rig of its own; what makes
    the 128 x 1 and 256 x 1 x 2 rigs of tests/test_gpu_fulldepth.py speak for them is bit-identity: with one key split a slot's results do not depend on
    the batch it sits in, nor on the slots per workgroup.  Checked here at the headline width (d = 1280, 20 heads; 2 + 2 layers, absorbed
    cross-attention forced with 1 split): slots of the first, a middle and the last batch tile and both sides of the workgroup's slot boundary
    against ONE-slot sessions of the same audio - encoder output, tokens, log-probs, bit for bit.
    The same comparison pins the grouped projection kernels (csrc/decoder32.hip, round 6): from five batch tiles on (160 slots: the threshold case) a
    projection workgroup handles two weight-row tiles (from four tiles on: the 112-slot case; 160 slots = the threshold of the four-tile form), the qkv / fc1 / fc2 projections four; the ONE-slot sessions run the one-tile kernels."""
    dims = weights.MODEL_DIMS["test-large-v3-l2"]
    model = api.Model(dims, weights.synthetic_state_dict(dims, seed=7))
    b1 = -(-slots // spw)
    check = sorted({0, 31, 32, b1 - 1, min(b1, slots - 1), slots - 33, slots - 1})
    xs = {b: synthetic_chunk(500 + b) for b in check}
    filler = synthetic_chunk(499)
    opts = api.DecodingOptions(**NOFALLBACK, sampleLength=16)
    big = api.Session(model, slots, crossAttentionMode=1, crossAttentionSplits=1, crossAttentionSlotsPerWorkgroup=spw)
    for b in range(slots):
        big.padOrTrim(xs.get(b, filler), b)
    big.logMelSpectrogram(slots); big.encodeFeatures(slots); big.prepareDecoderInputs(slots)
    prompt = big.prefillPrompt(opts)
    res = big.decodeText(prompt, opts, batch=slots)
    for b in check:
        one = api.Session(model, 1, crossAttentionMode=1, crossAttentionSplits=1)
        one.padOrTrim(xs[b], 0)
        one.logMelSpectrogram(1); one.encodeFeatures(1); one.prepareDecoderInputs(1)
        np.testing.assert_array_equal(one.getEncoderOutput(0), big.getEncoderOutput(b))
        r1 = one.decodeText(prompt, opts)[0]
        assert r1.tokens == res[b].tokens and r1.tokenLogProbs == res[b].tokenLogProbs, (slots, spw, b)
        one.close()
    big.close(); model.close()


def test_automatic_key_splits_follow_the_session_size_and_equal_split_counts_give_equal_bits():
    """csrc/xabs.hip xabs_auto_splits (round 6): an absorbed session created without a split count takes slots x splits within one round of the 256 CUs (4 / 3 / 2 / 1
    splits up to 64 / 85 / 128 / 256 slots; K / V-row sessions report 0).  The split count fixes the order of the key-split combine, so a 100-slot session (2 splits on its
    own) decodes every slot like a ONE-slot session that asks for 2 splits - bit for bit - and within the logits contract of a one-slot session with its own 4."""
    dims = weights.MODEL_DIMS["test-large-v3-l2"]
    model = api.Model(dims, weights.synthetic_state_dict(dims, seed=7))
    for B in (28, 64, 65, 86, 129):
        s = api.Session(model, B, crossAttentionMode=1)
        assert s.crossAttentionSplits == api.Session.xabsAutoSplits(B) == max(1, min(4, 256 // B)), B
        s.close()
    s = api.Session(model, 20, crossAttentionMode=0)
    assert s.crossAttentionSplits == 0
    s.close()
    slots, check = 100, [0, 31, 32, 99]
    xs = {b: synthetic_chunk(700 + b) for b in check}
    filler = synthetic_chunk(699)
    opts = api.DecodingOptions(**NOFALLBACK, sampleLength=16)
    big = api.Session(model, slots)                      # automatic mode (absorbed from 28 slots) and automatic splits
    assert big.crossAttentionMode == 1 and big.crossAttentionSplits == 2
    for b in range(slots):
        big.padOrTrim(xs.get(b, filler), b)
    big.logMelSpectrogram(slots); big.encodeFeatures(slots); big.prepareDecoderInputs(slots)
    prompt = big.prefillPrompt(opts)
    res = big.decodeText(prompt, opts, batch=slots)
    for b in check:
        one = api.Session(model, 1, crossAttentionMode=1, crossAttentionSplits=2)
        one.padOrTrim(xs[b], 0)
        one.logMelSpectrogram(1); one.encodeFeatures(1); one.prepareDecoderInputs(1)
        r1 = one.decodeText(prompt, opts)[0]
        assert r1.tokens == res[b].tokens and r1.tokenLogProbs == res[b].tokenLogProbs, b
        one.close()
        four = api.Session(model, 1, crossAttentionMode=1)
        assert four.crossAttentionSplits == 4
        four.padOrTrim(xs[b], 0)
        four.logMelSpectrogram(1); four.encodeFeatures(1); four.prepareDecoderInputs(1)
        r4 = four.decodeText(prompt, opts)[0]
        assert r4.tokens == res[b].tokens, b
        np.testing.assert_allclose(r4.tokenLogProbs, res[b].tokenLogProbs, rtol=0, atol=1e-3)
        four.close()
    big.close(); model.close()
