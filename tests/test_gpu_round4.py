"""Round-4 GPU tests: the 8-rank strong-scaling rehearsal on one GPU under the driver's launch line (VERDICT r03 "next round" 6) and the
failure behaviour of the RCCL transport of wh_comm_* when a communicator cannot form."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, nproc, port, timeout=900):
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--model", "tiny.en", "--batch", "8", "--inflight", "1", "--steps", "8", "--warmup", "1",
            "--no-cpu-baseline", "--no-roofline", "--no-other-configs", "--sample-length", "32"] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_eight_ranks_on_one_gpu_strong_scaling_rehearsal(tmp_path):
    """BASELINE configs[3]'s shape (a step's chunks block-partitioned over 8 ranks, G = 8 consecutive steps packed into one device batch
    per rank, the per-step result records all-gathered through wh_comm_*) executed with 8 processes on the ONE GPU of the box, under the
    driver's launch line `python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8` (the library's TCP transport: RCCL
    refuses duplicate devices).  The gathered records of a step must equal the records of the same step computed by one rank.  This
    is a rehearsal of the control flow, not a measurement: no 1 -> 8 GPU curve exists."""
    one = str(tmp_path / "one.json")
    eight = str(tmp_path / "eight.json")
    l1 = _bench(["--dump-records", one], 1, 0)
    l8 = _bench(["--dump-records", eight, "--single-device", "--dist-backend", "gloo"], 8, 29581)
    assert l8["n_gpus"] == 8 and l8["scaling"] == "strong" and l8["config"]["chunks_per_step"] == 8 and l8["config"]["chunks_per_gpu"] == 1
    assert l8["config"]["steps_per_device_batch"] == 8 and l8["config"]["device_batch_slots"] == 8
    assert l8["config"]["result_gather"].startswith("wh_comm_gather_records"), l8["config"]["result_gather"]
    assert l1["config"]["chunks_per_gpu"] == 8 and l1["config"]["steps_per_device_batch"] == 1
    r1, r8 = json.load(open(one)), json.load(open(eight))
    assert [r["chunk_index"] for r in r8] == list(range(8))
    assert [(r["chunk_index"], r["tokens"], r["steps"]) for r in r1] == [(r["chunk_index"], r["tokens"], r["steps"]) for r in r8]


_DUP = r"""
import os, sys, time
sys.path.insert(0, %r)
from whisperkit_amd import parallel
rank = int(sys.argv[1]); path = sys.argv[2]
def exchange(raw):
    if rank == 0:
        open(path + ".tmp", "wb").write(raw); os.replace(path + ".tmp", path); return raw
    for _ in range(600):
        if os.path.exists(path): return open(path, "rb").read()
        time.sleep(0.05)
    raise SystemExit("no id")
t0 = time.time()
try:
    c = parallel.Comm(2, rank, transport="rccl", device=0, exchange_id=exchange)
    print("CREATED", flush=True)
    c.close()
except Exception as e:
    print("STATUS", round(time.time() - t0, 1), str(e)[:200].replace("\n", " "), flush=True)
"""


def test_rccl_communicator_that_cannot_form_fails_with_a_status(tmp_path):
    """Two ranks that both name GPU 0 (a rank / device mismatch: the only multi-rank RCCL call a one-GPU box can make): ncclCommInitRank
    must come back with an error that wh_comm_create turns into a status and a message - not a hang.  Every rank also refuses an RCCL
    communicator without an explicit device (parallel.Comm)."""
    from whisperkit_amd import parallel
    with pytest.raises(ValueError):
        parallel.Comm(2, 0, transport="rccl", exchange_id=lambda raw: raw)
    script = tmp_path / "dup.py"
    script.write_text(_DUP % ROOT)
    idf = str(tmp_path / "id.bin")
    env = dict(os.environ, PYTHONPATH=ROOT, NCCL_DEBUG="WARN")
    ps = [subprocess.Popen([sys.executable, str(script), str(r), idf], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in ps:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out = "HANG"
        outs.append(out)
    assert all("STATUS" in o for o in outs), outs          # an error status on both ranks, no hang, no communicator
    assert all("ncclCommInitRank failed" in o or "RCCL" in o for o in outs), outs


def test_window_hooks_of_the_native_orchestrator():
    """TranscribeTask.windowPreprocess / windowPostProcess / segmentDiscoveryCallback (Core/TranscribeTask.swift:42-55,130,246,260) as
    hooks of wh_transcribe*: the pre-process hook sees every window's samples before the pipeline runs, segment discovery sees exactly
    the segments that end up in the result, a post-process hook that drops a window's last segment (and shifts the first one's
    times) removes its tokens from the result as well."""
    import numpy as np
    from whisperkit_amd import api, weights
    from whisperkit_amd.synth import synthetic_chunk
    dims = weights.MODEL_DIMS["test-micro"]
    model = api.Model(dims, weights.synthetic_state_dict(dims, seed=0))
    audio = np.concatenate([synthetic_chunk(61), synthetic_chunk(62), synthetic_chunk(63)[:240000]])
    opts = api.DecodingOptions(sampleLength=14, firstTokenLogProbThreshold=None, compressionRatioThreshold=None, logProbThreshold=None,
                               temperatureFallbackCount=0)
    sess = api.Session(model, 1)
    plain = sess.transcribe([audio], opts)[0]
    assert len(plain.seeks) == 3 and len(plain.segments) >= 3
    pre, found = [], []
    sess.setWindowHooks(windowPreprocess=lambda ai, x, seek, size: pre.append((ai, seek, size, float(x[0]), float(x[-1]))),
                        segmentDiscovery=lambda ai, segs: found.extend(segs))
    hooked = sess.transcribe([audio], opts)[0]
    assert hooked.tokens == plain.tokens and hooked.seeks == plain.seeks                     # observers change nothing
    assert [(p[0], p[1]) for p in pre] == [(0, s) for s in plain.seeks]
    for (_, seek, size, x0, x1) in pre:
        assert size == min(480000, len(audio) - seek) and x0 == float(audio[seek]) and x1 == float(audio[seek + size - 1])
    assert [(g.seek, g.tokens, g.start, g.end) for g in found] == [(g.seek, g.tokens, g.start, g.end) for g in plain.segments]
    by_window = {}
    for g in plain.segments:
        by_window.setdefault(g.seek, []).append(g)

    def post(ai, seek, size, segs, set_times):
        assert [g.tokens for g in segs] == [g.tokens for g in by_window[seek]]
        set_times(0, segs[0].start + 100.0, segs[0].end + 100.0)
        return max(len(segs) - 1, 1)                                                            # drop the window's last segment (keep at least one)
    found.clear()
    sess.setWindowHooks(windowPostProcess=post, segmentDiscovery=lambda ai, segs: found.extend(segs))
    cut = sess.transcribe([audio], opts)[0]
    want = [g for w in by_window.values() for g in w[:max(len(w) - 1, 1)]]
    assert [g.tokens for g in cut.segments] == [g.tokens for g in want] == [g.tokens for g in found]
    assert cut.tokens == [t for g in want for t in g.tokens]                                  # the dropped segments' tokens are gone too
    firsts = {w[0].seek: w[0] for w in by_window.values()}
    for g in cut.segments:
        if g.tokens == firsts[g.seek].tokens:
            assert g.start == pytest.approx(firsts[g.seek].start + 100.0) and g.end == pytest.approx(firsts[g.seek].end + 100.0)
    # a hook that raises: the exception must come out of transcribe() itself (ctypes used to swallow it and the C side read an
    # uninitialised `keep`, silently dropping segments), and the session must be usable afterwards
    class Boom(Exception):
        pass

    def bad_post(*_a):
        raise Boom("windowPostProcess failed")
    sess.setWindowHooks(windowPostProcess=bad_post)
    with pytest.raises(Boom):
        sess.transcribe([audio], opts)
    sess.setWindowHooks(segmentDiscovery=lambda *_a: (_ for _ in ()).throw(Boom("segmentDiscovery failed")))
    with pytest.raises(Boom):
        sess.transcribe([audio], opts)
    sess.setWindowHooks()
    assert sess.transcribe([audio], opts)[0].tokens == plain.tokens
    sess.close(); model.close()


@pytest.mark.parametrize("splits", [1, 2, 3, 4])
def test_absorbed_cross_attention_key_split_counts_vs_oracle(splits):
    """wh_session_create_tuned: the absorbed cross-attention with 1 .. 4 key splits per slot (slots x splits workgroups, one per CU:
    the share of the chip a session's cross-attention takes).  Every count against the oracle (logits <= 1e-3, alignment rows
    <= 1e-4) at large-v3's width and 20 heads (two head tiles), 9 slots (a ragged last group of 4 slots); the getter reports the count;
    counts outside 0 .. 4 are refused."""
    import numpy as np
    from oracle import decode as OD
    from oracle.model import OracleWhisper
    from whisperkit_amd import api, weights
    from whisperkit_amd.synth import synthetic_chunk
    dims = weights.MODEL_DIMS["test-large-v3-l2"]
    sd = weights.synthetic_state_dict(dims, seed=11)
    model = api.Model(dims, sd)
    if splits == 4:
        import ctypes
        for bad in (-1, 5):
            h = ctypes.c_void_p()
            with pytest.raises(api.WhisperError):
                api._check(model.lib.wh_session_create_tuned(model.handle, 4, 1, bad, ctypes.byref(h)))
        s0 = api.Session(model, 4, crossAttentionMode=0, crossAttentionSplits=2)
        assert s0.crossAttentionMode == 0 and s0.crossAttentionSplits == 0
    B = 9
    sess = api.Session(model, B, crossAttentionMode=1, crossAttentionSplits=splits)
    assert sess.crossAttentionMode == 1 and sess.crossAttentionSplits == splits
    xs = [synthetic_chunk(4000 + 17 * b) for b in range(B)]
    for b in range(B):
        sess.padOrTrim(xs[b], b)
    sess.logMelSpectrogram(B); sess.encodeFeatures(B); sess.prepareDecoderInputs(B)
    om = OracleWhisper(dims, sd)
    st, _ = OD.special_tokens_for_vocab(dims.n_vocab)
    check = [0, 4, B - 1]
    states = {b: om.new_state(sess.getEncoderOutput(b).astype(np.float16).astype(np.float32)) for b in check}
    steps = [(st.startOfTranscriptToken, 0), (st.englishToken, 1), (st.transcribeToken, 2), (1029, 3), (400, 150), (77, 222)]
    for t, pos in steps:
        got = sess.predictLogits([(t + 3 * b) % 50000 if pos > 2 else t for b in range(B)], [pos] * B)
        for b in check:
            ref = states[b].step(int((t + 3 * b) % 50000 if pos > 2 else t), pos)
            e = float(np.abs(got[b] - ref).max())
            assert e <= 1e-3, (splits, b, pos, e)
    rows = [p + 1 for _, p in steps if p + 1 < 224]
    for b in check:
        al = sess.getAlignmentWeights(b)
        assert np.abs(al[rows] - states[b].alignment[rows]).max() <= 1e-4, (splits, b)


_CAP = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from whisperkit_amd import api, weights
from whisperkit_amd.synth import synthetic_chunk
dims = weights.MODEL_DIMS["test-small-l2"]
model = api.Model(dims, weights.synthetic_state_dict(dims, seed=3))
sess = api.Session(model, 4)
for b in range(4):
    sess.padOrTrim(synthetic_chunk(900 + b), b)
sess.logMelSpectrogram(4); sess.encodeFeatures(4); sess.prepareDecoderInputs(4)
opts = api.DecodingOptions(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None, noSpeechThreshold=None,
                           temperatureFallbackCount=0, sampleLength=20)          # 19 decoder steps = 3 step graphs per configuration
prompt = sess.prefillPrompt(opts)
first, counts = {}, []
for batch in (1, 2, 3, 4, 1, 2, 4, 3, 1):
    res = sess.decodeText(prompt, opts, batch=batch)
    got = [(list(r.tokens), [float(x) for x in r.tokenLogProbs]) for r in res]
    if batch in first:
        assert got == first[batch], ("a configuration re-captured after eviction must give the same results", batch)
    first[batch] = got
    counts.append(sess.stepGraphCount)
print("COUNTS", counts)
assert max(counts) <= 6, counts            # WH_GRAPH_CAP=6: two configurations of three graphs
assert counts[0] >= 2 and counts[1] > counts[0] and counts[-1] == 6, counts     # grows to the cap, stays there
"""


def test_step_graph_cache_is_capped_least_recently_used_configuration_first(tmp_path):
    """a session holds up to 28 step graphs per (batch, alignment, sampler) configuration.  The cache is capped
    (WH_GRAPH_CAP, default 112): beyond it the configuration used longest ago is dropped whole, after the stream has drained; a
    configuration captured again later gives bit-identical results."""
    script = tmp_path / "cap.py"
    script.write_text(_CAP % ROOT)
    p = subprocess.run([sys.executable, str(script)], env=dict(os.environ, WH_GRAPH_CAP="6"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "COUNTS" in p.stdout
