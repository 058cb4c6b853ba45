"""N>1 path on CPU: world_size-2 gloo run of the chunk partition + result all-gather used by bench.py / multi-GPU runs."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from whisperkit_amd import parallel


def test_partition_is_contiguous_and_complete():
    for n in (1, 7, 8, 64, 65):
        for w in (1, 2, 4, 8):
            parts = [parallel.partition_chunks(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [e - s for s, e in parts]
            assert max(sizes) - min(sizes) <= 1


def test_record_roundtrip():
    r = parallel.pack_record(5, [50257, 50363, 11, 12, 50256], 32000, 223, -1.25, 0.2, 1.5, 0.0)
    u = parallel.unpack_record(r)
    assert u["chunk_index"] == 5 and u["tokens"] == [50257, 50363, 11, 12, 50256] and u["seek"] == 32000 and u["steps"] == 223
    assert abs(u["avg_logprob"] + 1.25) < 1e-7 and abs(u["temperature"] - 0.2) < 1e-7


def _worker(rank, world, port, n_chunks, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, e = parallel.partition_chunks(n_chunks, world, rank)
    recs = np.stack([parallel.pack_record(i, [1000 + i, 7, i], i * 160, 3, -0.5 * i, 0.0, 1.0) for i in range(s, e)]) if e > s \
        else np.zeros((0, parallel.RECORD_INTS), np.int32)
    out = parallel.gather_records(recs, (n_chunks + world - 1) // world)
    q.put((rank, [(r["chunk_index"], r["tokens"], r["seek"]) for r in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_records_world2_gloo():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_chunks, world = 5, 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [(i, [1000 + i, 7, i], i * 160) for i in range(n_chunks)]
    assert got[0] == expect and got[1] == expect       # every rank sees every chunk, in chunk order


# ---------------------------------------------------------------------------------------------- whole results across ranks
class _FakeSession:
    """Stands in for api.Session on a CPU-only host: 'transcribes' a chunk into segments that depend only on its samples, through
    the same host-side result container (wh_transcription_create) the GPU path fills."""
    B = 2

    def __init__(self, tokenizer):
        self.tokenizer = tokenizer

    def transcribe(self, arrays, options):
        from whisperkit_amd import api
        out = []
        for a in arrays:
            k = int(abs(float(a[:1000].sum())) * 1000) % 97 + 3
            toks = [50364] + [400 + (k * j) % 300 for j in range(4)] + [13, 50364 + 50 + k]
            seg = api.TranscriptionSegment(0, 0, 0.0, (50 + k) * 0.02, toks, [-0.01 * j for j in range(len(toks))], 0.0, -0.3, 1.1, 0.0, [])
            out.append(api.makeTranscriptionResult([seg], self.tokenizer, languageToken=50259,
                                                   timings={"input_audio_seconds": len(a) / 16000.0, "full_pipeline": 0.5, "pipeline_start": 10.0,
                                                            "total_decoding_windows": 1, "total_decoding_loops": len(toks)}))
        return out


def _long_audio():
    from whisperkit_amd.synth import synthetic_chunk
    gap = np.zeros(24000, np.float32)
    return np.concatenate([synthetic_chunk(91)[:400000], gap, synthetic_chunk(92)[:350000], gap, synthetic_chunk(93)[:320000], gap,
                           synthetic_chunk(94)[:300000]])


def _summary(ordered, merged):
    return ([(off, r.seekTime, r.text, [(g.id, g.seek, g.start, g.end, g.tokens, g.text) for g in r.segments]) for off, r in ordered],
            merged.text, [(g.id, g.start, g.end, g.tokens) for g in merged.segments], merged.timings["input_audio_seconds"],
            merged.timings["total_decoding_loops"])


def _sharded_worker(rank, world, port, tok_path, q):
    import torch.distributed as dist
    from whisperkit_amd import api
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ordered, merged = parallel.transcribe_chunked_sharded(_FakeSession(api.Tokenizer(tok_path)), _long_audio())
    q.put((rank, _summary(ordered, merged)))
    dist.barrier()
    dist.destroy_process_group()


def test_transcribe_chunked_sharded_world2_equals_single_process(tmp_path):
    """Long audio over two ranks (gloo): same VAD chunks on every rank, contiguous chunk blocks, results shifted by the chunk
    offsets and gathered as Codable JSON; every rank ends with the result a single process computes."""
    from whisperkit_amd import api, synth
    tok_path = synth.write_kat_tokenizer(str(tmp_path), 51865)
    ordered, merged = parallel.transcribe_chunked_sharded(_FakeSession(api.Tokenizer(tok_path)), _long_audio())
    want = _summary(ordered, merged)
    assert len(ordered) >= 3 and [off for off, _ in ordered] == [c0 for c0, _ in api.vadChunkAll(_long_audio())]
    assert all(r.seekTime == float(np.float32(off) / np.float32(16000)) for off, r in ordered)
    assert all(g.start >= off / 16000 - 1e-4 for off, r in ordered for g in r.segments) and merged.text.count(".") == len(ordered)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, tok_path, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == want and got[1] == want
