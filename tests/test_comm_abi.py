"""The multi-GPU step of the path behind the C ABI (wh_comm_*, include/whisperhip.h "multi-GPU"): chunk partition,
all-gather of fixed-size chunk records, gather + merge of whole TranscriptionResults - the reference's TaskGroup fan-out and in-process
merge (Core/WhisperKit.swift:735-812, Utilities/TranscriptionUtilities.swift:76-157).  CPU only: world sizes 2, 3 and 8 (configs[3]: 64 chunks over 8 ranks) over the library's
TCP transport, one process per rank, NO torch.distributed anywhere (a Swift / C host has none either).  The RCCL transport of the same
entry points runs in the -m gpu tests (tests/test_gpu_round3.py) and under bench.py --gpus N."""
import multiprocessing as mp
import socket

import numpy as np
import pytest

from test_parallel_gloo import _FakeSession, _long_audio, _summary
from whisperkit_amd import _lib as L
from whisperkit_amd import parallel


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_partition_abi_equals_python_rule():
    import ctypes as C
    lib = L.load()
    for n in (0, 1, 7, 8, 64, 65):
        for w in (1, 2, 3, 4, 8):
            for r in range(w):
                s, e = C.c_int(), C.c_int()
                assert lib.wh_partition_chunks(n, w, r, C.byref(s), C.byref(e)) == 0
                assert (s.value, e.value) == parallel.partition_chunks(n, w, r)
    assert lib.wh_partition_chunks(4, 2, 2, C.byref(s), C.byref(e)) != 0          # rank out of range -> invalid argument


def test_record_struct_is_the_packed_record():
    import ctypes as C
    assert C.sizeof(L.WhChunkRecord) == parallel.RECORD_INTS * 4 == 960
    r = parallel.pack_record(5, [50257, 50363, 11, 12, 50256], 32000, 223, -1.25, 0.2, 1.5, 0.0)
    q = L.WhChunkRecord.from_buffer_copy(r.tobytes())
    assert (q.n_tokens, q.chunk_index, q.seek, q.steps) == (5, 5, 32000, 223) and list(q.tokens[:5]) == [50257, 50363, 11, 12, 50256]
    assert q.avg_logprob == -1.25 and abs(q.temperature - 0.2) < 1e-7 and q.compression_ratio == 1.5
    # wh_chunk_record_from_result fills the same layout from a DecodingResult
    res = L.WhDecodingResult()
    res.n_tokens, res.steps, res.avg_logprob, res.temperature, res.compression_ratio = 3, 9, -0.5, 0.4, 2.0
    for i, t in enumerate((50257, 7, 50256)):
        res.tokens[i] = t
    out = L.WhChunkRecord()
    assert L.load().wh_chunk_record_from_result(C.byref(res), 11, 480000, C.byref(out)) == 0
    u = parallel.unpack_record(np.frombuffer(bytes(out), dtype=np.int32))
    assert u == dict(chunk_index=11, tokens=[50257, 7, 50256], seek=480000, steps=9, avg_logprob=-0.5, temperature=np.float32(0.4).item(),
                     compression_ratio=2.0, no_speech_prob=0.0)


def test_world_size_one_needs_no_peer():
    c = parallel.Comm(1, 0, transport="tcp")
    assert (c.world_size, c.rank, c.partition(5)) == (1, 0, (0, 5))
    recs = np.stack([parallel.pack_record(i, [7, i], 0, 1, 0.0, 0.0, 1.0) for i in (2, 0, 1)])
    assert [r["chunk_index"] for r in c.gather_records(recs, 3)] == [0, 1, 2]        # sorted by chunk index
    c.barrier()
    c.close()


def _records_worker(rank, world, port, n_chunks, q):
    c = parallel.Comm(world, rank, transport="tcp", tcp_address=f"127.0.0.1:{port}")
    s, e = c.partition(n_chunks)
    recs = np.stack([parallel.pack_record(i, [1000 + i, 7, i], i * 160, 3, -0.5 * i, 0.0, 1.0) for i in range(s, e)]) if e > s \
        else np.zeros((0, parallel.RECORD_INTS), np.int32)
    out = []
    for _ in range(3):                                  # the communicator is reused step after step (bench.py: one gather per step)
        out = c.gather_records(recs, (n_chunks + world - 1) // world)
    c.barrier()
    q.put((rank, [(r["chunk_index"], r["tokens"], r["seek"], r["avg_logprob"]) for r in out]))
    c.close()


@pytest.mark.parametrize("world,n_chunks", [(2, 5), (3, 7), (2, 1), (8, 64)])      # (8, 64): BASELINE configs[3], 64 chunks over 8 ranks
def test_gather_records_over_the_c_abi(world, n_chunks):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_records_worker, args=(r, world, port, n_chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [(i, [1000 + i, 7, i], i * 160, -0.5 * i) for i in range(n_chunks)]
    assert all(got[r] == expect for r in range(world))             # every rank sees every chunk, in chunk order


def _sharded_worker(rank, world, port, tok_path, q):
    from whisperkit_amd import api
    c = parallel.Comm(world, rank, transport="tcp", tcp_address=f"127.0.0.1:{port}")
    ordered, merged = parallel.transcribe_chunked_sharded(_FakeSession(api.Tokenizer(tok_path)), _long_audio(), comm=c)
    q.put((rank, _summary(ordered, merged)))
    c.barrier()
    c.close()


def test_whole_results_gathered_and_merged_over_the_c_abi(tmp_path):
    """Long audio over two ranks: same VAD chunks on every rank, contiguous chunk blocks, results shifted by the chunk offsets,
    gathered as the reference's Codable JSON through wh_comm_gather_transcriptions and merged with wh_merge_transcriptions - every
    rank ends with exactly what a single process computes."""
    from whisperkit_amd import api, synth
    tok_path = synth.write_kat_tokenizer(str(tmp_path), 51865)
    ordered, merged = parallel.transcribe_chunked_sharded(_FakeSession(api.Tokenizer(tok_path)), _long_audio())
    want = _summary(ordered, merged)
    port = _free_port()
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


# ---- failure behaviour: nothing hangs; a bad hello is dropped; an error on one rank is an error on every rank
def _timeout_worker(rank, world, port, token, q):
    import os
    os.environ["WH_COMM_TIMEOUT_S"] = "3"
    if token:
        os.environ["WH_COMM_TOKEN"] = token
    import time
    t0 = time.time()
    try:
        c = parallel.Comm(world, rank, transport="tcp", tcp_address=f"127.0.0.1:{port}")
        c.barrier()
        q.put((rank, "ok", time.time() - t0))
        c.close()
    except Exception as e:   # noqa: BLE001
        q.put((rank, "error: " + str(e)[:160], time.time() - t0))


def _run(target, argsets, timeout=60):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(*a, q)) for a in argsets]
    for p in procs:
        p.start()
    got = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    return {g[0]: g[1:] for g in got}


def test_missing_peer_is_an_error_after_the_deadline_not_a_hang():
    got = _run(_timeout_worker, [(0, 2, _free_port(), "")])
    status, dt = got[0]
    assert status.startswith("error") and "0 of 1 peers joined" in status and dt < 20, got


def test_wrong_job_token_is_refused_on_both_sides():
    port = _free_port()
    got = _run(_timeout_worker, [(0, 2, port, "job-a"), (1, 2, port, "job-b")])
    assert got[0][0].startswith("error") and got[1][0].startswith("error"), got
    assert max(got[0][1], got[1][1]) < 20


def _stray_then_join_worker(rank, world, port, q):
    import os
    import time
    os.environ["WH_COMM_TIMEOUT_S"] = "20"
    if rank == 1:                         # a stray connection first: four garbage bytes, then silence
        time.sleep(0.5)
        for _ in range(50):
            try:
                sk = socket.create_connection(("127.0.0.1", port), timeout=1)
                break
            except OSError:
                time.sleep(0.1)
        sk.sendall(b"\xff\xff\xff\x7f" + b"x" * 32)
        sk.close()
    c = parallel.Comm(world, rank, transport="tcp", tcp_address=f"127.0.0.1:{port}")
    c.barrier()
    q.put((rank, "ok", 0.0))
    c.close()


def test_bad_hello_is_dropped_and_the_real_peer_still_joins():
    port = _free_port()
    got = _run(_stray_then_join_worker, [(r, 2, port) for r in range(2)])
    assert got[0][0] == "ok" and got[1][0] == "ok", got


def _silent_then_join_worker(rank, world, port, q):
    import os
    import time
    os.environ["WH_COMM_TIMEOUT_S"] = "12"
    t0 = time.time()
    held = None
    if rank == 1:                         # a connection that says NOTHING and stays open (port scan, health probe, half-open peer) ...
        time.sleep(0.5)
        for _ in range(50):
            try:
                held = socket.create_connection(("127.0.0.1", port), timeout=1)
                break
            except OSError:
                time.sleep(0.1)
        time.sleep(0.3)                   # ... is in rank 0's accept loop before the real hello arrives
    c = parallel.Comm(world, rank, transport="tcp", tcp_address=f"127.0.0.1:{port}")
    c.barrier()
    q.put((rank, "ok", time.time() - t0))
    c.close()
    if held is not None:
        held.close()


def test_silent_connection_does_not_hold_the_accept_loop_for_the_join_deadline():
    """the hello of an accepted connection has its own short deadline (<= 3 s), so one mute connection costs seconds, not
    the whole WH_COMM_TIMEOUT_S (12 s here: the old code failed with '0 of 1 peers joined')."""
    port = _free_port()
    got = _run(_silent_then_join_worker, [(r, 2, port) for r in range(2)])
    assert got[0][0] == "ok" and got[1][0] == "ok", got
    assert max(got[0][1], got[1][1]) < 9.0, got


def _poison_worker(rank, world, port, tok_path, q):
    import ctypes as C
    from whisperkit_amd import api
    c = parallel.Comm(world, rank, transport="tcp", tcp_address=f"127.0.0.1:{port}")
    lib = c.lib
    ordered, _ = parallel.transcribe_chunked_sharded(_FakeSession(api.Tokenizer(tok_path)), _long_audio())      # local results (single process)
    handles = (C.c_void_p * 1)(ordered[0][1]._handle if rank == 0 else None)      # rank 1 has nothing it can serialise
    idx = (C.c_int32 * 1)(rank)
    out = (C.c_void_p * 8)(); oidx = (C.c_int32 * 8)(); n = C.c_int()
    rc = lib.wh_comm_gather_transcriptions(c.handle, handles, idx, 1, out, oidx, 8, C.byref(n))
    q.put((rank, rc, lib.wh_last_error().decode()[:120]))
    c.close()


def test_a_rank_that_cannot_serialise_fails_every_rank_together(tmp_path):
    from whisperkit_amd import synth
    tok_path = synth.write_kat_tokenizer(str(tmp_path), 51865)
    port = _free_port()
    got = _run(_poison_worker, [(r, 2, port, tok_path) for r in range(2)], timeout=120)
    assert got[0][0] != 0 and got[1][0] != 0, got                    # both return the error, nobody is left inside the collective
    assert "rank 1" in got[0][1] and "this rank" in got[1][1], got
