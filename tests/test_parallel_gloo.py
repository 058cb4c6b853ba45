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
