"""Two-rank rehearsal of the multi-GPU path on ONE GPU (launched by tests/test_gpu_round2.py through torch.distributed.run with
the gloo backend; every rank drives GPU 0 through its own model + session).  Checks, on rank 0:
  * chunk records: the gathered 2 x B records (parallel.gather_records) equal what one process computes for all 2 B chunks;
  * whole results: parallel.transcribe_chunked_sharded at world size 2 equals the single-process transcribeChunked + merge
    (the reference's TaskGroup fan-out + mergeTranscriptionResults, Core/WhisperKit.swift:735-812,
    Utilities/TranscriptionUtilities.swift:76-157).
Prints "REHEARSAL OK" on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from whisperkit_amd import api, parallel, weights  # noqa: E402
from whisperkit_amd.synth import synthetic_chunk  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dims = weights.MODEL_DIMS["test-micro"]
    model = api.Model(dims, weights.synthetic_state_dict(dims, seed=0), device=0)
    B = 3
    opts = api.DecodingOptions(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None,
                               temperatureFallbackCount=0, sampleLength=16)

    def records(sess, first, n):
        for b in range(n):
            sess.padOrTrim(synthetic_chunk(9000 + first + b), b)
        sess.logMelSpectrogram(n); sess.encodeFeatures(n); sess.prepareDecoderInputs(n)
        res = sess.decodeText(sess.prefillPrompt(opts), opts, batch=n)
        return np.stack([parallel.pack_record(first + b, r.tokens, 0, r.steps, r.avgLogProb, r.temperature, r.compressionRatio) for b, r in enumerate(res)])

    sess = api.Session(model, B)
    first, _ = parallel.partition_chunks(world * B, world, rank)
    mine = records(sess, first, B)
    got = parallel.gather_records(mine, B)
    assert [r["chunk_index"] for r in got] == list(range(world * B)), got
    # the same step behind the C ABI (wh_comm_*): partition, record gather, result gather + merge without torch.distributed on the data path
    # (TCP transport: RCCL refuses two ranks on one device; the RCCL transport of the same calls runs at world size 1 in test_gpu_round3.py)
    comm = parallel.Comm(world, rank, transport="tcp", tcp_address=f"127.0.0.1:{int(os.environ.get('MASTER_PORT', '29500')) + 23}")
    assert comm.partition(world * B) == parallel.partition_chunks(world * B, world, rank)
    assert comm.gather_records(mine, B) == got
    # long audio across ranks
    gap = np.zeros(24000, np.float32)
    audio = np.concatenate([synthetic_chunk(91)[:400000], gap, synthetic_chunk(92)[:350000], gap, synthetic_chunk(93)[:320000], gap, synthetic_chunk(94)[:300000]])
    topts = api.DecodingOptions(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None,
                                temperatureFallbackCount=0, sampleLength=10)
    s4 = api.Session(model, 4)
    ordered, merged = parallel.transcribe_chunked_sharded(s4, audio, topts)
    ordered_c, merged_c = parallel.transcribe_chunked_sharded(s4, audio, topts, comm=comm)
    assert [(o, r.tokens) for o, r in ordered_c] == [(o, r.tokens) for o, r in ordered] and merged_c.tokens == merged.tokens
    assert [(g.start, g.end, g.tokens) for g in merged_c.segments] == [(g.start, g.end, g.tokens) for g in merged.segments]
    if rank == 0:
        s_all = api.Session(model, world * B)
        ref = [parallel.unpack_record(r) for r in records(s_all, 0, world * B)]
        for a, b in zip(got, ref):
            assert a["tokens"] == b["tokens"] and a["steps"] == b["steps"], (a["chunk_index"], a["tokens"], b["tokens"])
            assert a["avg_logprob"] == b["avg_logprob"]
        single = s4.transcribeChunked(audio, topts)
        assert len(single) == len(ordered) >= 3
        for (o1, r1), (o2, r2) in zip(ordered, single):
            assert o1 == o2 and r1.tokens == r2.tokens
            assert [(g.start, g.end, g.tokens) for g in r1.segments] == [(g.start, g.end, g.tokens) for g in r2.segments]
        m2 = api.mergeTranscriptionResults([r for _, r in single])
        assert merged.tokens == m2.tokens and len(merged.segments) == len(m2.segments)
        print("REHEARSAL OK", world, "ranks,", len(got), "records,", len(ordered), "chunks", flush=True)
    comm.barrier()
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
