"""Chunk-level data parallelism: one process per GPU, independent 30 s chunks block-partitioned over ranks,
one fixed-size all-gather of the per-chunk result records at the end.

The reference's only parallelism is a TaskGroup over independent audio arrays that share the model objects
(Core/WhisperKit.swift:735-812) followed by an in-process merge (Utilities/TranscriptionUtilities.swift:76-157);
there is no collective in it.  Here the merge step becomes one `all_gather` (RCCL over xGMI on GPUs, gloo in the CPU
tests); ~1 KB per chunk, so it is latency-bound and never bandwidth-bound.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

RECORD_TOKENS = 232
RECORD_INTS = RECORD_TOKENS + 8   # tokens[232], n_tokens, chunk_index, seek, steps, 4 x float bits


def partition_chunks(n_chunks: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition that keeps output order: rank r owns [start, end)."""
    base, rem = divmod(n_chunks, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack_record(chunk_index: int, tokens: Sequence[int], seek: int, steps: int, avg_logprob: float, temperature: float,
                compression_ratio: float, no_speech_prob: float = 0.0) -> np.ndarray:
    r = np.zeros(RECORD_INTS, dtype=np.int32)
    n = min(len(tokens), RECORD_TOKENS)
    r[:n] = np.asarray(tokens[:n], dtype=np.int32)
    r[RECORD_TOKENS:RECORD_TOKENS + 4] = (n, chunk_index, seek, steps)
    r[RECORD_TOKENS + 4:] = np.array([avg_logprob, temperature, compression_ratio, no_speech_prob], dtype=np.float32).view(np.int32)
    return r


def unpack_record(r: np.ndarray) -> dict:
    r = np.asarray(r, dtype=np.int32)
    n, idx, seek, steps = (int(v) for v in r[RECORD_TOKENS:RECORD_TOKENS + 4])
    f = r[RECORD_TOKENS + 4:].view(np.float32)
    return dict(chunk_index=idx, tokens=[int(t) for t in r[:n]], seek=seek, steps=steps, avg_logprob=float(f[0]),
                temperature=float(f[1]), compression_ratio=float(f[2]), no_speech_prob=float(f[3]))


def gather_records(local_records: np.ndarray, max_per_rank: int, device=None, group=None) -> List[dict]:
    """All ranks contribute `local_records` [k, RECORD_INTS] (k <= max_per_rank, padded with chunk_index -1);
    every rank receives all valid records sorted by chunk index."""
    import torch
    import torch.distributed as dist

    buf = np.full((max_per_rank, RECORD_INTS), 0, dtype=np.int32)
    buf[:, RECORD_TOKENS + 1] = -1
    buf[: len(local_records)] = local_records
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        out = torch.empty((dist.get_world_size(group) * max_per_rank, RECORD_INTS), dtype=torch.int32, device=t.device)
        dist.all_gather_into_tensor(out, t, group=group)
    else:
        out = t
    recs = [unpack_record(r) for r in out.cpu().numpy()]
    return sorted((r for r in recs if r["chunk_index"] >= 0), key=lambda r: r["chunk_index"])
