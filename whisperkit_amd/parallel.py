"""Chunk-level data parallelism: one process per GPU, independent 30 s chunks block-partitioned over ranks,
one fixed-size all-gather of the per-chunk result records at the end.

The reference's only parallelism is a TaskGroup over independent audio arrays that share the model objects
(Core/WhisperKit.swift:735-812) followed by an in-process merge (Utilities/TranscriptionUtilities.swift:76-157);
there is no collective in it.  Here the merge step becomes one `all_gather` (RCCL over xGMI on GPUs, gloo in the CPU
tests); ~1 KB per chunk, so it is latency-bound and never bandwidth-bound.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

RECORD_TOKENS = 232
RECORD_INTS = RECORD_TOKENS + 8   # tokens[232], n_tokens, chunk_index, seek, steps, 4 x float bits


def partition_chunks(n_chunks: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition that keeps output order: rank r owns [start, end).  (wh_partition_chunks is the same rule behind
    the C ABI; tests/test_comm_abi.py checks that the two agree.)"""
    base, rem = divmod(n_chunks, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class Comm:
    """wh_comm behind the C ABI (include/whisperhip.h "multi-GPU"): the partition / gather / merge step of the path as a Swift or C
    host would drive it - RCCL all-gather over xGMI between one-process-per-GPU ranks (`transport="rccl"`), or the library's TCP
    star for CPU-only hosts and several ranks on one GPU (`transport="tcp"`).  `exchange_id(id_bytes) -> id_bytes` hands rank 0's
    128-byte id to the other ranks (any out-of-band channel: here a torch.distributed broadcast or the environment)."""

    def __init__(self, world_size: int, rank: int, transport: str = "rccl", device: Optional[int] = None, tcp_address: str = "127.0.0.1:29533",
                 exchange_id=None):
        import ctypes as C

        from . import _lib as L
        self._C, self._L, self.lib = C, L, L.load()
        self.transport = L.COMM_RCCL if transport == "rccl" else L.COMM_TCP
        if self.transport == L.COMM_RCCL and device is None:
            raise ValueError("an RCCL communicator needs the rank's device (its LOCAL_RANK): every rank defaulting to GPU 0 is refused by RCCL")
        device = 0 if device is None else int(device)
        ident = (C.c_uint8 * L.COMM_ID_BYTES)()
        if world_size > 1:
            if rank == 0 or self.transport == L.COMM_TCP:
                self._check(self.lib.wh_comm_unique_id(self.transport, tcp_address.encode(), ident))
            if self.transport == L.COMM_RCCL:
                if exchange_id is None:
                    raise ValueError("an RCCL communicator of more than one rank needs exchange_id (rank 0's ncclUniqueId)")
                raw = exchange_id(bytes(ident))
                ident = (C.c_uint8 * L.COMM_ID_BYTES).from_buffer_copy(raw)
        self.handle = C.c_void_p()
        self._check(self.lib.wh_comm_create(self.transport, ident if world_size > 1 else None, world_size, rank, device, C.byref(self.handle)))
        self.world_size, self.rank = world_size, rank

    def _check(self, code):
        from .api import _check
        _check(code)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.wh_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def barrier(self):
        self._check(self.lib.wh_comm_barrier(self.handle))

    def partition(self, n_chunks: int) -> Tuple[int, int]:
        C = self._C
        s, e = C.c_int(), C.c_int()
        self._check(self.lib.wh_partition_chunks(n_chunks, self.world_size, self.rank, C.byref(s), C.byref(e)))
        return s.value, e.value

    def gather_records(self, local_records: np.ndarray, max_per_rank: int) -> List[dict]:
        """`local_records` [k, RECORD_INTS] int32 (pack_record rows = wh_chunk_record) -> every rank's records by chunk index."""
        C, L = self._C, self._L
        loc = np.ascontiguousarray(local_records, dtype=np.int32).reshape(-1, RECORD_INTS)
        out = np.zeros((self.world_size * max_per_rank, RECORD_INTS), dtype=np.int32)
        n = C.c_int()
        self._check(self.lib.wh_comm_gather_records(self.handle, loc.ctypes.data_as(C.POINTER(L.WhChunkRecord)), len(loc), max_per_rank,
                                                    out.ctypes.data_as(C.POINTER(L.WhChunkRecord)), len(out), C.byref(n)))
        return [unpack_record(r) for r in out[: n.value]]

    def gather_results(self, local: Sequence[Tuple[int, object]], capacity: int = 4096) -> List[Tuple[int, object]]:
        """[(chunk_index, TranscriptionResult)] of this rank -> every rank's, by chunk index (wh_comm_gather_transcriptions)."""
        C, L = self._C, self._L
        from .api import _collect
        hs = (C.c_void_p * max(len(local), 1))(*[r._handle for _, r in local])
        idx = np.ascontiguousarray([i for i, _ in local], dtype=np.int32)
        outs = (C.c_void_p * capacity)()
        oidx = np.zeros(capacity, dtype=np.int32)
        n = C.c_int()
        self._check(self.lib.wh_comm_gather_transcriptions(self.handle, hs, idx.ctypes.data_as(L.PI32), len(local), outs,
                                                           oidx.ctypes.data_as(L.PI32), capacity, C.byref(n)))
        return [(int(oidx[i]), _collect(C.c_void_p(outs[i]))) for i in range(n.value)]


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


def gather_records(local_records: np.ndarray, max_per_rank: int, device=None, group=None, comm: "Comm" = None) -> List[dict]:
    """All ranks contribute `local_records` [k, RECORD_INTS] (k <= max_per_rank, padded with chunk_index -1);
    every rank receives all valid records sorted by chunk index.  With `comm` the gather runs behind the C ABI (wh_comm_*)."""
    if comm is not None:
        return comm.gather_records(local_records, max_per_rank)
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


# ------------------------------------------------------------------------------------------------ long audio across GPUs
def gather_results(local: Sequence[Tuple[int, object]], device=None, group=None, comm: "Comm" = None) -> List[Tuple[int, object]]:
    """All-gather of whole TranscriptionResults: `local` = [(chunk_index, TranscriptionResult)] of this rank; every rank receives
    every (chunk_index, result) sorted by chunk index.  The wire format is the reference's own Codable JSON document
    (`TranscriptionResult.toJSON`), so two collectives of a few KB: the payload lengths, then the padded payloads."""
    if comm is not None:
        return comm.gather_results(local)
    import json

    import torch
    import torch.distributed as dist

    from .api import TranscriptionResult
    payload = json.dumps([[int(i), r.toJSON()] for i, r in local]).encode("utf-8")
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return sorted(((i, r) for i, r in local), key=lambda x: x[0])
    world = dist.get_world_size(group)
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, n, group=group)
    cap = int(sizes.max().item())
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(buf.device)
    out = torch.empty(world * cap, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, buf, group=group)
    out = out.cpu().numpy()
    results = []
    for r in range(world):
        doc = bytes(out[r * cap:r * cap + int(sizes[r].item())]).decode("utf-8")
        results += [(int(i), TranscriptionResult.fromJSON(j)) for i, j in json.loads(doc)]
    return sorted(results, key=lambda x: x[0])


def transcribe_chunked_sharded(session, audio: np.ndarray, options=None, device=None, group=None, comm: "Comm" = None):
    """WhisperKit.transcribe(audioArray:) with `.vad` chunking (Core/WhisperKit.swift:867-931) over the GPUs of a node: every rank
    cuts the audio at the same places (VADAudioChunker, host code), transcribes its contiguous block of chunks as one device batch
    per `session.B` (clipTimestamps reset, :889-891), shifts the results by the chunk offsets (updateSeekOffsetsForResults), and one
    gather hands every rank all chunk results in order.  Returns ([(seekOffsetSamples, TranscriptionResult)], merged result)
    where merged = mergeTranscriptionResults over the chunks (Utilities/TranscriptionUtilities.swift:76-157)."""
    import dataclasses

    import torch.distributed as dist

    from . import api
    options = options or api.DecodingOptions()
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    chunks = api.vadChunkAll(audio, options=options) if len(audio) > 480000 else [(0, len(audio))]
    if comm is not None:
        world, rank = comm.world_size, comm.rank
    else:
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank(group) if world > 1 else 0
    s, e = partition_chunks(len(chunks), world, rank)
    chunk_options = dataclasses.replace(options, clipTimestamps=()) if len(chunks) > 1 else options
    local = []
    for b0 in range(s, e, session.B):
        block = chunks[b0:min(b0 + session.B, e)]
        res = session.transcribe([audio[c0:c1] for c0, c1 in block], chunk_options)
        local += [(b0 + k, r.withSeekOffset(block[k][0]) if len(chunks) > 1 else r) for k, r in enumerate(res)]
    everything = gather_results(local, device=device, group=group, comm=comm)
    ordered = [(chunks[i][0], r) for i, r in everything]
    return ordered, api.mergeTranscriptionResults([r for _, r in ordered])
