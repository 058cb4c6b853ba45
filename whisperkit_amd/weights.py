"""Whisper weight container for the HIP library (`WHIPW001` blob).

The reference never sees weights: they live inside CoreML bundles produced by
the external whisperkittools (reference `Sources/WhisperKit/Core/WhisperKit.swift:372-374`).
Our library consumes one flat blob: a small table of named tensors followed by
256-byte aligned payloads.  This module builds that blob from

* a dict of openai/whisper-named fp32 tensors (`encoder.blocks.0.attn.query.weight` ...),
  either synthetic (`synthetic_state_dict`) or converted from a HF
  `WhisperForConditionalGeneration` state dict (`from_hf_state_dict`), and
* the model dimensions (`WhisperDims`).

All layout decisions the kernels depend on are made here, once, at conversion time:

* GEMM weights are stored `[N][K]` fp16 (K contiguous) - the MFMA B operand reads
  8 consecutive K per lane straight from that layout.
* q/k/v projections are fused to one `[3d][d]` matrix; the `head_dim**-0.5`
  query scale (0.125, exact in fp16) is folded into the q rows and q bias.
* conv1/conv2 are re-ordered to `[C_out][tap][C_in]` so that the convolution is a plain
  GEMM over an overlapping-row view of the time-major input.
* cross-attention k/v projections of all decoder layers are fused to `[L*2d][d]`.
* LayerNorm params, biases and positional embeddings stay fp32.
"""
from __future__ import annotations

import dataclasses
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

MAGIC = b"WHIPW001"
DT_F16, DT_F32, DT_I32 = 0, 1, 2
_NP = {DT_F16: np.float16, DT_F32: np.float32, DT_I32: np.int32}
ALIGN = 256


@dataclasses.dataclass(frozen=True)
class WhisperDims:
    """openai/whisper `ModelDimensions` (SURVEY.md section 8 legend)."""

    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int

    @property
    def is_multilingual(self) -> bool:
        # reference Utilities/ModelUtilities.swift:124-126: 51864 == english-only
        return self.n_vocab >= 51865

    def as_tuple(self) -> Tuple[int, ...]:
        return dataclasses.astuple(self)


MODEL_DIMS: Dict[str, WhisperDims] = {
    "tiny.en": WhisperDims(80, 1500, 384, 6, 4, 51864, 448, 384, 6, 4),
    "tiny": WhisperDims(80, 1500, 384, 6, 4, 51865, 448, 384, 6, 4),
    "base": WhisperDims(80, 1500, 512, 8, 6, 51865, 448, 512, 8, 6),
    "small": WhisperDims(80, 1500, 768, 12, 12, 51865, 448, 768, 12, 12),
    "medium": WhisperDims(80, 1500, 1024, 16, 24, 51865, 448, 1024, 16, 24),
    "large-v2": WhisperDims(80, 1500, 1280, 20, 32, 51865, 448, 1280, 20, 32),
    "large-v3": WhisperDims(128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 32),
    # reduced-size configs for fast parity tests (same architecture, smaller numbers)
    "test-micro": WhisperDims(80, 1500, 128, 2, 2, 51864, 448, 128, 2, 2),
    "test-micro-ml": WhisperDims(128, 1500, 128, 2, 2, 51866, 448, 128, 2, 2),
    # headline widths with 2 + 2 layers: the kernel instantiations the benchmark runs (d = 768 / 1280, 12 / 20 heads,
    # 80 / 128 mel, V = 51865 / 51866; shapes pinned by the reference at UnitTests.swift:541-611,721-732) at an oracle cost of seconds
    "test-tiny-en-l2": WhisperDims(80, 1500, 384, 6, 2, 51864, 448, 384, 6, 2),
    "test-small-l2": WhisperDims(80, 1500, 768, 12, 2, 51865, 448, 768, 12, 2),
    "test-large-v3-l2": WhisperDims(128, 1500, 1280, 20, 2, 51866, 448, 1280, 20, 2),
}


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> np.ndarray:
    """Fixed encoder positional embedding (openai/whisper model.py `sinusoids`)."""
    assert channels % 2 == 0
    log_timescale_increment = np.log(max_timescale) / (channels // 2 - 1)
    inv_timescales = np.exp(-log_timescale_increment * np.arange(channels // 2, dtype=np.float64))
    scaled_time = np.arange(length, dtype=np.float64)[:, None] * inv_timescales[None, :]
    return np.concatenate([np.sin(scaled_time), np.cos(scaled_time)], axis=1).astype(np.float32)


def _f16_round(a: np.ndarray) -> np.ndarray:
    return a.astype(np.float16).astype(np.float32)


def _synthetic_specs(dims: WhisperDims, std: float, embed_std: Optional[float]):
    """(name, shape, kind, scale) in generation order.  kind: 'mat' N(0, s) rounded to fp16; 'vec' N(0, s) fp32;
    'gamma' 1 + N(0, 0.1); 'sin' the fixed encoder positional embedding."""
    d, dt = dims.n_audio_state, dims.n_text_state
    specs = []

    def mat(name, *shape, s=std):
        specs.append((name, shape, "mat", s))

    def vec(name, *shape, s=std):
        specs.append((name, shape, "vec", s))

    def ln(prefix, n):
        specs.append((prefix + ".weight", (n,), "gamma", 0.1))
        vec(prefix + ".bias", n)

    def attn(prefix, n):
        mat(prefix + ".query.weight", n, n)
        vec(prefix + ".query.bias", n)
        mat(prefix + ".key.weight", n, n)  # no bias (openai/whisper MultiHeadAttention)
        mat(prefix + ".value.weight", n, n)
        vec(prefix + ".value.bias", n)
        mat(prefix + ".out.weight", n, n)
        vec(prefix + ".out.bias", n)

    def mlp(prefix, n):
        mat(prefix + ".0.weight", 4 * n, n)
        vec(prefix + ".0.bias", 4 * n)
        mat(prefix + ".2.weight", n, 4 * n)
        vec(prefix + ".2.bias", n)

    # conv stem: a larger std keeps the activations O(1) like a trained model
    mat("encoder.conv1.weight", d, dims.n_mels, 3, s=std * 4)
    vec("encoder.conv1.bias", d)
    mat("encoder.conv2.weight", d, d, 3, s=std * 2)
    vec("encoder.conv2.bias", d)
    specs.append(("encoder.positional_embedding", (dims.n_audio_ctx, d), "sin", 0.0))
    for i in range(dims.n_audio_layer):
        p = f"encoder.blocks.{i}"
        ln(p + ".attn_ln", d)
        attn(p + ".attn", d)
        ln(p + ".mlp_ln", d)
        mlp(p + ".mlp", d)
    ln("encoder.ln_post", d)
    mat("decoder.token_embedding.weight", dims.n_vocab, dt, s=embed_std if embed_std else std)
    vec("decoder.positional_embedding", dims.n_text_ctx, dt)
    for i in range(dims.n_text_layer):
        p = f"decoder.blocks.{i}"
        ln(p + ".attn_ln", dt)
        attn(p + ".attn", dt)
        ln(p + ".cross_attn_ln", dt)
        attn(p + ".cross_attn", dt)
        ln(p + ".mlp_ln", dt)
        mlp(p + ".mlp", dt)
    ln("decoder.ln", dt)
    return specs


def _draw(rng, shape, kind, s, dims):
    if kind == "sin":
        return sinusoids(shape[0], shape[1])
    n = int(np.prod(shape))
    x = rng.standard_normal(shape if kind == "mat" else n, dtype=np.float32)
    if kind == "mat":
        x *= np.float32(s)
        return _f16_round(x)
    if kind == "gamma":
        return (1.0 + 0.1 * x).astype(np.float32)
    return (x * s).astype(np.float32).reshape(shape)


def synthetic_state_dict(dims: WhisperDims, seed: int = 0, std: float = 0.02,
                         embed_std: Optional[float] = None, parallel: Optional[bool] = None) -> Dict[str, np.ndarray]:
    """Deterministic random-init weights in openai/whisper naming, fp32 values that are
    exactly representable in fp16 (so the oracle and the HIP path see identical weights).

    N(0, std) matrices, LayerNorm gamma ~ 1 + N(0, 0.1), beta/bias ~ N(0, std): non-trivial
    LN/bias params make the parity tests sensitive to every term (SURVEY.md section 8d
    uses gamma=1/beta=0, which would hide a swapped or missing bias).

    parallel=False draws every tensor from one generator in sequence (the stream the committed golden
    fixtures were made with); parallel=True gives every tensor its own `default_rng([seed, index])` stream
    and draws them on a thread pool (large-v3 = 1.5 G parameters: ~10x faster).  Default: parallel for
    models above 200 M parameters.  Both are deterministic in (dims, seed).
    """
    specs = _synthetic_specs(dims, std, embed_std)
    if parallel is None:
        parallel = sum(int(np.prod(sh)) for _, sh, _, _ in specs) > 200_000_000
    if not parallel:
        rng = np.random.default_rng(seed)
        return {name: _draw(rng, sh, kind, s, dims) for name, sh, kind, s in specs}
    from concurrent.futures import ThreadPoolExecutor
    import os

    def job(i):
        name, sh, kind, s = specs[i]
        return _draw(np.random.default_rng([seed, i]), sh, kind, s, dims)
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        arrs = list(ex.map(job, range(len(specs))))
    return {specs[i][0]: arrs[i] for i in range(len(specs))}


_HF_MAP = [
    ("model.encoder.", "encoder."), ("model.decoder.", "decoder."),
    (".layers.", ".blocks."), (".self_attn_layer_norm", ".attn_ln"),
    (".encoder_attn_layer_norm", ".cross_attn_ln"), (".final_layer_norm", ".mlp_ln"),
    (".self_attn.", ".attn."), (".encoder_attn.", ".cross_attn."),
    (".q_proj", ".query"), (".k_proj", ".key"), (".v_proj", ".value"), (".out_proj", ".out"),
    (".fc1", ".mlp.0"), (".fc2", ".mlp.2"),
    ("encoder.layer_norm", "encoder.ln_post"), ("decoder.layer_norm", "decoder.ln"),
    ("decoder.embed_tokens", "decoder.token_embedding"),
    ("encoder.embed_positions.weight", "encoder.positional_embedding"),
    ("decoder.embed_positions.weight", "decoder.positional_embedding"),
]


def from_hf_state_dict(hf_sd) -> Dict[str, np.ndarray]:
    """Rename a HF `WhisperForConditionalGeneration` state dict to openai/whisper names."""
    out = {}
    for k, v in hf_sd.items():
        if k.startswith("proj_out"):
            continue  # tied to the token embedding
        n = k
        for a, b in _HF_MAP:
            n = n.replace(a, b)
        out[n] = np.asarray(v.detach().cpu().float().numpy() if hasattr(v, "detach") else v, dtype=np.float32)
    return out


def to_hf_state_dict(sd: Dict[str, np.ndarray]):
    """Inverse of `from_hf_state_dict` (used only to cross-check the oracle against HF)."""
    import torch

    inv = [(b, a) for a, b in reversed(_HF_MAP)]
    out = {}
    for k, v in sd.items():
        n = k
        # apply the specific renames in an order that undoes the forward map
        n = n.replace("encoder.positional_embedding", "encoder.embed_positions.weight")
        n = n.replace("decoder.positional_embedding", "decoder.embed_positions.weight")
        n = n.replace("decoder.token_embedding", "decoder.embed_tokens")
        n = n.replace("encoder.ln_post", "encoder.layer_norm")
        if n.startswith("decoder.ln."):
            n = n.replace("decoder.ln.", "decoder.layer_norm.")
        n = n.replace(".mlp.0", ".fc1").replace(".mlp.2", ".fc2")
        n = n.replace(".query", ".q_proj").replace(".key", ".k_proj").replace(".value", ".v_proj")
        n = n.replace(".out.", ".out_proj.")
        n = n.replace(".cross_attn_ln", ".encoder_attn_layer_norm").replace(".attn_ln", ".self_attn_layer_norm")
        n = n.replace(".mlp_ln", ".final_layer_norm")
        n = n.replace(".cross_attn.", ".encoder_attn.").replace(".attn.", ".self_attn.")
        n = n.replace(".blocks.", ".layers.")
        n = "model." + n
        out[n] = torch.from_numpy(np.ascontiguousarray(v))
    out["proj_out.weight"] = out["model.decoder.embed_tokens.weight"]
    return out


def default_alignment_heads(dims: WhisperDims) -> List[Tuple[int, int]]:
    """openai/whisper default: every head of the upper half of the decoder layers
    (model.py: `all_heads[self.dims.n_text_layer // 2:] = True`)."""
    return [(l, h) for l in range(dims.n_text_layer // 2, dims.n_text_layer) for h in range(dims.n_text_head)]


def kernel_tensors(dims: WhisperDims, sd: Dict[str, np.ndarray]) -> List[Tuple[str, np.ndarray]]:
    """Derive the tensors the HIP kernels read, in their final layouts."""
    d, dt = dims.n_audio_state, dims.n_text_state
    hd = d // dims.n_audio_head
    assert hd == 64 and dt // dims.n_text_head == 64, "Whisper head_dim is 64 for every size"
    scale = np.float32(hd ** -0.5)  # 0.125: exact in fp16
    jobs: List[Tuple[str, object, object]] = []   # (name, dtype, thunk or array): evaluated on a thread pool below

    def f16(name, a):
        jobs.append((name, np.float16, a))

    def f32(name, a):
        jobs.append((name, np.float32, a))

    def cat16(parts, scales=None):
        # concatenate along axis 0 straight into an fp16 buffer (no fp32 temporary of the fused matrix)
        def run():
            o = np.empty((sum(p.shape[0] for p in parts),) + parts[0].shape[1:], np.float16)
            r = 0
            for k, p_ in enumerate(parts):
                np.multiply(p_, scales[k] if scales else np.float32(1), out=o[r:r + p_.shape[0]], casting="unsafe")
                r += p_.shape[0]
            return o
        return run

    def qkv(prefix, src, n):
        f16(prefix + ".qkv.w", cat16([sd[src + ".query.weight"], sd[src + ".key.weight"], sd[src + ".value.weight"]],
                                     [scale, np.float32(1), np.float32(1)]))
        b = np.concatenate([sd[src + ".query.bias"] * scale, np.zeros(n, np.float32), sd[src + ".value.bias"]], 0)
        f32(prefix + ".qkv.b", b)

    # conv taps: W[co][ci][kk] -> [co][kk][ci]
    f16("enc.conv1.w", sd["encoder.conv1.weight"].transpose(0, 2, 1).reshape(d, -1))
    f32("enc.conv1.b", sd["encoder.conv1.bias"])
    f16("enc.conv2.w", sd["encoder.conv2.weight"].transpose(0, 2, 1).reshape(d, -1))
    f32("enc.conv2.b", sd["encoder.conv2.bias"])
    f32("enc.pos", sd["encoder.positional_embedding"])
    for i in range(dims.n_audio_layer):
        s, p = f"encoder.blocks.{i}", f"enc.{i}"
        f32(p + ".ln1.g", sd[s + ".attn_ln.weight"]); f32(p + ".ln1.b", sd[s + ".attn_ln.bias"])
        qkv(p, s + ".attn", d)
        f16(p + ".o.w", sd[s + ".attn.out.weight"]); f32(p + ".o.b", sd[s + ".attn.out.bias"])
        f32(p + ".ln2.g", sd[s + ".mlp_ln.weight"]); f32(p + ".ln2.b", sd[s + ".mlp_ln.bias"])
        f16(p + ".fc1.w", sd[s + ".mlp.0.weight"]); f32(p + ".fc1.b", sd[s + ".mlp.0.bias"])
        f16(p + ".fc2.w", sd[s + ".mlp.2.weight"]); f32(p + ".fc2.b", sd[s + ".mlp.2.bias"])
    f32("enc.lnp.g", sd["encoder.ln_post.weight"]); f32("enc.lnp.b", sd["encoder.ln_post.bias"])

    f16("dec.emb", sd["decoder.token_embedding.weight"])
    f32("dec.pos", sd["decoder.positional_embedding"])
    ckv_w, ckv_b = [], []
    for i in range(dims.n_text_layer):
        s, p = f"decoder.blocks.{i}", f"dec.{i}"
        f32(p + ".ln1.g", sd[s + ".attn_ln.weight"]); f32(p + ".ln1.b", sd[s + ".attn_ln.bias"])
        qkv(p, s + ".attn", dt)
        f16(p + ".o.w", sd[s + ".attn.out.weight"]); f32(p + ".o.b", sd[s + ".attn.out.bias"])
        f32(p + ".ln2.g", sd[s + ".cross_attn_ln.weight"]); f32(p + ".ln2.b", sd[s + ".cross_attn_ln.bias"])
        f16(p + ".cq.w", sd[s + ".cross_attn.query.weight"] * scale)
        f32(p + ".cq.b", sd[s + ".cross_attn.query.bias"] * scale)
        ckv_w += [sd[s + ".cross_attn.key.weight"], sd[s + ".cross_attn.value.weight"]]
        ckv_b += [np.zeros(dt, np.float32), sd[s + ".cross_attn.value.bias"]]
        f16(p + ".co.w", sd[s + ".cross_attn.out.weight"]); f32(p + ".co.b", sd[s + ".cross_attn.out.bias"])
        f32(p + ".ln3.g", sd[s + ".mlp_ln.weight"]); f32(p + ".ln3.b", sd[s + ".mlp_ln.bias"])
        f16(p + ".fc1.w", sd[s + ".mlp.0.weight"]); f32(p + ".fc1.b", sd[s + ".mlp.0.bias"])
        f16(p + ".fc2.w", sd[s + ".mlp.2.weight"]); f32(p + ".fc2.b", sd[s + ".mlp.2.bias"])
    f16("dec.ckv.w", cat16(ckv_w))   # [L*2d][d]: (K_0, V_0, K_1, V_1, ...)
    f32("dec.ckv.b", np.concatenate(ckv_b, 0))
    f32("dec.ln.g", sd["decoder.ln.weight"]); f32("dec.ln.b", sd["decoder.ln.bias"])

    def realise(job):
        name, dt, a = job
        a = a() if callable(a) else a
        return name, np.ascontiguousarray(a if a.dtype == dt else np.asarray(a, dtype=np.float32).astype(dt))
    if sum(int(np.prod(v.shape)) for v in sd.values()) < 50_000_000:
        return [realise(j_) for j_ in jobs]
    from concurrent.futures import ThreadPoolExecutor
    import os
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        return list(ex.map(realise, jobs))


def pack_blob(dims: WhisperDims, sd: Dict[str, np.ndarray], alignment_heads: Optional[Iterable[Tuple[int, int]]] = None) -> np.ndarray:
    """Serialise to the `WHIPW001` container read by `wh_model_create` (include/whisperhip.h).
    Returns a uint8 array (buffer protocol; `bytes(blob)` / `blob.tofile(path)` for a file).
    `alignment_heads` ((layer, head) pairs, e.g. HF generation_config.alignment_heads) travel inside the blob as the optional
    int32 tensor `dec.alignment_heads` [n][2], so that a file-based `wh_model_load` selects the same word-timestamp heads."""
    tensors = kernel_tensors(dims, sd)
    if alignment_heads:
        ah = np.ascontiguousarray(np.array([(int(l), int(h)) for l, h in alignment_heads], dtype=np.int32).reshape(-1, 2))
        tensors.append(("dec.alignment_heads", ah))
    n = len(tensors)
    entry = struct.Struct("<64sii4qqq")  # name, dtype, ndim, shape[4], offset, nbytes
    header_size = len(MAGIC) + 4 * 10 + 4 + 4 + entry.size * n
    off = (header_size + ALIGN - 1) // ALIGN * ALIGN
    table, payload_offsets = [], []
    for name, a in tensors:
        dt = DT_F16 if a.dtype == np.float16 else DT_F32 if a.dtype == np.float32 else DT_I32
        shape = list(a.shape) + [1] * (4 - a.ndim)
        table.append(entry.pack(name.encode(), dt, a.ndim, *shape, off, a.nbytes))
        payload_offsets.append(off)
        off = (off + a.nbytes + ALIGN - 1) // ALIGN * ALIGN
    buf = np.zeros(off, dtype=np.uint8)
    head = MAGIC + struct.pack("<10i", *dims.as_tuple()) + struct.pack("<ii", n, 0) + b"".join(table)
    buf[: len(head)] = np.frombuffer(head, dtype=np.uint8)
    def put(k):
        a, o = tensors[k][1], payload_offsets[k]
        buf[o: o + a.nbytes] = a.reshape(-1).view(np.uint8)
    if off < 200_000_000:
        for k in range(n):
            put(k)
    else:
        from concurrent.futures import ThreadPoolExecutor
        import os
        with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
            list(ex.map(put, range(n)))
    return buf


def unpack_blob(blob) -> Tuple[WhisperDims, Dict[str, np.ndarray]]:
    """Parse a blob back (host-side sanity checks and tests)."""
    blob = bytes(blob) if not isinstance(blob, (bytes, bytearray)) else blob
    assert blob[:8] == MAGIC
    dims = WhisperDims(*struct.unpack_from("<10i", blob, 8))
    n, _ = struct.unpack_from("<ii", blob, 48)
    entry = struct.Struct("<64sii4qqq")
    out = {}
    for i in range(n):
        name, dt, ndim, s0, s1, s2, s3, off, nb = entry.unpack_from(blob, 56 + i * entry.size)
        shape = (s0, s1, s2, s3)[:ndim]
        out[name.rstrip(b"\0").decode()] = np.frombuffer(blob, dtype=_NP[dt], count=int(np.prod(shape)), offset=off).reshape(shape)
    return dims, out
