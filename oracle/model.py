"""Whisper encoder / decoder-step oracle (torch CPU fp32) - test infrastructure only.

Restates openai/whisper `model.py` (AudioEncoder, TextDecoder, ResidualAttentionBlock,
MultiHeadAttention) = transformers 5.15.0 `modeling_whisper.py:267-309,371-377,431-446,
566-573,622-624,676-682,965-970`.  The reference executes this math inside CoreML bundles
(Sources/WhisperKit/Core/AudioEncoder.swift:60, Core/TextDecoder.swift:406), one decoder
call per token with an explicit KV cache (Core/TextDecoder.swift:573-717); `DecoderState.step`
mirrors that call pattern: (token, cache_length) -> logits, and appends this step's K/V.

Weights come in as a dict of openai/whisper-named float32 numpy arrays
(`whisperkit_amd.weights.synthetic_state_dict` or a converted checkpoint).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _t(a) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


class OracleWhisper:
    def __init__(self, dims, sd: Dict[str, np.ndarray], alignment_heads: Optional[Sequence[Tuple[int, int]]] = None):
        self.dims = dims
        self.w = {k: _t(v) for k, v in sd.items()}
        self.alignment_heads = list(alignment_heads) if alignment_heads is not None else [
            (l, h) for l in range(dims.n_text_layer // 2, dims.n_text_layer) for h in range(dims.n_text_head)]

    # ---------------------------------------------------------------- encoder
    def _mha(self, p: str, x: torch.Tensor, xa: torch.Tensor, n_head: int, mask=None, return_qk=False):
        w = self.w
        q = F.linear(x, w[p + ".query.weight"], w[p + ".query.bias"])
        k = F.linear(xa, w[p + ".key.weight"])
        v = F.linear(xa, w[p + ".value.weight"], w[p + ".value.bias"])
        return self._attend(p, q, k, v, n_head, mask, return_qk)

    def _attend(self, p, q, k, v, n_head, mask=None, return_qk=False):
        w = self.w
        tq, d = q.shape
        hd = d // n_head
        qh = q.view(tq, n_head, hd).permute(1, 0, 2) * hd ** -0.5
        kh = k.view(-1, n_head, hd).permute(1, 2, 0)
        vh = v.view(-1, n_head, hd).permute(1, 0, 2)
        qk = qh @ kh                                   # [H, tq, tk]
        if mask is not None:
            qk = qk + mask
        pr = torch.softmax(qk, dim=-1)
        o = (pr @ vh).permute(1, 0, 2).reshape(tq, d)
        o = F.linear(o, w[p + ".out.weight"], w[p + ".out.bias"])
        return (o, pr) if return_qk else o

    def encode(self, mel: np.ndarray) -> np.ndarray:
        """mel [n_mels, 3000] -> encoder output [1500, d] float32."""
        w, dims = self.w, self.dims
        with torch.no_grad():
            x = _t(mel)[None]
            x = F.gelu(F.conv1d(x, w["encoder.conv1.weight"], w["encoder.conv1.bias"], padding=1))
            x = F.gelu(F.conv1d(x, w["encoder.conv2.weight"], w["encoder.conv2.bias"], stride=2, padding=1))
            x = x[0].T + w["encoder.positional_embedding"]
            for i in range(dims.n_audio_layer):
                p = f"encoder.blocks.{i}"
                xn = F.layer_norm(x, (x.shape[-1],), w[p + ".attn_ln.weight"], w[p + ".attn_ln.bias"])
                x = x + self._mha(p + ".attn", xn, xn, dims.n_audio_head)
                xn = F.layer_norm(x, (x.shape[-1],), w[p + ".mlp_ln.weight"], w[p + ".mlp_ln.bias"])
                h = F.gelu(F.linear(xn, w[p + ".mlp.0.weight"], w[p + ".mlp.0.bias"]))
                x = x + F.linear(h, w[p + ".mlp.2.weight"], w[p + ".mlp.2.bias"])
            x = F.layer_norm(x, (x.shape[-1],), w["encoder.ln_post.weight"], w["encoder.ln_post.bias"])
        return x.numpy()

    # ---------------------------------------------------------------- decoder
    def new_state(self, enc: np.ndarray, kvFloat16: bool = False, crossFloat16: Optional[bool] = None) -> "DecoderState":
        return DecoderState(self, enc, kvFloat16, crossFloat16)


class DecoderState:
    """Per-window decode state = the reference's `DecodingInputs` (Core/Models.swift:291-323):
    self-attention key/value caches of 224 positions + the alignment matrix [224, 1500]."""

    MAX_CTX = 224

    def __init__(self, model: OracleWhisper, enc: np.ndarray, kvFloat16: bool = False, crossFloat16: Optional[bool] = None):
        """kvFloat16: keys and values (cross and self) are rounded to Float16 when they are stored - the storage type of the
        reference's caches (FloatType key/value MLMultiArrays, Core/Models.swift:291-323) and of the device's.  Default off: the
        fp32 model the logits tolerance is quoted against; on for fixtures whose sharpened attention amplifies that rounding.
        crossFloat16 (default: = kvFloat16): the same choice for the cross-attention keys / values alone - False models the device's
        weight-absorbed cross-attention, which never materialises (or rounds) them, beside a Float16 self-attention cache."""
        self.m = model
        self.kv16 = kvFloat16
        self.cross16 = kvFloat16 if crossFloat16 is None else crossFloat16
        w, dims = model.w, model.dims
        xa = _t(enc)
        rnd = (lambda t: t.half().float()) if self.cross16 else (lambda t: t)
        with torch.no_grad():
            self.cross_k = [rnd(F.linear(xa, w[f"decoder.blocks.{i}.cross_attn.key.weight"])) for i in range(dims.n_text_layer)]
            self.cross_v = [rnd(F.linear(xa, w[f"decoder.blocks.{i}.cross_attn.value.weight"],
                                         w[f"decoder.blocks.{i}.cross_attn.value.bias"])) for i in range(dims.n_text_layer)]
        self.reset()

    def fresh_like(self) -> "DecoderState":
        """A reset decode state of the SAME window: shares the (read-only) cross keys / values instead of recomputing them - what
        makes beam search and the temperature ladder affordable at 32 layers (every pass / beam needs its own self-attention cache)."""
        s = object.__new__(DecoderState)
        s.m, s.kv16, s.cross16, s.cross_k, s.cross_v = self.m, self.kv16, self.cross16, self.cross_k, self.cross_v
        s.reset()
        return s

    def _store(self, t):
        return t.half().float() if self.kv16 else t

    def reset(self):
        dims = self.m.dims
        d = dims.n_text_state
        self.k = [torch.zeros(self.MAX_CTX, d) for _ in range(dims.n_text_layer)]
        self.v = [torch.zeros(self.MAX_CTX, d) for _ in range(dims.n_text_layer)]
        self.alignment = np.zeros((self.MAX_CTX, dims.n_audio_ctx), dtype=np.float32)
        self.alignment_heads = np.zeros((self.MAX_CTX, max(len(self.m.alignment_heads), 1), dims.n_audio_ctx), dtype=np.float32)
        self.alignment_written = np.zeros(self.MAX_CTX, dtype=bool)

    def copy_from(self, other: "DecoderState"):
        """Key / value cache of another state of the same audio (beam search: openai/whisper's rearrange_kv_cache)."""
        for i in range(len(self.k)):
            self.k[i].copy_(other.k[i])
            self.v[i].copy_(other.v[i])

    def step(self, token: int, pos: int, want_alignment: bool = True):
        """One decoder call (TextDecoder.predictLogits + updateKVCache + updateAlignmentWeights,
        Core/TextDecoder.swift:381-418,218-296): returns logits [V]; writes K/V at `pos` and the
        alignment row at `pos + 1` (reference writes row tokenIndex+1)."""
        m, w, dims = self.m, self.m.w, self.m.dims
        nh = dims.n_text_head
        with torch.no_grad():
            x = (w["decoder.token_embedding.weight"][token] + w["decoder.positional_embedding"][pos])[None]
            align_rows = []
            for i in range(dims.n_text_layer):
                p = f"decoder.blocks.{i}"
                xn = F.layer_norm(x, (x.shape[-1],), w[p + ".attn_ln.weight"], w[p + ".attn_ln.bias"])
                q = F.linear(xn, w[p + ".attn.query.weight"], w[p + ".attn.query.bias"])
                self.k[i][pos] = self._store(F.linear(xn, w[p + ".attn.key.weight"])[0])
                self.v[i][pos] = self._store(F.linear(xn, w[p + ".attn.value.weight"], w[p + ".attn.value.bias"])[0])
                x = x + m._attend(p + ".attn", q, self.k[i][: pos + 1], self.v[i][: pos + 1], nh)
                xn = F.layer_norm(x, (x.shape[-1],), w[p + ".cross_attn_ln.weight"], w[p + ".cross_attn_ln.bias"])
                q = F.linear(xn, w[p + ".cross_attn.query.weight"], w[p + ".cross_attn.query.bias"])
                o, pr = m._attend(p + ".cross_attn", q, self.cross_k[i], self.cross_v[i], nh, return_qk=True)
                x = x + o
                for (l, h) in m.alignment_heads:
                    if l == i:
                        align_rows.append(pr[h, 0])
                xn = F.layer_norm(x, (x.shape[-1],), w[p + ".mlp_ln.weight"], w[p + ".mlp_ln.bias"])
                hdn = F.gelu(F.linear(xn, w[p + ".mlp.0.weight"], w[p + ".mlp.0.bias"]))
                x = x + F.linear(hdn, w[p + ".mlp.2.weight"], w[p + ".mlp.2.bias"])
            x = F.layer_norm(x, (x.shape[-1],), w["decoder.ln.weight"], w["decoder.ln.bias"])
            logits = F.linear(x, w["decoder.token_embedding.weight"])[0]
            if want_alignment and align_rows and pos + 1 < self.MAX_CTX:
                self.alignment[pos + 1] = torch.stack(align_rows).mean(0).numpy()
                self.alignment_heads[pos + 1] = torch.stack(align_rows).numpy()
                self.alignment_written[pos + 1] = True
        return logits.numpy()

    def forward_full(self, tokens: Sequence[int], logits_at: Optional[Sequence[int]] = None, want_alignment: bool = True):
        """Teacher-forced pass over positions 0 .. len(tokens) - 1 in ONE causal forward (openai/whisper TextDecoder.forward with a
        causal mask, model.py / modeling_whisper.py:431-446): the same function of (tokens, encoder output) as len(tokens) calls of
        `step`, at the cost of a few of them - what makes full-depth checks (32 layers, 223 positions) affordable on the CPU.
        Fills the K/V caches and the alignment rows exactly like the stepped calls (a following `step(t, len(tokens))` continues the
        sequence) and returns {position: logits [V]} for the positions in `logits_at` (default: all).  Pinned against `step` by
        tests/test_oracle_golden.py (same values to fp32 round-off)."""
        m, w, dims = self.m, self.m.w, self.m.dims
        nh = dims.n_text_head
        n = len(tokens)
        assert 0 < n <= self.MAX_CTX
        ids = torch.tensor([int(t) for t in tokens], dtype=torch.long)
        with torch.no_grad():
            x = w["decoder.token_embedding.weight"][ids] + w["decoder.positional_embedding"][:n]
            mask = torch.full((n, n), float("-inf")).triu_(1)
            align_rows = []
            for i in range(dims.n_text_layer):
                p = f"decoder.blocks.{i}"
                xn = F.layer_norm(x, (x.shape[-1],), w[p + ".attn_ln.weight"], w[p + ".attn_ln.bias"])
                q = F.linear(xn, w[p + ".attn.query.weight"], w[p + ".attn.query.bias"])
                self.k[i][:n] = self._store(F.linear(xn, w[p + ".attn.key.weight"]))
                self.v[i][:n] = self._store(F.linear(xn, w[p + ".attn.value.weight"], w[p + ".attn.value.bias"]))
                x = x + m._attend(p + ".attn", q, self.k[i][:n], self.v[i][:n], nh, mask=mask)
                xn = F.layer_norm(x, (x.shape[-1],), w[p + ".cross_attn_ln.weight"], w[p + ".cross_attn_ln.bias"])
                q = F.linear(xn, w[p + ".cross_attn.query.weight"], w[p + ".cross_attn.query.bias"])
                o, pr = m._attend(p + ".cross_attn", q, self.cross_k[i], self.cross_v[i], nh, return_qk=True)
                x = x + o
                for (l, h) in m.alignment_heads:
                    if l == i:
                        align_rows.append(pr[h])                       # [n, 1500]
                xn = F.layer_norm(x, (x.shape[-1],), w[p + ".mlp_ln.weight"], w[p + ".mlp_ln.bias"])
                hdn = F.gelu(F.linear(xn, w[p + ".mlp.0.weight"], w[p + ".mlp.0.bias"]))
                x = x + F.linear(hdn, w[p + ".mlp.2.weight"], w[p + ".mlp.2.bias"])
            x = F.layer_norm(x, (x.shape[-1],), w["decoder.ln.weight"], w["decoder.ln.bias"])
            pos = list(range(n)) if logits_at is None else [int(p_) for p_ in logits_at]
            lg = F.linear(x[pos], w["decoder.token_embedding.weight"]).numpy()
            if want_alignment and align_rows:
                rows = torch.stack(align_rows)                         # [heads, n, 1500]
                last = min(n, self.MAX_CTX - 1)                        # row pos + 1 for pos = 0 .. last - 1
                self.alignment[1:last + 1] = rows.mean(0)[:last].numpy()
                self.alignment_heads[1:last + 1] = rows.permute(1, 0, 2)[:last].numpy()
                self.alignment_written[1:last + 1] = True
        return {p_: lg[i] for i, p_ in enumerate(pos)}

    def postprocessed_alignment(self, z_normalize: bool = True, median_filter_width: int = 7) -> np.ndarray:
        """openai/whisper timing.py find_alignment / transformers generation_whisper.py:341-349 on the rows written so far:
        weights [heads, tokens, frames] -> (w - mean over tokens) / std over tokens (unbiased=False) -> median filter along the
        frames (reflect padding, `_median_filter` :43-60) -> mean over heads.  Returns [224, 1500] (unwritten rows zero)."""
        rows = np.nonzero(self.alignment_written)[0]
        out = np.zeros_like(self.alignment)
        if len(rows) == 0:
            return out
        w = torch.from_numpy(self.alignment_heads[rows]).permute(1, 0, 2).double()      # [heads, tokens, frames]
        if z_normalize:
            std = torch.std(w, dim=-2, keepdim=True, unbiased=False)
            mean = torch.mean(w, dim=-2, keepdim=True)
            w = torch.where(std > 0, (w - mean) / std, torch.zeros_like(w))
        if median_filter_width and median_filter_width > 1:
            p = median_filter_width // 2
            w = F.pad(w, (p, p), mode="reflect")
            w = w.unfold(-1, median_filter_width, 1).sort()[0][..., p]
        out[rows] = w.mean(0).float().numpy()
        return out
