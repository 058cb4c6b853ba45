"""Checkpoint ingest: openai/whisper and Hugging Face Whisper checkpoints -> the WHIPW001 weight blob `wh_model_load` reads.

The reference never touches these formats: it downloads pre-compiled CoreML bundles from `argmaxinc/whisperkit-coreml`
(Core/WhisperKit.swift:163,318; generated offline by whisperkittools from the same openai/HF checkpoints).  This module
is the equivalent offline step for the HIP path:

    python -m whisperkit_amd.checkpoint <hf-folder | model.safetensors | whisper.pt> out.whipw

  * HF folder: `config.json` + `model.safetensors` (or the sharded `model.safetensors.index.json`) or `pytorch_model.bin`,
    optional `generation_config.json` whose `alignment_heads` ([[layer, head], ...]) select the word-timestamp heads
  * openai/whisper `.pt`: a torch pickle {"dims": {...}, "model_state_dict": {...}}

Nothing here needs a GPU; the blob is uploaded by `wh_model_load` / `api.Model.from_pretrained`.
"""
from __future__ import annotations

import json
import os
import sys
from typing import Dict, List, Optional, Tuple

import numpy as np

from .weights import WhisperDims, from_hf_state_dict, pack_blob


def _to_f32(t) -> np.ndarray:
    return np.asarray(t.detach().cpu().float().numpy() if hasattr(t, "detach") else t, dtype=np.float32)


def dims_from_hf_config(cfg: dict) -> WhisperDims:
    """HF `WhisperConfig` -> openai/whisper ModelDimensions."""
    d = int(cfg["d_model"])
    return WhisperDims(int(cfg["num_mel_bins"]), int(cfg.get("max_source_positions", 1500)), d, int(cfg["encoder_attention_heads"]),
                       int(cfg["encoder_layers"]), int(cfg["vocab_size"]), int(cfg.get("max_target_positions", 448)), d,
                       int(cfg["decoder_attention_heads"]), int(cfg["decoder_layers"]))


def _read_safetensors(path: str) -> Dict[str, np.ndarray]:
    from safetensors import safe_open
    out = {}
    with safe_open(path, framework="pt") as f:       # "pt": bf16 checkpoints load too
        for k in f.keys():
            out[k] = _to_f32(f.get_tensor(k))
    return out


def load_hf_checkpoint(folder: str) -> Tuple[WhisperDims, Dict[str, np.ndarray], Optional[List[Tuple[int, int]]]]:
    """(dims, openai-named fp32 state dict, alignment heads or None) from a HF Whisper model folder."""
    with open(os.path.join(folder, "config.json"), encoding="utf-8") as f:
        cfg = json.load(f)
    dims = dims_from_hf_config(cfg)
    index = os.path.join(folder, "model.safetensors.index.json")
    single = os.path.join(folder, "model.safetensors")
    hf: Dict[str, np.ndarray] = {}
    if os.path.exists(index):
        with open(index, encoding="utf-8") as f:
            shards = sorted(set(json.load(f)["weight_map"].values()))
        for s in shards:
            hf.update(_read_safetensors(os.path.join(folder, s)))
    elif os.path.exists(single):
        hf = _read_safetensors(single)
    elif os.path.exists(os.path.join(folder, "pytorch_model.bin")):
        import torch
        hf = {k: _to_f32(v) for k, v in torch.load(os.path.join(folder, "pytorch_model.bin"), map_location="cpu", weights_only=True).items()}
    else:
        raise FileNotFoundError(f"{folder}: no model.safetensors(.index.json) or pytorch_model.bin")
    sd = from_hf_state_dict(hf)
    heads = None
    gen = os.path.join(folder, "generation_config.json")
    if os.path.exists(gen):
        with open(gen, encoding="utf-8") as f:
            ah = json.load(f).get("alignment_heads")
        if ah:
            heads = [(int(l), int(h)) for l, h in ah]
    _check_shapes(dims, sd)
    return dims, sd, heads


def load_openai_checkpoint(path: str) -> Tuple[WhisperDims, Dict[str, np.ndarray], Optional[List[Tuple[int, int]]]]:
    """openai/whisper `.pt`: {"dims": ModelDimensions fields, "model_state_dict": tensors in openai naming}."""
    import torch
    ck = torch.load(path, map_location="cpu", weights_only=True)
    d = ck["dims"]
    dims = WhisperDims(d["n_mels"], d["n_audio_ctx"], d["n_audio_state"], d["n_audio_head"], d["n_audio_layer"], d["n_vocab"], d["n_text_ctx"],
                       d["n_text_state"], d["n_text_head"], d["n_text_layer"])
    sd = {k: _to_f32(v) for k, v in ck["model_state_dict"].items()}
    _check_shapes(dims, sd)
    return dims, sd, None


def _check_shapes(dims: WhisperDims, sd: Dict[str, np.ndarray]):
    want = {
        "encoder.conv1.weight": (dims.n_audio_state, dims.n_mels, 3),
        "encoder.conv2.weight": (dims.n_audio_state, dims.n_audio_state, 3),
        "decoder.token_embedding.weight": (dims.n_vocab, dims.n_text_state),
        "decoder.positional_embedding": (dims.n_text_ctx, dims.n_text_state),
        f"decoder.blocks.{dims.n_text_layer - 1}.mlp.0.weight": (4 * dims.n_text_state, dims.n_text_state),
        f"encoder.blocks.{dims.n_audio_layer - 1}.attn.key.weight": (dims.n_audio_state, dims.n_audio_state),
    }
    for k, shape in want.items():
        if k not in sd:
            raise KeyError(f"checkpoint has no tensor '{k}' (not a Whisper checkpoint, or an unknown naming scheme)")
        if tuple(sd[k].shape) != shape:
            raise ValueError(f"tensor '{k}' has shape {tuple(sd[k].shape)}, the config says {shape}")
    if "encoder.positional_embedding" not in sd:       # HF always stores it; openai registers a buffer
        from .weights import sinusoids
        sd["encoder.positional_embedding"] = sinusoids(dims.n_audio_ctx, dims.n_audio_state)


def load_checkpoint(src: str):
    """Dispatch on what `src` is: HF folder, a bare .safetensors next to its config.json, or an openai .pt."""
    if os.path.isdir(src):
        return load_hf_checkpoint(src)
    if src.endswith(".safetensors") or src.endswith(".bin"):
        return load_hf_checkpoint(os.path.dirname(os.path.abspath(src)))
    return load_openai_checkpoint(src)


def convert(src: str, out_path: str) -> Tuple[WhisperDims, Optional[List[Tuple[int, int]]]]:
    """Write the WHIPW001 blob of checkpoint `src` to `out_path`; the checkpoint's alignment heads are stored inside the blob
    (`dec.alignment_heads`, read by wh_model_load) and, for inspection, in `<out>.alignment_heads.json`."""
    dims, sd, heads = load_checkpoint(src)
    blob = pack_blob(dims, sd, alignment_heads=heads)
    np.asarray(blob, dtype=np.uint8).tofile(out_path)
    if heads:
        with open(out_path + ".alignment_heads.json", "w", encoding="utf-8") as f:
            json.dump(heads, f)
    return dims, heads


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) != 2:
        print(__doc__)
        return 2
    dims, heads = convert(argv[0], argv[1])
    print(f"wrote {argv[1]}: {os.path.getsize(argv[1]) / 1e6:.1f} MB, dims {dims}, alignment heads {'default' if not heads else len(heads)}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
