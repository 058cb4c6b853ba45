"""Checkpoint ingest (whisperkit_amd/checkpoint.py): HF folders and openai .pt files -> state dict -> WHIPW001 blob.  CPU only."""
import json
import os

import numpy as np
import pytest

from whisperkit_amd import checkpoint, weights


def _hf_model(dims, sd):
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
    cfg = tr.WhisperConfig(vocab_size=dims.n_vocab, num_mel_bins=dims.n_mels, d_model=dims.n_audio_state,
                           encoder_layers=dims.n_audio_layer, encoder_attention_heads=dims.n_audio_head,
                           decoder_layers=dims.n_text_layer, decoder_attention_heads=dims.n_text_head,
                           encoder_ffn_dim=4 * dims.n_audio_state, decoder_ffn_dim=4 * dims.n_text_state,
                           max_source_positions=dims.n_audio_ctx, max_target_positions=dims.n_text_ctx)
    m = tr.WhisperForConditionalGeneration(cfg).eval()
    missing = m.load_state_dict(weights.to_hf_state_dict(sd), strict=False)
    assert not [k for k in missing.missing_keys if "proj_out" not in k]
    return m


def test_hf_folder_and_openai_pt_round_trip(tmp_path):
    torch = pytest.importorskip("torch")
    dims = weights.MODEL_DIMS["test-micro"]
    sd = weights.synthetic_state_dict(dims, seed=3)
    folder = tmp_path / "hf"
    _hf_model(dims, sd).save_pretrained(str(folder), safe_serialization=True)
    with open(folder / "generation_config.json", "w") as f:
        json.dump({"alignment_heads": [[1, 0], [1, 1]]}, f)
    d2, sd2, heads = checkpoint.load_checkpoint(str(folder))
    assert d2 == dims and heads == [(1, 0), (1, 1)]
    assert set(sd2) == set(sd)
    for k in sd:
        assert sd2[k].dtype == np.float32 and np.array_equal(sd2[k], sd[k]), k
    # openai layout
    pt = tmp_path / "w.pt"
    torch.save({"dims": {f: getattr(dims, f) for f in ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer", "n_vocab",
                                                         "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")},
                "model_state_dict": {k: torch.from_numpy(v) for k, v in sd.items() if k != "encoder.positional_embedding"}}, str(pt))
    d3, sd3, h3 = checkpoint.load_checkpoint(str(pt))
    assert d3 == dims and h3 is None and all(np.array_equal(sd3[k], sd[k]) for k in sd)     # the sinusoid buffer is regenerated
    # blob: identical bytes whichever way the weights arrived
    out = tmp_path / "m.whipw"
    assert checkpoint.main([str(folder), str(out)]) == 0
    blob = np.fromfile(str(out), dtype=np.uint8)
    assert np.array_equal(blob, weights.pack_blob(dims, sd, alignment_heads=[(1, 0), (1, 1)]))
    d4, tensors = weights.unpack_blob(blob)
    assert d4 == dims and json.load(open(str(out) + ".alignment_heads.json")) == [[1, 0], [1, 1]]
    assert tensors["dec.alignment_heads"].tolist() == [[1, 0], [1, 1]]     # wh_model_load picks the heads up from the blob itself
    # a bf16 safetensors file loads as well (values rounded to bf16, returned as fp32)
    from safetensors.torch import save_file
    hf = weights.to_hf_state_dict(sd)
    bf = tmp_path / "bf16"
    bf.mkdir()
    save_file({k: v.to(torch.bfloat16).contiguous() for k, v in hf.items() if k != "proj_out.weight"}, str(bf / "model.safetensors"))
    (bf / "config.json").write_text((folder / "config.json").read_text())
    d5, sd5, _ = checkpoint.load_hf_checkpoint(str(bf))
    k = "decoder.blocks.0.mlp.0.weight"
    assert d5 == dims and np.array_equal(sd5[k], torch.from_numpy(sd[k]).to(torch.bfloat16).float().numpy())


def test_checkpoint_errors(tmp_path):
    dims = weights.MODEL_DIMS["test-micro"]
    with pytest.raises(FileNotFoundError):
        checkpoint.load_hf_checkpoint(str(tmp_path))
    (tmp_path / "config.json").write_text(json.dumps({"d_model": 128, "num_mel_bins": 80, "encoder_attention_heads": 2, "encoder_layers": 2,
                                                      "vocab_size": dims.n_vocab, "decoder_attention_heads": 2, "decoder_layers": 2}))
    with pytest.raises(FileNotFoundError):
        checkpoint.load_hf_checkpoint(str(tmp_path))
    from safetensors.numpy import save_file
    save_file({"model.encoder.conv1.weight": np.zeros((64, 80, 3), np.float32)}, str(tmp_path / "model.safetensors"))
    with pytest.raises((KeyError, ValueError)):
        checkpoint.load_hf_checkpoint(str(tmp_path))


def test_loaded_checkpoint_reproduces_hf_logits(tmp_path):
    """The whole chain HF folder -> loader -> oracle model gives HF's own encoder output and logits (same weights, fp32)."""
    torch = pytest.importorskip("torch")
    from oracle.model import OracleWhisper
    dims = weights.MODEL_DIMS["test-micro"]
    sd = weights.synthetic_state_dict(dims, seed=4)
    hf = _hf_model(dims, sd)
    folder = tmp_path / "hf"
    hf.save_pretrained(str(folder), safe_serialization=True)
    d2, sd2, _ = checkpoint.load_checkpoint(str(folder))
    om = OracleWhisper(d2, sd2)
    mel = np.random.default_rng(0).standard_normal((dims.n_mels, 3000)).astype(np.float32) * 0.3
    toks = [50257, 50362, 400, 370]
    with torch.no_grad():
        enc_hf = hf.model.encoder(torch.from_numpy(mel)[None]).last_hidden_state[0].numpy()
        logits_hf = hf(input_features=torch.from_numpy(mel)[None], decoder_input_ids=torch.tensor([toks])).logits[0].numpy()
    enc = om.encode(mel)
    assert np.abs(enc - enc_hf).max() < 2e-4
    state = om.new_state(enc)
    for i, t in enumerate(toks):
        lg = state.step(t, i)
        assert np.abs(np.asarray(lg) - logits_hf[i]).max() < 5e-4, i
