"""Synthetic weights with the activation statistics of a TRAINED Whisper (test infrastructure; VERDICT r03 "What's weak" 1).

`weights.synthetic_state_dict` is benign: N(0, 0.02) matrices give logits with sigma ~ 0.7 at large-v3, LayerNorm gains near 1, no
outlier channels, near-uniform cross-attention.  A 1e-3 ABSOLUTE logits tolerance measured there says little about a real checkpoint,
whose logits are an order of magnitude larger.  `realistic_state_dict` keeps the same random matrices and adds what a trained model
has (powers of two wherever a matrix is scaled, so every f16-stored weight stays exactly representable):

  * the tied token embedding x 32           - decisive next-token distributions, logits sigma ~ 20
  * log-normal LayerNorm gains (sigma 0.5)  - gains between ~0.3 and ~3, as in trained checkpoints
  * outlier channels in BOTH residual streams: 6 channels carry a constant offset of 30 - 50 x the stream's typical magnitude (the
    encoder's through its positional table, the decoder's through the learned positions) - the massive-activation channels every large
    transformer has; they dominate the LayerNorm variance and stress the f16 operand rounding
  * a sharp, audio-dependent cross-attention: query / key x 16, output x 32 (as tests/test_gpu_round2._audio_sensitive)

The parity tests on it report the error relative to the logits' standard deviation beside the absolute one.
"""
import numpy as np

from whisperkit_amd import weights


def realistic_state_dict(dims, seed=0):
    sd = dict(weights.synthetic_state_dict(dims, seed=seed))
    rng = np.random.default_rng([seed, 9001])
    sd["decoder.token_embedding.weight"] = sd["decoder.token_embedding.weight"] * np.float32(32)
    for k in list(sd):
        if k.endswith("_ln.weight") or k in ("encoder.ln_post.weight", "decoder.ln.weight"):
            sd[k] = np.exp(rng.normal(0.0, 0.5, sd[k].shape)).astype(np.float32)
    d, dt = dims.n_audio_state, dims.n_text_state
    enc_pos = sd["encoder.positional_embedding"].copy()          # fp32 table (sinusoids): values in [-1, 1]
    for ch, f in zip(rng.choice(d, 6, replace=False), (30, -35, 40, -45, 50, -30)):
        enc_pos[:, ch] += np.float32(f)
    sd["encoder.positional_embedding"] = enc_pos
    dec_pos = sd["decoder.positional_embedding"].copy()          # token rows are ~N(0, 0.64) after the x 32
    for ch, f in zip(rng.choice(dt, 6, replace=False), (30, -35, 40, -45, 50, -30)):
        dec_pos[:, ch] += np.float32(0.64 * f)
    sd["decoder.positional_embedding"] = dec_pos
    for i in range(dims.n_text_layer):
        for w, f in ((".cross_attn.out.weight", 32), (".cross_attn.query.weight", 16), (".cross_attn.key.weight", 16)):
            k = f"decoder.blocks.{i}" + w
            sd[k] = sd[k] * np.float32(f)
    return sd
