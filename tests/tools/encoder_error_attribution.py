#!/usr/bin/env python3
"""Where does the encoder's share of the end-to-end logits error come from?  (VERDICT r05 "next round" 2; CPU only, no GPU minutes.)

tests/test_gpu_realistic.py measures, at large-v3 on the realistic-statistics weights, a device error of 2.07 x the Float16-encoder-output
floor (one Float16 rounding of the fp32 oracle's own encoder output).  This tool re-runs the ORACLE encoder (torch fp32, oracle/model.py)
with the device's Float16 rounding points switched on ONE AT A TIME, and all together, and reports each variant's end-to-end logits error
against the all-fp32 oracle, as a multiple of that floor:

    mel        the log-mel operand of conv1 (csrc/mel.hip writes the time-major conv operand as f16)
    conv1      GELU(conv1) -> h1, the operand of conv2
    ln         both LayerNorm outputs of every layer (the A operands of the QKV and fc1 GEMMs)
    q / k / v  the attention operands the QKV GEMM epilogue writes (q pre-scaled by head_dim^-0.5)
    p          the unnormalised softmax probabilities exp(s - max) fed to the P V matrix product
    att        the attention output (the A operand of the out projection)
    gelu       GELU(fc1) -> the A operand of fc2
    out        the encoder output itself (enc16: the reference's AudioEncoderOutput type) = the floor

Matrix products accumulate in fp32 in every variant (as the MFMA does); weights are f16-exact in the fixture.  Usage:

    python tools/encoder_error_attribution.py [--model large-v3] [--tokens 96] [--out profiles/r06_encoder_error_attribution.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import decode as OD  # noqa: E402
from oracle import mel as omel  # noqa: E402
from oracle.model import OracleWhisper  # noqa: E402
from realistic import realistic_state_dict  # noqa: E402
from whisperkit_amd import weights  # noqa: E402
from whisperkit_amd.synth import synthetic_chunk  # noqa: E402

STAGES = ["mel", "conv1", "ln", "q", "k", "v", "p", "att", "gelu"]


def r16(x, on):
    return x.half().float() if on else x


def encode(om, mel, on, exact_last=0):
    """oracle/model.py OracleWhisper.encode with the rounding points in `on` (a set of STAGES) applied; the last `exact_last` layers keep
    fp32 everywhere (what hi | lo operand pairs in those layers' GEMMs would buy)"""
    w, dims = om.w, om.dims
    H = dims.n_audio_head
    with torch.no_grad():
        x = r16(torch.from_numpy(np.ascontiguousarray(mel, dtype=np.float32)), "mel" in on)[None]
        x = r16(F.gelu(F.conv1d(x, w["encoder.conv1.weight"], w["encoder.conv1.bias"], padding=1)), "conv1" in on)
        x = F.gelu(F.conv1d(x, w["encoder.conv2.weight"], w["encoder.conv2.bias"], stride=2, padding=1))
        x = x[0].T + w["encoder.positional_embedding"]
        T, d = x.shape
        hd = d // H
        all_on = on
        for i in range(dims.n_audio_layer):
            on = all_on if i < dims.n_audio_layer - exact_last else set()
            p = f"encoder.blocks.{i}"
            xn = r16(F.layer_norm(x, (d,), w[p + ".attn_ln.weight"], w[p + ".attn_ln.bias"]), "ln" in on)
            q = r16(F.linear(xn, w[p + ".attn.query.weight"], w[p + ".attn.query.bias"]) * hd ** -0.5, "q" in on)
            k = r16(F.linear(xn, w[p + ".attn.key.weight"]), "k" in on)
            v = r16(F.linear(xn, w[p + ".attn.value.weight"], w[p + ".attn.value.bias"]), "v" in on)
            qh, kh, vh = (t.view(T, H, hd).permute(1, 0, 2) for t in (q, k, v))
            s = qh @ kh.transpose(1, 2)
            e = r16(torch.exp(s - s.max(dim=-1, keepdim=True).values), "p" in on)
            o = (e @ vh) / e.sum(dim=-1, keepdim=True)
            o = r16(o.permute(1, 0, 2).reshape(T, d), "att" in on)
            x = x + F.linear(o, w[p + ".attn.out.weight"], w[p + ".attn.out.bias"])
            xn = r16(F.layer_norm(x, (d,), w[p + ".mlp_ln.weight"], w[p + ".mlp_ln.bias"]), "ln" in on)
            h = r16(F.gelu(F.linear(xn, w[p + ".mlp.0.weight"], w[p + ".mlp.0.bias"])), "gelu" in on)
            x = x + F.linear(h, w[p + ".mlp.2.weight"], w[p + ".mlp.2.bias"])
        x = F.layer_norm(x, (d,), w["encoder.ln_post.weight"], w["encoder.ln_post.bias"])
    return x.numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--tokens", type=int, default=96)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_encoder_error_attribution.json"))
    ap.add_argument("--variants", default="", help="comma-separated extra variants, each a '+'-joined stage set (e.g. ln+gelu)")
    ap.add_argument("--exact-last", default="", help="comma-separated layer counts: all stages on, but the last N encoder layers in fp32")
    ap.add_argument("--skip-single", action="store_true", help="only the extra variants")
    args = ap.parse_args()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    dims = weights.MODEL_DIMS[args.model]
    sd = realistic_state_dict(dims, seed=0)
    om = OracleWhisper(dims, sd)
    st, langs = OD.special_tokens_for_vocab(dims.n_vocab)
    mel = omel.log_mel_spectrogram(synthetic_chunk(args.seed), dims.n_mels).astype(np.float32)
    t0 = time.time()
    ref_enc = encode(om, mel, set())
    assert np.abs(ref_enc - om.encode(mel)).max() < 1e-4 * max(1.0, float(np.abs(ref_enc).max())), "the restated encoder must be the oracle's"
    print(f"fp32 encoder: {time.time() - t0:.1f} s", flush=True)
    # decoder inputs: the fp32 oracle's own greedy run on its own encoder output (the restated WhisperKit loop)
    kw = dict(firstTokenLogProbThreshold=None, logProbThreshold=None, compressionRatioThreshold=None, temperatureFallbackCount=0, sampleLength=args.tokens)
    oopts = OD.DecodingOptions(**kw)
    state = om.new_state(ref_enc)
    ores = OD.decode_text(lambda t, p: state.step(t, p, want_alignment=False), OD.prefill_prompt(oopts, st, dims.is_multilingual),
                          OD.GreedyTokenSampler(0.0, st.endToken, oopts), oopts, st, dims.is_multilingual, langs)
    inputs = ores.tokens[:-1][: args.tokens - 1]
    n = len(inputs)
    full = om.new_state(ref_enc).forward_full(inputs, want_alignment=False)
    sigma = float(np.std(np.stack([full[p] for p in range(0, n, 8)])))
    rms = float(np.sqrt((ref_enc ** 2).mean()))
    print(f"{n} decoder inputs, logits sigma {sigma:.2f}, encoder output rms {rms:.3f} ({time.time() - t0:.1f} s)", flush=True)

    def logits_err(enc):
        got = om.new_state(enc).forward_full(inputs, want_alignment=False)
        return max(float(np.abs(got[p] - full[p]).max()) for p in range(n))

    floor = logits_err(ref_enc.astype(np.float16).astype(np.float32))
    rows = {"out (floor: one Float16 rounding of the fp32 encoder output)": {"logits_max_abs_err": floor, "over_floor": 1.0, "rel_sigma": floor / sigma}}
    variants = ([] if args.skip_single else [({s}, 0) for s in STAGES] + [(set(STAGES), 0)]) + [(set(v.split("+")), 0) for v in args.variants.split(",") if v] + \
        [(set(STAGES), int(n_)) for n_ in args.exact_last.split(",") if n_]
    for on, exact_last in variants:
        name = "+".join(s for s in STAGES if s in on) if len(on) < len(STAGES) else "all stages (the device's rounding points)"
        if exact_last:
            name += f", last {exact_last} layers fp32"
        t1 = time.time()
        enc = encode(om, mel, on, exact_last)
        e_enc = np.abs(enc - ref_enc)
        # the device hands the decoder enc16: the variant's output rounded to Float16, as the reference's AudioEncoderOutput
        e_log = logits_err(enc.astype(np.float16).astype(np.float32))
        e_raw = logits_err(enc)
        rows[name] = {"logits_max_abs_err": e_log, "over_floor": e_log / floor, "rel_sigma": e_log / sigma,
                      "logits_max_abs_err_before_the_output_rounding": e_raw, "before_output_rounding_over_floor": e_raw / floor,
                      "encoder_max_abs_err": float(e_enc.max()), "encoder_rms_err_over_rms": float(np.sqrt((e_enc ** 2).mean())) / rms}
        print(f"{name:45s} logits err {e_log:.4f} = {e_log / floor:.2f} x floor (stage alone, fp32 output: {e_raw / floor:.2f} x)  [{time.time() - t1:.0f} s]", flush=True)
    doc = {"model": args.model, "weights": "tests/realistic.py realistic_state_dict(seed 0)", "chunk_seed": args.seed, "decoder_inputs": n,
           "logits_sigma": sigma, "encoder_output_rms": rms, "floor_logits_max_abs_err": floor, "variants": rows,
           "note": "oracle only (torch fp32 on the CPU); every variant's encoder output is rounded to Float16 before the decoder reads it, as the device's enc16 is"}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print("written", args.out)


if __name__ == "__main__":
    main()
