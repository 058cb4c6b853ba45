"""Full-depth GPU parity on weights with the activation statistics of a trained model (VERDICT r03 "next round" 1b): large-v3 (32 + 32
layers, 64 slots - the absorbed cross-attention path) and small (12 + 12, 8 slots, word-timestamp alignment rows, in BOTH cross-attention
modes) with tests/realistic.py's weights: token embedding x 32 (logits sigma 20 - 30 instead of 0.7), log-normal LayerNorm gains, 30 - 50 x
outlier channels in both residual streams, sharp audio-dependent cross-attention.

The benign N(0, 0.02) fixtures of tests/test_gpu_fulldepth.py meet BASELINE's 1e-3 ABSOLUTE logits tolerance with logits of sigma 0.7,
i.e. about 1e-3 relative.  A trained checkpoint's logits are an order of magnitude larger, so what carries over is the RELATIVE figure:
this test measures max |delta logits| over ALL positions of the run and all checked slots, absolute and divided by the logits' standard
deviation, and writes everything to gpurun_out/r04_realistic_errors.json (committed copy under profiles/).  What it found (MI355X):

  * stage-isolated (oracle decoder on the GPU's encoder output), absorbed cross-attention: 6.4e-4 sigma against the fp32 oracle at
    large-v3 - the decoder meets 1e-3 RELATIVE under realistic statistics: asserted (REL_BOUND);
  * the per-layer K / V path stores keys and values as Float16 (the reference's cache type): it agrees with the oracle that stores them
    the same way (3.2e-4 sigma, asserted) but is 7e-3 sigma away from the fp32 oracle once the attention is sharp - the rounding of the
    cross keys, which the absorbed path does not have (recorded; the reason the absorbed path is the more accurate one);
  * end to end from PCM (the oracle's own fp64 mel + fp32 encoder) the logits differ by ~2e-2 sigma: the encoder output is a Float16
    tensor (the reference's AudioEncoderOutput type, Core/Models.swift:938) computed from fp16 GEMM operands - 3e-4 rms relative error,
    one Float16 rounding's worth - and a sharp cross-attention amplifies key errors by the score magnitude.  No Float16 encoder output can
    meet 1e-3 of sigma here, the reference's included; recorded and asserted at 2 x the measured value, with the fraction of positions
    whose arg-max equals the fp32 end-to-end oracle's.

Greedy token ids must equal the oracle's restated loop on the stage-isolated logits, a difference passing only as a near-tie proven from the
oracle's own logits at twice the asserted bound.
"""
import json
import os

import numpy as np
import pytest

from oracle import decode as OD
from oracle import mel as omel
from realistic import realistic_state_dict
from test_gpu_fulldepth import ALIGNMENT_HEADS, FollowingSampler, Rig   # noqa: F401
from whisperkit_amd import weights

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# id -> (model, (slots, checked slots, word timestamps), cross-attention mode: None = the library's choice)
CONFIGS = {"large-v3": ("large-v3", (64, [0, 63], False), None),
           "small-absorbed": ("small", (8, [0, 7], True), 1),
           "small-kv-rows": ("small", (8, [0, 7], True), 0)}
REL_BOUND = 1.0e-3          # max |delta logits| / sigma(logits), stage-isolated, against the oracle with the path's own key / value storage
# end to end from PCM: measured on MI355X when the test was written (profiles/r04_realistic_errors.json), asserted at 2 x
E2E_MEASURED = {"large-v3": 2.45e-2, "small-absorbed": 1.15e-2, "small-kv-rows": 1.58e-2}
SAMPLE_LENGTH = 96          # 95 decoder steps per slot: the oracle's full-depth passes per checked slot stay within a minute
_REPORT = {}


def _write():
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "r04_realistic_errors.json"), "w") as f:
        json.dump(_REPORT, f, indent=1, sort_keys=True)


@pytest.fixture(scope="module", params=list(CONFIGS))
def rig(request):
    name, cfg, mode = CONFIGS[request.param]
    r = Rig(name, sd=realistic_state_dict(weights.MODEL_DIMS[name], seed=0), tag=request.param, config=cfg, report=_REPORT,
            sample_length=SAMPLE_LENGTH, mode=mode)
    r.absorbed = r.sess.crossAttentionMode == 1
    if request.param == "large-v3":
        assert r.absorbed                 # 64 slots at d = 1280: the library's own choice is the absorbed path
    yield r
    _write()
    r.sess.close(); r.model.close()


def test_realistic_stage_isolated_logits_and_greedy_tokens(rig):
    worst16, worst32, align16, align32, sigma, ties, compared, worst_lp = 0.0, 0.0, 0.0, 0.0, 0.0, {}, 0, 0.0
    n = rig.n_in
    for b in rig.check:
        res = rig.res[b]
        enc16 = rig.enc[b].astype(np.float16).astype(np.float32)
        inputs = res.tokens[:n]
        st16 = rig.om.new_state(enc16, kvFloat16=True)
        full16 = st16.forward_full(inputs)
        state = rig.om.new_state(enc16)
        full = state.forward_full(inputs)
        sig = float(np.std(np.stack([full[p] for p in range(0, n, 8)])))
        sigma = max(sigma, sig)
        for p in range(n):
            worst16 = max(worst16, float(np.abs(rig.dev_logits[b][p] - full16[p]).max()))
            worst32 = max(worst32, float(np.abs(rig.dev_logits[b][p] - full[p]).max()))
        rows = list(range(1, min(n, 223)))
        align32 = max(align32, float(np.abs(rig.align_tf[b][rows] - state.alignment[rows]).max()))
        align16 = max(align16, float(np.abs(rig.align_tf[b][rows] - st16.alignment[rows]).max()))
        ref = full if rig.absorbed else full16          # the oracle with this path's key / value storage

        def step(t, p, _full=ref, _inputs=inputs, _b=b):
            assert t == _inputs[p], (rig.name, _b, p, t, _inputs[p])
            return _full[p]
        sampler = FollowingSampler(rig.st.endToken, rig.oopts, res.tokens, len(rig.prompt), logit_tol=2.0 * REL_BOUND * sig)
        ores = OD.decode_text(step, rig.prompt, sampler, rig.oopts, rig.st, rig.ml, rig.langs)
        assert ores.tokens == res.tokens, (rig.name, b)
        lp_o = [list(d.values())[0] for d in ores.tokenLogProbs]
        worst_lp = max(worst_lp, float(np.abs(np.asarray(res.tokenLogProbs) - np.asarray(lp_o)).max()))
        ties[b] = sampler.near_ties
        compared += sampler.compared
    rig.report["stage_isolated"] = {"positions": f"all {n}", "logits_sigma": sigma,
                                    "logits_max_abs_err_vs_f32_kv_oracle": worst32, "logits_rel_err_vs_f32_kv_oracle": worst32 / sigma,
                                    "logits_max_abs_err_vs_f16_kv_oracle": worst16, "logits_rel_err_vs_f16_kv_oracle": worst16 / sigma,
                                    "alignment_rows_max_abs_err_vs_f32_kv_oracle": align32, "alignment_rows_max_abs_err_vs_f16_kv_oracle": align16,
                                    "token_logprob_max_abs_err": worst_lp,
                                    "greedy_tokens_compared": compared, "proven_near_ties_at_steps": {str(k): v for k, v in ties.items()}}
    _write()
    worst, align = (worst32, align32) if rig.absorbed else (worst16, align16)
    assert worst / sigma <= REL_BOUND, (rig.name, worst, sigma)
    assert align <= (1e-4 if rig.absorbed else 3e-4), (rig.name, align)      # (K / V rows: Float16 keys under a sharp softmax, 1.1e-4 measured)
    assert all(len(v) <= 2 for v in ties.values()), (rig.name, ties)


def test_realistic_end_to_end_from_pcm(rig):
    enc_max, enc_rel, enc_rms_rel, logit_max, sigma, same, total = 0.0, 0.0, 0.0, 0.0, 0.0, 0, 0
    n = rig.n_in
    for b in rig.check:
        ref_enc = rig.om.encode(omel.log_mel_spectrogram(rig.xs[b], rig.dims.n_mels).astype(np.float32))
        err = np.abs(rig.enc[b] - ref_enc)
        rms = float(np.sqrt((ref_enc ** 2).mean()))
        full = rig.om.new_state(ref_enc).forward_full(rig.res[b].tokens[:n], want_alignment=False)
        sigma = max(sigma, float(np.std(np.stack([full[p] for p in range(0, n, 8)]))))
        logit_max = max(logit_max, max(float(np.abs(rig.dev_logits[b][p] - full[p]).max()) for p in range(n)))
        same += sum(int(np.argmax(rig.dev_logits[b][p]) == np.argmax(full[p])) for p in range(n))
        total += n
        enc_max, enc_rel = max(enc_max, float(err.max())), max(enc_rel, float(err.max()) / rms)
        enc_rms_rel = max(enc_rms_rel, float(np.sqrt((err ** 2).mean())) / rms)
    rig.report["end_to_end"] = {"positions": f"all {n}", "encoder_max_abs_err": enc_max, "encoder_max_err_over_rms": enc_rel,
                                "encoder_rms_err_over_rms": enc_rms_rel, "logits_max_abs_err": logit_max, "logits_sigma": sigma,
                                "logits_rel_err": logit_max / sigma, "argmax_equal_positions": same, "positions_compared": total}
    _write()
    assert logit_max / sigma <= 2.0 * E2E_MEASURED[rig.name], (rig.name, logit_max, sigma)
    assert same >= 0.97 * total, (rig.name, same, total)
