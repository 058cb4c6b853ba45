"""Full-depth GPU parity on weights with the activation statistics of a trained model, in the cross-attention mode the LIBRARY picks for
every BASELINE configuration (VERDICT r04 "next round" 1a / 1b / 1d):

  * large-v3, 32 + 32 layers, 64 slots (configs[3]: the absorbed cross-attention, the library's choice from 28 slots)
  * small, 12 + 12, 8 slots, word-timestamp alignment rows (configs[2]: the library's choice, per-layer K / V rows) - and the same
    fixture with the absorbed path forced
  * tiny.en, 4 + 4, 1 slot (configs[1]: the library's choice; d = 384 has no absorbed kernel)

with tests/realistic.py's weights: token embedding x 32 (logits sigma 17 - 30 instead of 0.7), log-normal LayerNorm gains, 30 - 50 x
outlier channels in both residual streams, sharp audio-dependent cross-attention.

The benign N(0, 0.02) fixtures of tests/test_gpu_fulldepth.py meet BASELINE's 1e-3 ABSOLUTE logits tolerance with logits of sigma 0.7,
i.e. about 1e-3 relative.  A trained checkpoint's logits are an order of magnitude larger, so what carries over is the RELATIVE figure:
this test measures max |delta logits| over ALL positions of the run and all checked slots, absolute and divided by the logits' standard
deviation, and writes everything to gpurun_out/r05_realistic_errors.json (committed copy under profiles/).  The contract it asserts:

  * STAGE-ISOLATED (oracle decoder on the GPU's encoder output): <= 1e-3 sigma against the fp32 oracle (openai/whisper in fp32: keys
    and values never rounded) in EVERY shipped mode.  Round 4 met this on the absorbed path only: the K / V-row path stored the cross
    keys / values as Float16 and sat at 7.1e-3 sigma (the keys' rounding under a sharp softmax; the values' alone is 1.9e-3 sigma,
    measured with the oracle on the CPU).  Since round 5 the rows carry 19 mantissa bits in 3 bytes (a Float16 + an 8-bit residual,
    csrc/kernels.h hr24; csrc/gemm.hip EPI_CROSS_KV writes them, csrc/decoder.hip reads them).
  * END TO END from PCM (the oracle's own fp64 mel + fp32 encoder): the encoder output is a Float16 tensor (the reference's
    AudioEncoderOutput type, Core/Models.swift:938) and a sharp cross-attention amplifies key errors by the score magnitude, so no
    Float16 encoder output can meet 1e-3 sigma here, the reference's included.  The floor is measured IN THIS TEST with the oracle alone:
    ONE Float16 rounding of the fp32 oracle's own encoder output moves its logits by `floor` (1e-2 - 2e-2 sigma); the device must stay
    within K_FLOOR x that floor, and every position whose arg-max differs from the fp32 end-to-end oracle's must be a near-tie PROVEN from
    the oracle's own logits: the oracle's margin between its choice and the device's choice is below twice the measured error (two
    values move against each other).  No percentage threshold.

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
           "small": ("small", (8, [0, 7], True), None),
           "small-absorbed": ("small", (8, [0, 7], True), 1),
           "tiny.en": ("tiny.en", (1, [0], False), None)}
LIBRARY_MODE = {"large-v3": 1, "small": 0, "tiny.en": 0}      # what wh_session_create picks for configs[3] / [2] / [1]
REL_BOUND = 1.0e-3          # max |delta logits| / sigma(logits), stage-isolated, against the fp32 oracle, every mode
K_FLOOR = 2.5               # end to end: device error <= K_FLOOR x the Float16-encoder-output floor of the same audio (measured 1.1 - 1.3)
SAMPLE_LENGTH = 96          # 95 decoder steps per slot: the oracle's full-depth passes per checked slot stay within a minute
_REPORT = {}


def _write():
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "r05_realistic_errors.json"), "w") as f:
        json.dump(_REPORT, f, indent=1, sort_keys=True)


@pytest.fixture(scope="module", params=list(CONFIGS))
def rig(request):
    name, cfg, mode = CONFIGS[request.param]
    r = Rig(name, sd=realistic_state_dict(weights.MODEL_DIMS[name], seed=0), tag=request.param, config=cfg, report=_REPORT,
            sample_length=SAMPLE_LENGTH, mode=mode)
    r.absorbed = r.sess.crossAttentionMode == 1
    if mode is None:
        assert r.sess.crossAttentionMode == LIBRARY_MODE[request.param], request.param      # the mode the bench's configuration runs
    r.report["mode_is_the_librarys_choice"] = mode is None
    yield r
    _write()
    r.sess.close(); r.model.close()


def test_realistic_stage_isolated_logits_and_greedy_tokens(rig):
    worst_own, worst32, align_own, align32, sigma, ties, compared, worst_lp = 0.0, 0.0, 0.0, 0.0, 0.0, {}, 0, 0.0
    n = rig.n_in
    for b in rig.check:
        res = rig.res[b]
        enc16 = rig.enc[b].astype(np.float16).astype(np.float32)
        inputs = res.tokens[:n]
        st_own = rig.om.new_state(enc16, kvFloat16=True, crossFloat16=False)       # the device's storage: Float16 self-attention cache, cross keys / values never rounded
        full_own = st_own.forward_full(inputs)
        state = rig.om.new_state(enc16)                                            # openai/whisper in fp32
        full = state.forward_full(inputs)
        sig = float(np.std(np.stack([full[p] for p in range(0, n, 8)])))
        sigma = max(sigma, sig)
        for p in range(n):
            worst_own = max(worst_own, float(np.abs(rig.dev_logits[b][p] - full_own[p]).max()))
            worst32 = max(worst32, float(np.abs(rig.dev_logits[b][p] - full[p]).max()))
        rows = list(range(1, min(n, 223)))
        align32 = max(align32, float(np.abs(rig.align_tf[b][rows] - state.alignment[rows]).max()))
        align_own = max(align_own, float(np.abs(rig.align_tf[b][rows] - st_own.alignment[rows]).max()))

        def step(t, p, _full=full, _inputs=inputs, _b=b):
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
                                    "logits_max_abs_err_vs_f32_oracle": worst32, "logits_rel_err_vs_f32_oracle": worst32 / sigma,
                                    "logits_max_abs_err_vs_f16_self_cache_oracle": worst_own, "logits_rel_err_vs_f16_self_cache_oracle": worst_own / sigma,
                                    "alignment_rows_max_abs_err_vs_f32_oracle": align32, "alignment_rows_max_abs_err_vs_f16_self_cache_oracle": align_own,
                                    "token_logprob_max_abs_err": worst_lp,
                                    "greedy_tokens_compared": compared, "proven_near_ties_at_steps": {str(k): v for k, v in ties.items()}}
    _write()
    assert worst32 / sigma <= REL_BOUND, (rig.name, worst32, sigma)          # the contract, against fp32, whatever the mode
    assert align32 <= 1e-4, (rig.name, align32)
    assert all(len(v) <= 2 for v in ties.values()), (rig.name, ties)


def test_realistic_end_to_end_from_pcm(rig):
    enc_max, enc_rel, enc_rms_rel, logit_max, floor_max, sigma, same, total = 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 0
    mismatches = []
    n = rig.n_in
    for b in rig.check:
        ref_enc = rig.om.encode(omel.log_mel_spectrogram(rig.xs[b], rig.dims.n_mels).astype(np.float32))
        err = np.abs(rig.enc[b] - ref_enc)
        rms = float(np.sqrt((ref_enc ** 2).mean()))
        inputs = rig.res[b].tokens[:n]
        full = rig.om.new_state(ref_enc).forward_full(inputs, want_alignment=False)
        # the floor: the SAME oracle decoder on its own encoder output rounded once to Float16 (the reference's AudioEncoderOutput type)
        rounded = rig.om.new_state(ref_enc.astype(np.float16).astype(np.float32)).forward_full(inputs, want_alignment=False)
        sigma = max(sigma, float(np.std(np.stack([full[p] for p in range(0, n, 8)]))))
        for p in range(n):
            e = float(np.abs(rig.dev_logits[b][p] - full[p]).max())
            logit_max = max(logit_max, e)
            floor_max = max(floor_max, float(np.abs(rounded[p] - full[p]).max()))
            i_dev, i_or = int(np.argmax(rig.dev_logits[b][p])), int(np.argmax(full[p]))
            if i_dev != i_or:
                mismatches.append({"slot": b, "position": p, "device": i_dev, "oracle": i_or, "position_error": e,
                                   "oracle_margin": float(full[p][i_or] - full[p][i_dev]),
                                   "oracle_top2_gap": float(np.diff(np.sort(full[p])[-2:])[0])})
            else:
                same += 1
            total += 1
        enc_max, enc_rel = max(enc_max, float(err.max())), max(enc_rel, float(err.max()) / rms)
        enc_rms_rel = max(enc_rms_rel, float(np.sqrt((err ** 2).mean())) / rms)
    rig.report["end_to_end"] = {"positions": f"all {n}", "encoder_max_abs_err": enc_max, "encoder_max_err_over_rms": enc_rel,
                                "encoder_rms_err_over_rms": enc_rms_rel, "logits_max_abs_err": logit_max, "logits_sigma": sigma,
                                "logits_rel_err": logit_max / sigma, "f16_encoder_output_floor_abs": floor_max, "f16_encoder_output_floor_rel": floor_max / sigma,
                                "error_over_floor": logit_max / floor_max, "asserted_k": K_FLOOR,
                                "argmax_equal_positions": same, "positions_compared": total, "argmax_mismatches": mismatches}
    _write()
    assert logit_max <= K_FLOOR * floor_max, (rig.name, logit_max, floor_max, sigma)
    for m in mismatches:        # a different arg-max is accepted only as a near-tie of the ORACLE's own logits, at the error measured in this run
        assert m["oracle_margin"] <= 2.0 * logit_max and m["oracle_top2_gap"] <= 2.0 * logit_max, (rig.name, m, logit_max)
