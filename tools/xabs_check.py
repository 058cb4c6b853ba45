"""Bring-up check of the weight-absorbed cross-attention (csrc/xabs.hip) on the GPU box: two sessions of one model, WH_XABS=0 (per-layer
cross K / V rows) and WH_XABS=1 (absorbed), the same audio, the same teacher-forced steps.  Compares the logits, the last layer's
cross-attention output planes, and - stage by stage, against numpy on the peeked buffers - the absorbed queries Q' = W_k^T q, every key
split's (log-sum-exp, normalised O') and the alignment rows.  Prints one JSON line per check.

    python tools/xabs_check.py [model] [batch]          (default test-large-v3-l2, 8)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperkit_amd import api, weights          # noqa: E402
from whisperkit_amd import _lib as L             # noqa: E402
from whisperkit_amd.synth import synthetic_chunk  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "test-large-v3-l2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dims = weights.MODEL_DIMS[name]
sd = weights.synthetic_state_dict(dims, seed=11)
model = api.Model(dims, sd)
d, H, Lyr = dims.n_text_state, dims.n_text_head, dims.n_text_layer
S = 4            # key splits per slot: replaced by the session's count below


def peek(sess, nm, shape, dtype):
    out = np.empty(shape, dtype)
    api._check(sess.lib.wh_debug_peek(sess.handle, nm.encode(), out.ctypes.data, out.nbytes))
    return out


def make(mode):
    os.environ["WH_XABS"] = str(mode)
    s = api.Session(model, B)
    assert s.lib.wh_session_cross_attention_mode(s.handle) == mode, "mode not taken"
    for b in range(B):
        s.padOrTrim(synthetic_chunk(4000 + 17 * b), b)
    s.logMelSpectrogram(B); s.encodeFeatures(B); s.prepareDecoderInputs(B)
    return s


def planes(sess, nm):       # [n_bt][d/16][2][32][8] hi | lo -> [R][d]
    R = (B + 31) // 32 * 32
    hi = peek(sess, nm + "_hi", (R // 32, d // 16, 2, 32, 8), np.float16).astype(np.float32)
    lo = peek(sess, nm + "_lo", (R // 32, d // 16, 2, 32, 8), np.float16).astype(np.float32)
    z = hi + lo / 2048.0
    return z.transpose(0, 3, 1, 2, 4).reshape(R, d)


s0, s1 = make(0), make(1)
S = s1.crossAttentionSplits
st = model.specialTokens
steps = [(st.start_of_transcript_token, 0), (st.english_token, 1), (st.transcribe_token, 2), (1029, 3), (400, 150)]
for tok, pos in steps:
    toks = [(tok + 3 * b) % 50000 if pos > 2 else tok for b in range(B)]
    l0 = s0.predictLogits(toks, [pos] * B)
    l1 = s1.predictLogits(toks, [pos] * B)
    a0, a1 = planes(s0, "zb")[:B], planes(s1, "zb")[:B]
    print(json.dumps({"check": "logits", "pos": pos, "max_abs_diff": float(np.abs(l0 - l1).max()), "sigma": float(l0.std()),
                      "att_last_layer_max_abs_diff": float(np.abs(a0 - a1).max()), "att_rms": float(np.sqrt((a0 ** 2).mean())),
                      "nan": bool(np.isnan(l1).any())}))

# ---- stage checks on the last step / last layer
l = Lyr - 1
R = (B + 31) // 32 * 32
q = peek(s1, "q", (R, d), np.float32)[:B]
q_old = peek(s0, "q", (R, d), np.float32)[:B]
print(json.dumps({"check": "q_equal_between_modes", "max_abs_diff": float(np.abs(q - q_old).max())}))
Wk = sd[f"decoder.blocks.{l}.cross_attn.key.weight"].astype(np.float32)
Wv = sd[f"decoder.blocks.{l}.cross_attn.value.weight"].astype(np.float32)
bv = sd[f"decoder.blocks.{l}.cross_attn.value.bias"].astype(np.float32)
NHT = 2 if H > 16 else 1
KS = d // 32
HP = NHT * 16
hi = peek(s1, "qf_hi", (B, HP, d), np.float16).astype(np.float32)
lo = peek(s1, "qf_lo", (B, HP, d), np.float16).astype(np.float32)
# rows 0..15: hi (qf_hi) | lo (qf_lo) of heads 0..15; packed second tile: qf_hi rows 16..23 = hi, rows 24..31 = lo of heads 16..23
qf = np.concatenate([hi[:, :16] + lo[:, :16] / 2048.0] + ([hi[:, 16:24] + hi[:, 24:32] / 2048.0] if NHT == 2 else []), axis=1)
qp = qf[:, :H]
qp_ref = np.einsum("bhj,hjc->bhc", q.reshape(B, H, 64), Wk.reshape(H, 64, d))
print(json.dumps({"check": "absorbed_queries", "max_abs_diff": float(np.abs(qp - qp_ref).max()), "rms": float(np.sqrt((qp_ref ** 2).mean()))}))

enc = peek(s1, "enc16", (B, 1500, d), np.float16).astype(np.float32)
part = peek(s1, "part", (S, H, d // 8, B, 8), np.float32).transpose(0, 3, 1, 2, 4).reshape(S, B, H, d)
ml = peek(s1, "ml", (S, H, B, 2), np.float32)
NT = (1500 + 15) // 16
worst_lse, worst_o = 0.0, 0.0
att_ref = np.zeros((B, d), np.float32)
for b in range(B):
    sc = qp_ref[b] @ enc[b].T                               # [H][1500]
    p = np.exp(sc - sc.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
    o = p @ enc[b]                                          # [H][d]
    att_ref[b] = np.einsum("hjc,hc->hj", Wv.reshape(H, 64, d), o).reshape(d) + bv
    for sp in range(S):
        k0, k1 = sp * NT // S * 16, min((sp + 1) * NT // S * 16, 1500)
        s_ = sc[:, k0:k1]
        lse = np.log(np.exp(s_ - s_.max(1, keepdims=True)).sum(1)) + s_.max(1)
        got_lse = ml[sp, :, b, 0] + np.log(ml[sp, :, b, 1])
        worst_lse = max(worst_lse, float(np.abs(lse - got_lse).max()))
        pn = np.exp(s_ - lse[:, None])
        worst_o = max(worst_o, float(np.abs(pn @ enc[b, k0:k1] - part[sp, b] / ml[sp, :, b, 1][:, None]).max()))
print(json.dumps({"check": "split_partials", "logsumexp_max_abs_diff": worst_lse, "normalised_o_max_abs_diff": worst_o}))
att = planes(s1, "zb")[:B]
print(json.dumps({"check": "att_vs_numpy", "max_abs_diff": float(np.abs(att - att_ref).max()), "old_path_vs_numpy": float(np.abs(planes(s0, "zb")[:B] - att_ref).max()),
                  "rms": float(np.sqrt((att_ref ** 2).mean()))}))
al0, al1 = s0.getAlignmentWeights(B - 1), s1.getAlignmentWeights(B - 1)
rows = [p + 1 for _, p in steps]
print(json.dumps({"check": "alignment_rows", "max_abs_diff": float(np.abs(al0[rows] - al1[rows]).max()), "row_sums": [float(x) for x in al1[rows].sum(1)]}))
