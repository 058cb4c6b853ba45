"""Proof obligations for a token difference between the HIP path and the fp32 oracle (test infrastructure).

The contract (tests/test_gpu_parity.py header): sampled ids are identical to the oracle's restated loop; a difference is
accepted only where the ORACLE's own filtered logits show that a perturbation below the logits tolerance changes the
outcome.  Nothing here skips or xfails: `assert_tokens_or_proven_near_tie` either proves the near-tie from the oracle's
recorded logits or fails the test.
"""
import numpy as np

from oracle import decode as OD

LOGIT_TOL = 2e-3     # twice the teacher-forced logits tolerance (two values move against each other)


def _explain(filtered, got, want, temperature, seed, counter, top_k, logit_tol=LOGIT_TOL):
    """True when a <= logit_tol change of the oracle's filtered logits can turn `want` into `got` at this sampling step.
    (logit_tol: twice the logits tolerance of the fixture - LOGIT_TOL for the benign synthetic weights; the realistic-statistics
    fixtures pass twice their relative bound x the logits' standard deviation.)"""
    x = np.asarray(filtered, dtype=np.float64)
    if temperature == 0.0:
        order = np.argsort(-x, kind="stable")
        return bool(x[order[0]] - x[order[1]] < logit_tol and got == int(order[1]) and want == int(order[0]))
    t = float(np.float16(temperature))
    x = x * float(np.float32(1.0) / np.float32(t))
    tol = logit_tol / t
    order = np.argsort(-x, kind="stable")[: top_k + 1]
    v = x[order]
    if np.any(v[:-1] - v[1:] < tol):          # candidate set or candidate order can change
        return True
    m = x.max()
    lse = m + np.log(np.exp(x - m).sum())
    probs = np.exp(v[:top_k] - lse)
    rnd = OD.uniform01(seed, counter) * probs.sum()
    edges = np.cumsum(probs)[:-1]
    return bool(len(edges) and np.min(np.abs(edges - rnd)) < 2.0 * tol * probs.sum())   # the draw sits on an interval boundary


def assert_tokens_or_proven_near_tie(got_tokens, want_tokens, record, start=0, temperature=0.0, seed=0, top_k=5):
    """`record` = decode_text(record_logits=...) of the oracle run that produced `want_tokens`; result token k is
    currentTokens[start + k], sampled by decode step start + k - 1.  Returns the number of leading tokens that are
    comparable (all of them when the sequences agree; k when a proven near-tie at k ends the comparison)."""
    got_tokens, want_tokens = list(got_tokens), list(want_tokens)
    if got_tokens == want_tokens:
        return len(want_tokens)
    k = next((i for i, (a, b) in enumerate(zip(got_tokens, want_tokens)) if a != b), min(len(got_tokens), len(want_tokens)))
    assert k < min(len(got_tokens), len(want_tokens)), f"token lists differ in length only: {len(got_tokens)} vs {len(want_tokens)}"
    step = start + k - 1
    token_index, _, _, filtered = record[step]
    ok = _explain(filtered, got_tokens[k], want_tokens[k], temperature, seed, token_index, top_k)
    fin = np.sort(np.asarray(filtered)[np.isfinite(filtered)])[-2:]
    assert ok, (f"token mismatch at result index {k} (decode step {step}): got {got_tokens[k]}, oracle {want_tokens[k]}, "
                f"oracle top-2 gap {float(fin[1] - fin[0]):.3e}, T={temperature}: not a near-tie")
    return k
