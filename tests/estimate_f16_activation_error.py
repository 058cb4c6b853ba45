"""Offline estimate (CPU oracle; not collected by pytest): logits error when the decoder projections round their activations to f16
(an MFMA operand) instead of keeping them f32 as the GEMV kernels do, and with an f16 hi | lo activation pair.
Last run: small (12 layers) f16 1.7e-3, hi|lo 1.6e-6; test-micro f16 3.7e-4, hi|lo 3.6e-7 (bar: 1e-3)."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from oracle import model as OM
from whisperkit_amd import weights
torch.set_num_threads(8)
for name in ("test-micro", "small"):
    dims = weights.MODEL_DIMS[name]
    sd = weights.synthetic_state_dict(dims, seed=0)
    om = OM.OracleWhisper(dims, sd)
    enc = (np.random.default_rng(0).standard_normal((1500, dims.n_audio_state)) * 0.5).astype(np.float32)
    toks = [dims.n_vocab - 1608, dims.n_vocab - 1502, 400, 370, 452, 7177]
    def run(mode):
        orig = F.linear
        def lin(x, w, b=None):
            if mode == "f16": x = x.half().float()
            elif mode == "hilo":
                hi = x.half().float(); lo = (x - hi).half().float(); x = hi + lo
            return orig(x, w, b)
        OM.F.linear = lin
        try:
            st = om.new_state(enc)
            return np.stack([st.step(t, i) for i, t in enumerate(toks)])
        finally:
            OM.F.linear = orig
    ref = run("f32")
    for mode in ("f16", "hilo"):
        got = run(mode)
        print(name, mode, "max |dlogit|", float(np.abs(got - ref).max()), "logit scale", float(np.abs(ref).max()))
