"""Synthetic 30 s / 16 kHz PCM chunks: 0.05 N(0,1) noise + three sinusoids, clipped to [-1, 1].

Used by bench.py and the parity tests so both sides see byte-identical inputs; there are no
datasets in the image (no network).
"""
import numpy as np

WINDOW_SAMPLES = 480000
SAMPLE_RATE = 16000


def synthetic_chunk(seed: int, n: int = WINDOW_SAMPLES) -> np.ndarray:
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / SAMPLE_RATE
    x = 0.05 * rng.standard_normal(n)
    for f in (220.0, 440.0, 1760.0):
        x += 0.1 * np.sin(2 * np.pi * f * t + rng.uniform(0, 2 * np.pi))
    # slow amplitude envelope so that the spectrogram is not stationary
    x *= 0.6 + 0.4 * np.sin(2 * np.pi * 0.37 * t + rng.uniform(0, 2 * np.pi))
    return np.clip(x, -1.0, 1.0).astype(np.float32)
