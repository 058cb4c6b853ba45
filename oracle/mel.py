"""Log-mel spectrogram oracle (numpy, float64 arithmetic) - test infrastructure only.

Follows openai/whisper `audio.py:log_mel_spectrogram` as shipped in transformers 5.15.0
`models/whisper/feature_extraction_whisper.py:95-167` (`_np_extract_fbank_features`) and
`audio_utils.py` (`mel_filter_bank(..., norm="slaney", mel_scale="slaney")`, `window_function("hann")`,
`spectrogram(center=True, pad_mode="reflect", power=2.0)`):

  1. reflect-pad 200 samples each side, frames of 400 at hop 160 -> 3001 frames, drop the last
  2. periodic Hann window, 400-point DFT, power spectrum over bins 0..200
  3. slaney-scale / slaney-normalised triangular mel filterbank (0..8000 Hz), 80 or 128 bands
  4. log10(max(x, 1e-10)); clamp to (global max - 8); (x + 4) / 4

Replaces the CoreML `MelSpectrogram` call of the reference
(Sources/WhisperKit/Core/FeatureExtractor.swift:40-56; padding: AudioProcessor.swift:151-174).
"""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000
N_FFT = 400
HOP = 160
N_SAMPLES = 480000
N_FRAMES = 3000
N_BINS = N_FFT // 2 + 1


def hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    min_log_hz, min_log_mel, logstep = 1000.0, 15.0, 27.0 / np.log(6.4)
    mels = 3.0 * f / 200.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log_region = f >= min_log_hz
        mels = np.where(log_region, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) * logstep, mels)
    return mels


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    min_log_hz, min_log_mel, logstep = 1000.0, 15.0, np.log(6.4) / 27.0
    f = 200.0 * m / 3.0
    log_region = m >= min_log_mel
    return np.where(log_region, min_log_hz * np.exp(logstep * (m - min_log_mel)), f)


def mel_filters(n_mels: int) -> np.ndarray:
    """[201, n_mels] float64 slaney filterbank (audio_utils.mel_filter_bank)."""
    fft_freqs = np.linspace(0, SAMPLE_RATE // 2, N_BINS)
    mel_pts = np.linspace(hz_to_mel_slaney(0.0), hz_to_mel_slaney(8000.0), n_mels + 2)
    filter_freqs = mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(filter_freqs)
    slopes = filter_freqs[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (filter_freqs[2: n_mels + 2] - filter_freqs[:n_mels])
    return fb * enorm[None, :]


def hann_periodic(n: int = N_FFT) -> np.ndarray:
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)


def pad_or_trim(pcm: np.ndarray, length: int = N_SAMPLES) -> np.ndarray:
    """AudioProcessor.padOrTrimAudio (AudioProcessor.swift:151-174): copy <= length samples, zero pad."""
    out = np.zeros(length, dtype=np.float32)
    n = min(len(pcm), length)
    out[:n] = np.asarray(pcm[:n], dtype=np.float32)
    return out


def log_mel_spectrogram(pcm: np.ndarray, n_mels: int = 80) -> np.ndarray:
    """pcm: <=480000 float32 samples -> [n_mels, 3000] float64 (callers cast)."""
    x = pad_or_trim(pcm).astype(np.float64)
    xp = np.pad(x, (N_FFT // 2, N_FFT // 2), mode="reflect")
    idx = np.arange(N_FFT)[None, :] + HOP * np.arange(N_FRAMES)[:, None]   # frames 0..2999 (3001st dropped)
    frames = xp[idx] * hann_periodic()[None, :]
    spec = np.fft.rfft(frames, n=N_FFT, axis=1)
    power = spec.real ** 2 + spec.imag ** 2                                 # [3000, 201]
    mel = power @ mel_filters(n_mels)                                       # [3000, n_mels]
    log_spec = np.log10(np.maximum(mel, 1e-10)).T                           # [n_mels, 3000]
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0
