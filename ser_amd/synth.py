"""Deterministic synthetic inputs named by BASELINE.md / SURVEY.md section 8d.

Used by ``bench.py``, the tests and ``__graft_entry__.smoke()``; nothing here
depends on the oracle or on the GPU.
"""

from __future__ import annotations

import math

import numpy as np


def peak_normalise(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    peak = float(np.max(np.abs(x))) if x.size else 0.0
    return x if peak == 0.0 else (x / np.float32(peak)).astype(np.float32)


def harmonic_clip(rng: np.random.Generator, n_samples: int, sample_rate: int = 16000) -> np.ndarray:
    """Config-3/4 generator: 3-6 harmonics of f0 in U[80, 400] Hz with an AM envelope plus
    white noise (sigma 0.01), peak-normalised float32."""
    t = np.arange(n_samples, dtype=np.float64) / float(sample_rate)
    f0 = rng.uniform(80.0, 400.0)
    n_h = int(rng.integers(3, 7))
    x = np.zeros(n_samples, dtype=np.float64)
    for h in range(1, n_h + 1):
        x += (rng.uniform(0.3, 1.0) / h) * np.sin(2.0 * np.pi * f0 * h * t + rng.uniform(0, 2 * np.pi))
    env = 0.6 + 0.4 * np.sin(2.0 * np.pi * rng.uniform(0.5, 4.0) * t + rng.uniform(0, 2 * np.pi))
    x = x * env + 0.01 * rng.standard_normal(n_samples)
    return peak_normalise(x.astype(np.float32))


def harmonic_clips(seed: int, n_clips: int, n_samples: int, sample_rate: int = 16000) -> np.ndarray:
    """``[n_clips, n_samples]`` float32 of independent :func:`harmonic_clip` rows."""
    rng = np.random.default_rng(seed)
    return np.stack([harmonic_clip(rng, n_samples, sample_rate) for _ in range(n_clips)])


def sample_wav_like(seed: int = 7, n_samples: int = 69937, sample_rate: int = 16000) -> np.ndarray:
    """Config-1 stand-in for the reference's ``sample.wav`` (mono 16 kHz, 69 937 samples)."""
    return harmonic_clip(np.random.default_rng(seed), n_samples, sample_rate)


def ravdess_sine_clip(actor_id: int, emotion_code: int, *, sample_rate: int = 16000,
                      duration_seconds: float = 1.5, amplitude: float = 0.15) -> np.ndarray:
    """Config-2 clip: the PCM16 sine wave of ``scripts/build_synthetic_ravdess_dataset.py:76-141``
    as decoded float32 (int16 / 32768), not yet peak-normalised."""
    total = int(round(sample_rate * duration_seconds))
    freq = 180.0 + float(emotion_code) * 22.0 + float(actor_id) * 7.0
    idx = np.arange(total, dtype=np.float64)
    value = amplitude * np.sin(2.0 * math.pi * freq * (idx / float(sample_rate)))
    pcm = np.rint(np.clip(value, -1.0, 1.0) * 32767).astype(np.int16)
    return pcm.astype(np.float32) / np.float32(32768.0)


def write_wav_pcm16(path, samples_int16: np.ndarray, sample_rate: int) -> None:
    import wave

    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(np.asarray(samples_int16, dtype="<i2").tobytes())


def harmonic_clips_torch(seed: int, n_clips: int, n_samples: int, device, sample_rate: int = 16000):
    """Same family as :func:`harmonic_clips`, generated on ``device`` with torch (bench inputs:
    nothing is read from disk and nothing crosses PCIe).  Returns ``[n_clips, n_samples]`` float32."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    out = torch.empty((n_clips, n_samples), dtype=torch.float32, device=device)
    t = torch.arange(n_samples, dtype=torch.float32, device=device) / float(sample_rate)
    step = 512
    for lo in range(0, n_clips, step):
        n = min(step, n_clips - lo)

        def u(a, b, *shape):
            return a + (b - a) * torch.rand((n, *shape), generator=g, device=device)

        f0 = u(80.0, 400.0, 1)
        n_h = torch.randint(3, 7, (n, 1), generator=g, device=device)
        x = torch.zeros((n, n_samples), dtype=torch.float32, device=device)
        for h in range(1, 7):
            amp = u(0.3, 1.0, 1) / h * (n_h >= h)
            x += amp * torch.sin(2.0 * torch.pi * f0 * h * t[None, :] + u(0.0, 2 * torch.pi, 1))
        env = 0.6 + 0.4 * torch.sin(2.0 * torch.pi * u(0.5, 4.0, 1) * t[None, :] + u(0.0, 2 * torch.pi, 1))
        x = x * env + 0.01 * torch.randn((n, n_samples), generator=g, device=device)
        out[lo : lo + n] = x / x.abs().amax(dim=1, keepdim=True)
    return out
