"""CPU restatement of ``whisper.audio.log_mel_spectrogram`` (config 5).  TEST INFRASTRUCTURE ONLY.

openai-whisper 20250625 (reference ``uv.lock:1448-1449``) is not installed; the
reference reaches it at
``ser/_internal/transcript/backends/stable_whisper_mps_compat.py:192-201``
(``log_mel_spectrogram(audio, 80, 0)``).  Its published algorithm:
``torch.stft(audio, 400, 160, window=hann_window(400), return_complex=True)``
(center=True, reflect padding), drop the last column, ``|.|^2``, Slaney mel
filterbank (``librosa.filters.mel(sr=16000, n_fft=400, n_mels=80)``),
``log10(clamp(min=1e-10))``, ``maximum(x, x.max() - 8.0)``, ``(x + 4) / 4``.

Pinned in ``tests/test_oracle_pins.py`` against the HF
``WhisperFeatureExtractor`` numpy path (an independent implementation).
"""

from __future__ import annotations

import numpy as np
import scipy.fft

from .librosa_restated import mel_filter

SAMPLE_RATE, N_FFT, HOP = 16000, 400, 160


def log_mel_spectrogram(audio: np.ndarray, n_mels: int = 80, padding: int = 0) -> np.ndarray:
    """float32 ``[n_mels, n_samples // 160]``; arithmetic in float64, rounded at the end."""
    x = np.asarray(audio, dtype=np.float32)
    if padding > 0:
        x = np.pad(x, (0, padding))
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(N_FFT) / N_FFT)      # periodic Hann
    padded = np.pad(x.astype(np.float64), (N_FFT // 2, N_FFT // 2), mode="reflect")
    frames = np.lib.stride_tricks.sliding_window_view(padded, N_FFT)[::HOP]
    spec = scipy.fft.rfft(frames * window[None, :], axis=-1).T                # [201, 1 + n//160]
    magnitudes = np.abs(spec[:, :-1]) ** 2
    filters = mel_filter(sr=SAMPLE_RATE, n_fft=N_FFT, n_mels=n_mels).astype(np.float64)
    mel_spec = filters @ magnitudes
    log_spec = np.log10(np.maximum(mel_spec, 1e-10))
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    log_spec = (log_spec + 4.0) / 4.0
    return log_spec.astype(np.float32)
