"""Whisper-style log-mel front end of the transcript lane, on the GPU.

``log_mel_spectrogram(audio, n_mels=80, padding=0)`` has the call shape of
``whisper.audio.log_mel_spectrogram`` as the reference probes it
(``ser/_internal/transcript/backends/stable_whisper_mps_compat.py:192-201``):
16 kHz mono float32 in, ``[n_mels, n_samples // 160]`` float32 out (a leading
batch axis is kept when the input has one).  Runs ``serhip_whisper_logmel``.

Batch semantics: every clip of a ``[b, n]`` input is processed INDEPENDENTLY -- the ``max - 8`` clamp uses
the clip's own maximum, which is what Hugging Face's ``WhisperFeatureExtractor`` does per waveform and what
the reference's call site gets (it passes one clip).  ``whisper.audio.log_mel_spectrogram`` itself, handed a
batched tensor, would take ``log_spec.max()`` over the WHOLE batch, so there every clip but the loudest
would be clamped differently; call this function clip by clip if that is what you need to reproduce.
(``tests/test_gpu_parity.py::test_config5_full_length_clips_vs_hf_extractor`` pins the per-clip form with a
clip scaled by 1e-3 inside a batch.)
"""

from __future__ import annotations

import numpy as np

from .engine import get_engine

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160


def log_mel_spectrogram(audio, n_mels: int = 80, padding: int = 0, *, as_numpy: bool = True):
    """Log-mel spectrogram of one clip ``[n]`` or a batch ``[b, n]`` ."""
    if n_mels not in (80, 128):
        raise ValueError(f"Unsupported n_mels: {n_mels}")
    engine = get_engine()
    t = engine.to_device(audio)
    batched = t.dim() == 2
    if padding > 0:
        import torch.nn.functional as F

        t = F.pad(t, (0, int(padding)))
    out = engine.whisper_logmel(t, n_mels=n_mels)
    if not batched:
        out = out[0]
    return out.cpu().numpy() if as_numpy else out


def frames_for(n_samples: int) -> int:
    return int(n_samples) // HOP_LENGTH


__all__ = ["log_mel_spectrogram", "frames_for", "SAMPLE_RATE", "N_FFT", "HOP_LENGTH"]
_ = np  # numpy kept importable for callers that pass arrays
