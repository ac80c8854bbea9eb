"""CPU restatement of the reference's own glue on the fast-profile feature path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Each function cites
the reference file:line it follows (paths relative to ``/root/reference``).
Arithmetic is delegated to ``oracle.librosa_restated`` exactly where the
reference delegates to librosa.
"""

from __future__ import annotations

import math
import warnings
from dataclasses import dataclass

import numpy as np

from . import librosa_restated as lr

FEATURE_BLOCKS = (("mfcc", 40), ("chroma", 12), ("mel", 128), ("contrast", 7), ("tonnetz", 6))


@dataclass(frozen=True)
class Flags:
    """``FeatureFlags`` (ser/_internal/config/schema.py:219-227); tonnetz is out of scope."""

    mfcc: bool = True
    chroma: bool = True
    mel: bool = True
    contrast: bool = True
    tonnetz: bool = False


def feature_dim(flags: Flags) -> int:
    """``HandcraftedBackend.feature_dim`` (repr/handcrafted.py:45-59)."""
    return sum(width for name, width in FEATURE_BLOCKS if getattr(flags, name))


def pad_audio_for_fft(audio: np.ndarray, minimum_window: int = 512) -> np.ndarray:
    """``_pad_audio_for_fft`` (utils/dsp.py:38-45)."""
    if audio.size >= minimum_window:
        return audio
    return np.pad(audio, (0, minimum_window - audio.size), mode="constant")


def extract_feature_from_signal(audio: np.ndarray, sample_rate: int, *, flags: Flags | None = None,
                                return_debug: bool = False):
    """``extract_feature_from_signal`` (utils/dsp.py:67-151), tonnetz excluded.

    Three STFTs per call exactly like the reference (explicit, inside mfcc,
    inside melspectrogram); filterbanks rebuilt per call.
    """
    if sample_rate <= 0:
        raise ValueError("Sample rate must be a positive integer.")
    if audio.ndim != 1:
        raise ValueError("Audio must be mono (1D array).")
    if audio.size == 0:
        raise ValueError("Audio contains no samples.")
    flags = flags if flags is not None else Flags()
    if flags.tonnetz:
        raise NotImplementedError("tonnetz is outside the oracle's scope (SURVEY.md 8a row a9)")
    prepared = pad_audio_for_fft(np.asarray(audio, dtype=np.float32))
    if not bool(np.all(np.isfinite(prepared))):
        raise ValueError("Audio buffer is not finite everywhere.")
    n_fft = min(prepared.size, 2048)
    parts: list[np.ndarray] = []
    debug: dict = {"n_fft": n_fft}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        stft_magnitude = np.abs(lr.stft(prepared, n_fft=n_fft))                 # dsp.py:100
        stft_power_db = lr.power_to_db(np.square(stft_magnitude), ref=np.max)   # dsp.py:101-104
        if flags.mfcc:                                                          # dsp.py:106-111
            m = lr.mfcc(prepared, sr=sample_rate, n_mfcc=40, n_fft=n_fft)
            parts.append(np.asarray(np.mean(m, axis=1), dtype=np.float64))
        if flags.chroma:                                                        # dsp.py:113-118
            c, tuning = lr.chroma_stft(stft_magnitude, sr=sample_rate, n_fft=n_fft,
                                       return_tuning=True)
            debug["tuning"] = tuning
            parts.append(np.asarray(np.mean(c, axis=1), dtype=np.float64))
        if flags.mel:                                                           # dsp.py:120-125
            ms = lr.melspectrogram(prepared, sr=sample_rate, n_fft=n_fft)
            parts.append(np.asarray(np.mean(ms, axis=1), dtype=np.float64))
        if flags.contrast:                                                      # dsp.py:127-136
            sc = lr.spectral_contrast(stft_power_db, sr=sample_rate, n_fft=n_fft)
            parts.append(np.asarray(np.mean(sc, axis=1), dtype=np.float64))
    if not parts:
        out = np.empty(0, dtype=np.float64)
    else:
        out = np.concatenate(parts).astype(np.float64, copy=False)
    if return_debug:
        return out, debug
    return out


def window_bounds(n_samples: int, sample_rate: int, frame_size_seconds, frame_stride_seconds):
    """Window (start, end) sample indices of ``encode_sequence`` (repr/handcrafted.py:78-97)."""
    frame_length = max(1, int(round(frame_size_seconds * sample_rate)))
    frame_step = max(1, int(round(frame_stride_seconds * sample_rate)))
    return [(s, min(s + frame_length, n_samples)) for s in range(0, n_samples, frame_step)]


def encode_sequence(audio: np.ndarray, sample_rate: int, *, frame_size_seconds=3,
                    frame_stride_seconds=1, flags: Flags | None = None):
    """``HandcraftedBackend.encode_sequence`` (repr/handcrafted.py:65-107).

    Returns (embeddings f32 [W, D], starts f64 [W], ends f64 [W]).
    """
    if sample_rate <= 0:
        raise ValueError("sample_rate must be a positive integer.")
    if audio.ndim != 1:
        raise ValueError("audio must be mono (1D array).")
    if audio.size == 0:
        raise ValueError("audio must contain at least one sample.")
    rows, starts, ends = [], [], []
    for s, e in window_bounds(audio.size, sample_rate, frame_size_seconds, frame_stride_seconds):
        frame = audio[s:e]
        if frame.size == 0:
            continue
        vec = extract_feature_from_signal(frame, sample_rate, flags=flags)
        rows.append(np.asarray(vec, dtype=np.float32))
        starts.append(float(s) / float(sample_rate))
        ends.append(float(e) / float(sample_rate))
    return (np.vstack(rows).astype(np.float32, copy=False),
            np.asarray(starts, dtype=np.float64), np.asarray(ends, dtype=np.float64))


def extract_vector(audio: np.ndarray, sample_rate: int, *, flags: Flags | None = None):
    """``HandcraftedBackend.extract_vector`` (repr/handcrafted.py:124-137)."""
    return np.asarray(extract_feature_from_signal(audio, sample_rate, flags=flags), dtype=np.float64)


def overlap_frame_mask(starts: np.ndarray, ends: np.ndarray, w_start: float, w_end: float):
    """``overlap_frame_mask`` (repr/backend.py:81-111)."""
    min_start, max_end = float(starts[0]), float(ends[-1])
    if w_start < min_start or w_end > max_end:
        raise ValueError(
            "Pooling window is outside encoded sequence range: "
            f"[{w_start}, {w_end}] vs [{min_start}, {max_end}]"
        )
    mask = np.logical_and(ends > w_start, starts < w_end)
    if not np.any(mask):
        raise ValueError(f"Pooling window does not overlap any encoded frames: [{w_start}, {w_end}]")
    return mask


def mean_pool(emb: np.ndarray, starts, ends, windows):
    """``HandcraftedBackend.pool`` (repr/handcrafted.py:109-122)."""
    if not windows:
        return np.empty((0, emb.shape[1]), dtype=np.float64)
    rows = [np.asarray(emb[overlap_frame_mask(starts, ends, a, b)].mean(axis=0), dtype=np.float64)
            for a, b in windows]
    return np.vstack(rows)


def mean_std_pool(emb: np.ndarray, starts, ends, windows):
    """``mean_std_pool`` (pool/stats_pool.py:15-43): [mean | population std] in float64."""
    dim = int(emb.shape[1])
    if not windows:
        return np.empty((0, dim * 2), dtype=np.float64)
    rows = []
    for a, b in windows:
        sel = np.asarray(emb[overlap_frame_mask(starts, ends, a, b)], dtype=np.float64)
        rows.append(np.concatenate((sel.mean(axis=0), sel.std(axis=0)), axis=0))
    return np.vstack(rows).astype(np.float64, copy=False)


def temporal_pooling_windows(starts, ends, *, window_size_seconds: float,
                             window_stride_seconds: float):
    """``temporal_pooling_windows`` (pool/windowing.py:10-71) -> list of (start, end)."""
    if window_size_seconds <= 0.0 or not np.isfinite(window_size_seconds):
        raise ValueError("window_size_seconds must be a positive finite float.")
    if window_stride_seconds <= 0.0 or not np.isfinite(window_stride_seconds):
        raise ValueError("window_stride_seconds must be a positive finite float.")
    clip_start, clip_end = float(starts[0]), float(ends[-1])
    clip_duration = clip_end - clip_start
    if clip_duration <= 0.0:
        raise ValueError("Encoded sequence duration must be positive.")
    effective = min(window_size_seconds, clip_duration)
    if np.isclose(effective, clip_duration):
        return [(clip_start, clip_end)]
    windows, eps, cursor = [], 1e-9, clip_start
    while cursor + effective <= clip_end + eps:
        windows.append((cursor, min(clip_end, cursor + effective)))
        cursor += window_stride_seconds
    if not windows:
        return [(max(clip_start, clip_end - effective), clip_end)]
    if windows[-1][1] < clip_end - eps:
        tail = (max(clip_start, clip_end - effective), clip_end)
        prev = windows[-1]
        if not (np.isclose(prev[0], tail[0]) and np.isclose(prev[1], tail[1])):
            windows.append(tail)
    return windows


def prepare_audio_buffer(raw: np.ndarray) -> np.ndarray:
    """``_prepare_audio_buffer`` (utils/audio_utils.py:28-60): f32, NaN/inf->0, mono, peak-normalise."""
    prepared = np.asarray(raw, dtype=np.float32)
    prepared = np.nan_to_num(prepared, copy=True, nan=0.0, posinf=0.0, neginf=0.0)
    if prepared.ndim == 2:
        if prepared.shape[1] == 0:
            prepared = np.array([], dtype=np.float32)
        else:
            prepared = np.asarray(np.mean(prepared, axis=1, dtype=np.float32), dtype=np.float32)
    elif prepared.ndim != 1:
        raise OSError(f"Unsupported audio shape: {prepared.shape}")
    if prepared.size == 0:
        raise OSError("Audio file contains no samples.")
    max_abs = float(np.max(np.abs(prepared)))
    if max_abs == 0:
        return np.zeros_like(prepared)
    return prepared / max_abs


def librosa_load_mono(frames: np.ndarray) -> np.ndarray:
    """The mix-down inside ``librosa.load(path, sr=None)`` as ``read_audio_file`` calls it
    (utils/audio_utils.py:104-109; librosa 0.11 ``core/audio.py``: ``y = sf.read(dtype=float32).T`` then
    ``to_mono(y) = np.mean(y, axis=tuple(range(y.ndim - 1)))``).  ``frames`` is what soundfile returns:
    ``[n]`` or ``[n, channels]`` float32.  No scrub happens here: a NaN in one channel makes the frame NaN."""
    y = np.asarray(frames, dtype=np.float32).T
    if y.ndim > 1:
        with np.errstate(invalid="ignore", over="ignore"):
            y = np.mean(y, axis=tuple(range(y.ndim - 1)))
    return np.asarray(y, dtype=np.float32)


def read_audio_prepared(frames: np.ndarray) -> np.ndarray:
    """Primary path of ``read_audio_file`` after decoding (utils/audio_utils.py:104-113):
    ``librosa.load`` (mono mix-down) -> ``_prepare_audio_buffer`` (scrub, peak-normalise)."""
    return prepare_audio_buffer(librosa_load_mono(frames))


def synthetic_ravdess_clip(actor_id: int, emotion_code: int, *, sample_rate: int = 16000,
                           duration_seconds: float = 1.5, amplitude: float = 0.15) -> np.ndarray:
    """PCM16 sine clip of ``scripts/build_synthetic_ravdess_dataset.py:76-141`` as decoded audio.

    Returns the int16 samples scaled by 1/32768 (what a PCM16 WAV decodes to),
    before ``prepare_audio_buffer``.
    """
    total = int(round(sample_rate * duration_seconds))
    freq = 180.0 + float(emotion_code) * 22.0 + float(actor_id) * 7.0
    pcm = np.empty(total, dtype=np.int16)
    for i in range(total):
        value = amplitude * math.sin(2.0 * math.pi * freq * (i / float(sample_rate)))
        pcm[i] = int(round(max(-1.0, min(1.0, value)) * 32767))
    return pcm.astype(np.float32) / np.float32(32768.0)
