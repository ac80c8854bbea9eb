"""Boundary data types of the fast feature path.

Same names, fields, validation and error strings as the reference's
``FeatureFlags`` / ``AudioReadConfig`` (ser/_internal/config/schema.py:219-248),
``PoolingWindow`` / ``EncodedSequence`` / ``FeatureBackend`` protocols
(ser/_internal/repr/backend.py:19-155) and ``FeatureFrame``
(ser/_internal/features/feature_extractor.py:26-32), so callers can swap the
package in.  Plain picklable dataclasses: no GPU state lives here.
"""

from __future__ import annotations

import os
from collections.abc import Sequence
from dataclasses import dataclass
from typing import Protocol, runtime_checkable

import numpy as np
from numpy.typing import NDArray

EmbeddingMatrix = NDArray[np.float32]
TimeVector = NDArray[np.float64]
FeatureMatrix = NDArray[np.float64]
FeatureVector = NDArray[np.float64]

#: width of each feature block, in output order (dsp.py:106-151)
BLOCK_WIDTHS = (("mfcc", 40), ("chroma", 12), ("mel", 128), ("contrast", 7), ("tonnetz", 6))


@dataclass(frozen=True)
class FeatureFlags:
    """Feature extraction toggles (reference default: all on, 193 dims).

    ``tonnetz`` (HPSS + constant-Q chroma, 6 dims) is outside the accelerated
    path: enabling it makes extraction raise ``NotImplementedError``.
    """

    mfcc: bool = True
    chroma: bool = True
    mel: bool = True
    contrast: bool = True
    tonnetz: bool = True

    def width(self) -> int:
        return sum(w for name, w in BLOCK_WIDTHS if getattr(self, name))

    def native_bits(self) -> int:
        return (1 if self.mfcc else 0) | (2 if self.chroma else 0) | (4 if self.mel else 0) | (8 if self.contrast else 0)


#: the 187-dim configuration this package accelerates
FAST_PATH_FLAGS = FeatureFlags(tonnetz=False)

#: environment opt-in that makes ``feature_flags=None`` / ``settings=None`` mean the 187-dim flags
DEFAULT_FLAGS_ENV = "SERHIP_DEFAULT_FLAGS"

TONNETZ_MESSAGE = (
    "tonnetz features (HPSS + constant-Q chroma, 6 of the reference's default 193 dims) are outside the "
    "accelerated path: pass FeatureFlags(tonnetz=False) (187 dims) explicitly, or set "
    f"{DEFAULT_FLAGS_ENV}=fast187 to make that the meaning of feature_flags=None / settings=None"
)


def default_feature_flags() -> FeatureFlags:
    """What ``feature_flags=None`` means.  The reference's default is ``FeatureFlags()`` with tonnetz on
    (193 dims; dsp.py:92, repr/handcrafted.py:38, config/schema.py:219-227), and so is this package's:
    extraction then raises ``NotImplementedError`` naming the way out, instead of silently returning 187
    values to a caller whose artifact expects 193.  ``SERHIP_DEFAULT_FLAGS=fast187`` is the one explicit
    opt-in to the accelerated 187-dim configuration as the default."""
    choice = os.environ.get(DEFAULT_FLAGS_ENV, "").strip().lower()
    if choice == "fast187":
        return FAST_PATH_FLAGS
    if choice not in ("", "reference", "193"):
        raise ValueError(f"{DEFAULT_FLAGS_ENV} must be 'fast187' or 'reference', not {choice!r}")
    return FeatureFlags()


def coerce_feature_flags(flags) -> FeatureFlags:
    """``None`` -> :func:`default_feature_flags`; any object with the five boolean attributes (the
    reference's own ``FeatureFlags`` dataclass qualifies) -> this package's ``FeatureFlags``."""
    if flags is None:
        return default_feature_flags()
    if isinstance(flags, FeatureFlags):
        return flags
    return FeatureFlags(*(bool(getattr(flags, name)) for name, _ in BLOCK_WIDTHS))


def require_accelerated(flags: FeatureFlags) -> None:
    """Raise the loud tonnetz error before any work is done."""
    if flags.tonnetz:
        raise NotImplementedError(TONNETZ_MESSAGE)


@dataclass(frozen=True)
class AudioReadConfig:
    """How often and how patiently ``read_audio_file`` retries a read that failed for a possibly transient reason
    (fields and defaults of config/schema.py:243-248)."""

    max_retries: int = 3
    retry_delay_seconds: float = 1.0


@dataclass(frozen=True)
class PoolingWindow:
    """One ``[start_seconds, end_seconds)`` span over which frame embeddings are pooled (repr/backend.py:19-35:
    finite, start >= 0, end > start)."""

    start_seconds: float
    end_seconds: float

    def __post_init__(self) -> None:
        lo, hi = self.start_seconds, self.end_seconds
        if not (np.isfinite(lo) and np.isfinite(hi)):
            raise ValueError("PoolingWindow bounds must be finite numbers.")
        if lo < 0.0:
            raise ValueError("PoolingWindow start_seconds must be non-negative.")
        if hi <= lo:
            raise ValueError("PoolingWindow end_seconds must be greater than start_seconds.")


@dataclass(frozen=True)
class EncodedSequence:
    """What ``encode_sequence`` returns: a ``[frames, D]`` embedding matrix plus the start and end time of every
    frame and the id of the backend that produced it (validation rules of repr/backend.py:37-78)."""

    embeddings: EmbeddingMatrix
    frame_start_seconds: TimeVector
    frame_end_seconds: TimeVector
    backend_id: str

    def __post_init__(self) -> None:
        emb, t0, t1 = self.embeddings, self.frame_start_seconds, self.frame_end_seconds
        if not self.backend_id:
            raise ValueError("EncodedSequence backend_id must be a non-empty string.")
        if emb.ndim != 2:
            raise ValueError("EncodedSequence embeddings must be 2D (frames, features).")
        if t0.ndim != 1 or t1.ndim != 1:
            raise ValueError("Frame timestamp arrays must be 1D.")
        n_frames = int(emb.shape[0])
        if n_frames <= 0:
            raise ValueError("EncodedSequence must contain at least one frame.")
        for label, arr in (("frame_start_seconds", t0), ("frame_end_seconds", t1)):
            if arr.size != n_frames:
                raise ValueError(f"{label} length must match embeddings frame count.")
        for label, arr in (("embeddings", emb), ("frame_start_seconds", t0), ("frame_end_seconds", t1)):
            if not np.all(np.isfinite(arr)):
                raise ValueError(f"EncodedSequence {label} contain non-finite values.")
        for label, arr in (("frame_start_seconds", t0), ("frame_end_seconds", t1)):
            if np.any(np.diff(arr) < 0.0):
                raise ValueError(f"{label} must be non-decreasing.")
        if np.any(t1 <= t0):
            raise ValueError("Each frame must satisfy end_seconds > start_seconds.")


@dataclass(frozen=True)
class FeatureFrame:
    """One analysis window of a file with its float64 feature vector (features/feature_extractor.py:26-32)."""

    start_seconds: float
    end_seconds: float
    features: FeatureVector


def overlap_frame_range(encoded: EncodedSequence, window: PoolingWindow) -> tuple[int, int]:
    """Half-open frame range ``[lo, hi)`` that overlaps ``window``.

    Frame times are non-decreasing (``EncodedSequence`` enforces it), so the
    frames with ``end > window.start and start < window.end`` are contiguous;
    the range is what the GPU pooling kernel consumes.  Raises exactly like
    ``overlap_frame_mask`` (repr/backend.py:94-110).
    """
    t0, t1 = encoded.frame_start_seconds, encoded.frame_end_seconds
    first, last = float(t0[0]), float(t1[-1])
    if window.start_seconds < first or window.end_seconds > last:
        raise ValueError(
            "Pooling window is outside encoded sequence range: "
            f"[{window.start_seconds}, {window.end_seconds}] vs [{first}, {last}]"
        )
    lo = int(np.searchsorted(t1, window.start_seconds, side="right"))   # first frame with end > start
    hi = int(np.searchsorted(t0, window.end_seconds, side="left"))      # first frame with start >= end
    if hi <= lo:
        raise ValueError(
            f"Pooling window does not overlap any encoded frames: [{window.start_seconds}, {window.end_seconds}]"
        )
    return lo, hi


def overlap_frame_mask(encoded: EncodedSequence, window: PoolingWindow) -> NDArray[np.bool_]:
    """Boolean mask form of :func:`overlap_frame_range` (reference signature)."""
    lo, hi = overlap_frame_range(encoded, window)
    mask = np.zeros(encoded.frame_start_seconds.shape[0], dtype=bool)
    mask[lo:hi] = True
    return mask


@runtime_checkable
class FeatureBackend(Protocol):
    """Structural type every representation backend satisfies: ``backend_id``, ``feature_dim``,
    ``encode_sequence`` and ``pool`` (repr/backend.py:114-143; runtime-checkable like the reference's)."""

    @property
    def backend_id(self) -> str: ...

    @property
    def feature_dim(self) -> int: ...

    def encode_sequence(self, audio: NDArray[np.float32], sample_rate: int) -> EncodedSequence: ...

    def pool(self, encoded: EncodedSequence, windows: Sequence[PoolingWindow]) -> FeatureMatrix: ...


@runtime_checkable
class VectorFeatureBackend(FeatureBackend, Protocol):
    """Backends that can also turn a whole clip into ONE vector (``extract_vector``, repr/backend.py:146-155)."""

    def extract_vector(self, audio: NDArray[np.float32], sample_rate: int) -> FeatureVector: ...
