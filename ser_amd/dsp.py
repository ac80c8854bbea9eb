"""Signal-level feature extraction with the reference's call signature.

``extract_feature_from_signal(audio, sample_rate, *, feature_flags=None)`` mirrors
``ser/_internal/utils/dsp.py:67-151``: same validation order and messages, same
output order ``[mfcc40 | chroma12 | mel128 | contrast7]`` and float64 result;
the arithmetic runs in the gfx950 kernels behind ``serhip_fast_features``.
"""

from __future__ import annotations

import logging

import numpy as np
from numpy.typing import NDArray

from .engine import get_engine
from .types import FeatureFlags, FeatureVector, coerce_feature_flags, require_accelerated

logger = logging.getLogger(__name__)

#: librosa raises ParameterError when 200*2**k octave band edges reach Nyquist
CONTRAST_MIN_SAMPLE_RATE = 12800


class ParameterError(Exception):
    """Counterpart of ``librosa.util.exceptions.ParameterError`` raised through dsp.py:129."""


def samples_all_finite(prepared: NDArray[np.float32]) -> bool:
    """A finite sum of squares proves every sample finite (NaN and inf propagate) in one BLAS pass without the
    boolean temporary of np.isfinite; an overflowing one falls through to the exact test."""
    return bool(np.isfinite(np.dot(prepared, prepared))) or bool(np.all(np.isfinite(prepared)))


def validate_signal(audio: NDArray[np.float32], sample_rate: int, *, finite_check: bool = True) -> NDArray[np.float32]:
    """Input checks of dsp.py:85-95, in the reference's order; returns a float32 view/copy.  ``finite_check=False``
    leaves the last of them (every sample finite) to the caller, who passes ``check_finite=True`` to
    ``FeatureEngine.features``: the staging threads then test the samples in the pass that copies them."""
    if sample_rate <= 0:
        raise ValueError("Sample rate must be a positive integer.")
    if audio.ndim != 1:
        raise ValueError("Audio must be mono (1D array).")
    if audio.size == 0:
        raise ValueError("Audio contains no samples.")
    prepared = np.asarray(audio, dtype=np.float32)
    if finite_check and not samples_all_finite(prepared):
        raise ValueError("Audio buffer is not finite everywhere.")
    return prepared


def check_flags_for_rate(flags: FeatureFlags, sample_rate: int) -> None:
    """Errors the reference's librosa calls raise for a (flags, sample_rate) pair."""
    if flags.contrast and sample_rate <= CONTRAST_MIN_SAMPLE_RATE:
        raise ParameterError("Frequency band exceeds Nyquist. Reduce either fmin or n_bands.")


def extract_feature_from_signal(
    audio: NDArray[np.float32],
    sample_rate: int,
    *,
    feature_flags: FeatureFlags | None = None,
) -> FeatureVector:
    """Extract the configured features from an in-memory mono signal (float64 vector).

    ``feature_flags=None`` means what it means in the reference: ``FeatureFlags()`` with tonnetz on
    (193 dims, dsp.py:92).  Tonnetz is not implemented here, so that default raises
    ``NotImplementedError`` (after the input validation, like any other extraction failure) unless
    ``SERHIP_DEFAULT_FLAGS=fast187`` is set; pass ``FeatureFlags(tonnetz=False)`` for the 187-dim vector.
    """
    flags = coerce_feature_flags(feature_flags)
    prepared = validate_signal(audio, sample_rate)
    try:
        require_accelerated(flags)
        if flags.width() == 0:
            return np.empty(0, dtype=np.float64)
        check_flags_for_rate(flags, sample_rate)
        engine = get_engine()
        out = engine.features(prepared, [0], [prepared.size], sample_rate, flags)
        vector = out[0].cpu().numpy()
    except Exception as err:
        logger.warning("Error extracting features from signal: %s", err)
        raise
    return vector.astype(np.float64, copy=False)
