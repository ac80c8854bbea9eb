"""File-level feature extraction with the reference's public signatures.

``extract_feature`` / ``extract_feature_frames`` / ``extended_extract_feature`` /
``extract_feature_from_signal`` follow
``ser/_internal/features/feature_extractor.py:106-179``; ``settings`` is any object
with ``feature_flags`` and ``audio_read`` attributes (the reference's ``AppConfig``
qualifies).  ``settings=None`` uses :data:`DEFAULT_SETTINGS` -- the 187-dim fast-path
flags -- instead of reloading an environment snapshot.
"""

from __future__ import annotations

import logging
from dataclasses import dataclass, field

import numpy as np
from numpy.typing import NDArray

from . import dsp
from .audio_io import read_audio_file
from .handcrafted import HandcraftedBackend
from .types import FAST_PATH_FLAGS, AudioReadConfig, FeatureFlags, FeatureFrame, FeatureVector

logger = logging.getLogger(__name__)


@dataclass(frozen=True)
class FeatureSettings:
    """The two settings the path reads from the reference's ``AppConfig``."""

    feature_flags: FeatureFlags = FAST_PATH_FLAGS
    audio_read: AudioReadConfig = field(default_factory=AudioReadConfig)


DEFAULT_SETTINGS = FeatureSettings()


def _resolve(settings) -> FeatureSettings:
    return settings if settings is not None else DEFAULT_SETTINGS


def extract_feature_from_signal(audio: NDArray[np.float32], sample_rate: int, *, settings=None) -> FeatureVector:
    """Settings-flavoured wrapper around :func:`ser_amd.dsp.extract_feature_from_signal`."""
    return dsp.extract_feature_from_signal(audio, sample_rate, feature_flags=_resolve(settings).feature_flags)


def extract_feature(file: str, *, settings=None) -> FeatureVector:
    """One feature vector for a whole audio file."""
    active = _resolve(settings)
    try:
        audio, sample_rate = read_audio_file(file, audio_read_config=active.audio_read)
    except Exception as err:
        logger.error("Error reading file %s: %s", file, err)
        raise
    return HandcraftedBackend(feature_flags=active.feature_flags).extract_vector(audio=audio, sample_rate=sample_rate)


def extract_feature_frames(audiofile: str, frame_size: int = 3, frame_stride: int = 1, *, settings=None
                           ) -> list[FeatureFrame]:
    """Frame-wise features with explicit start/end timestamps (3 s windows, 1 s stride by default)."""
    if frame_size <= 0:
        raise ValueError("frame_size must be greater than zero.")
    if frame_stride <= 0:
        raise ValueError("frame_stride must be greater than zero.")
    active = _resolve(settings)
    audio, sample_rate = read_audio_file(audiofile, audio_read_config=active.audio_read)
    backend = HandcraftedBackend(
        frame_size_seconds=frame_size, frame_stride_seconds=frame_stride, feature_flags=active.feature_flags
    )
    encoded = backend.encode_sequence(audio=audio, sample_rate=sample_rate)
    return [
        FeatureFrame(
            start_seconds=float(encoded.frame_start_seconds[i]),
            end_seconds=float(encoded.frame_end_seconds[i]),
            features=np.asarray(encoded.embeddings[i], dtype=np.float64),
        )
        for i in range(encoded.embeddings.shape[0])
    ]


def extended_extract_feature(audiofile: str, frame_size: int = 3, frame_stride: int = 1, *, settings=None
                             ) -> list[FeatureVector]:
    """Frame-wise feature vectors of an audio file, without the timestamps."""
    return [f.features for f in extract_feature_frames(audiofile, frame_size, frame_stride, settings=settings)]
