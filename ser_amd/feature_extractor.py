"""File-level feature extraction with the reference's public signatures.

``extract_feature`` / ``extract_feature_frames`` / ``extended_extract_feature`` /
``extract_feature_from_signal`` follow
``ser/_internal/features/feature_extractor.py:106-179``; ``settings`` is any object
with ``feature_flags`` and ``audio_read`` attributes (the reference's ``AppConfig``
qualifies).  ``settings=None`` means the reference's defaults (``FeatureFlags()`` with tonnetz on,
``AudioReadConfig()``) instead of reloading an environment snapshot; because tonnetz is not
implemented, that default raises ``NotImplementedError`` unless ``SERHIP_DEFAULT_FLAGS=fast187``
is set (``ser_amd.types.default_feature_flags``).  Pass ``FeatureSettings(FeatureFlags(tonnetz=False))``
or the application's own settings object with tonnetz switched off.
"""

from __future__ import annotations

import logging
from dataclasses import dataclass, field

import numpy as np
from numpy.typing import NDArray

from . import dsp
from .audio_io import read_audio_file
from .handcrafted import HandcraftedBackend
from .types import AudioReadConfig, FeatureFlags, FeatureFrame, FeatureVector, coerce_feature_flags

logger = logging.getLogger(__name__)


@dataclass(frozen=True)
class FeatureSettings:
    """The two settings the path reads from the reference's ``AppConfig``."""

    feature_flags: FeatureFlags = field(default_factory=FeatureFlags)
    audio_read: AudioReadConfig = field(default_factory=AudioReadConfig)


def _resolve(settings) -> FeatureSettings:
    """``settings`` may be None, a :class:`FeatureSettings`, or any object with ``feature_flags`` and
    ``audio_read`` attributes (the reference's ``AppConfig``; its dataclasses are read by attribute)."""
    if settings is None:
        return FeatureSettings(feature_flags=coerce_feature_flags(None))
    audio_read = getattr(settings, "audio_read", None)
    if not isinstance(audio_read, AudioReadConfig):
        audio_read = AudioReadConfig(
            max_retries=int(getattr(audio_read, "max_retries", 3)),
            retry_delay_seconds=float(getattr(audio_read, "retry_delay_seconds", 1.0)),
        )
    return FeatureSettings(feature_flags=coerce_feature_flags(settings.feature_flags), audio_read=audio_read)


def _long_file_on_device(file: str):
    """``(audio device tensor, sample_rate)`` for a long PCM16 / float32 WAVE file read and prepared without a host
    pass over its samples (``ingest.prepared_file_on_device``), else None: short files, every other encoding and
    every failure take ``read_audio_file``, whose results, retries and errors are the reference's
    (``SERHIP_INGEST=sequential`` switches the shortcut off)."""
    import os

    from . import ingest

    try:
        if os.environ.get("SERHIP_INGEST", "pipelined") == "sequential" or os.path.getsize(file) < ingest.DEVICE_PREPARE_FILE_BYTES:
            return None
        from .engine import get_engine

        return ingest.prepared_file_on_device(get_engine(), str(file))
    except OSError:
        return None


def extract_feature_from_signal(audio: NDArray[np.float32], sample_rate: int, *, settings=None) -> FeatureVector:
    """Settings-flavoured wrapper around :func:`ser_amd.dsp.extract_feature_from_signal`."""
    return dsp.extract_feature_from_signal(audio, sample_rate, feature_flags=_resolve(settings).feature_flags)


def extract_feature(file: str, *, settings=None) -> FeatureVector:
    """One feature vector for a whole audio file."""
    active = _resolve(settings)
    on_device = _long_file_on_device(file)
    if on_device is not None:
        from .engine import get_engine
        from .types import require_accelerated

        audio_dev, sample_rate = on_device
        require_accelerated(active.feature_flags)
        dsp.check_flags_for_rate(active.feature_flags, sample_rate)
        row = get_engine(audio_dev.device.index).features(audio_dev, [0], [int(audio_dev.numel())], sample_rate,
                                                         active.feature_flags)
        return np.asarray(row.cpu().numpy()[0], dtype=np.float64)
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
    backend = HandcraftedBackend(
        frame_size_seconds=frame_size, frame_stride_seconds=frame_stride, feature_flags=active.feature_flags
    )
    on_device = _long_file_on_device(audiofile)
    if on_device is not None:
        encoded = backend.encode_sequence_device(*on_device)
    else:
        audio, sample_rate = read_audio_file(audiofile, audio_read_config=active.audio_read)
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
