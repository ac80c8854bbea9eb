"""``HandcraftedBackend`` -- the fast profile's feature backend, GPU edition.

Interface of ``ser/_internal/repr/handcrafted.py:22-137`` (constructor keywords,
``backend_id``, ``feature_dim``, ``prepare_runtime``, ``encode_sequence``, ``pool``,
``extract_vector``).  Where the reference loops over windows in Python and calls the
DSP function once per window, this backend describes all windows of a recording
as (offset, length) pairs and featurises them in ONE batched GPU call; the
batch entry points ``extract_vectors`` / ``encode_sequences`` do the same
across many recordings (the dataset featurisation loop of
``ser/_internal/data/data_loader.py:335-535``).
"""

from __future__ import annotations

import os
from collections.abc import Sequence

import numpy as np
from numpy.typing import NDArray

from . import dsp
from .engine import FeatureEngine, get_engine
from .types import (
    EncodedSequence,
    FeatureFlags,
    FeatureMatrix,
    FeatureVector,
    PoolingWindow,
    coerce_feature_flags,
    require_accelerated,
)


def sliding_window_bounds(n_samples: int, sample_rate: int, size_seconds, stride_seconds):
    """Window sample ranges of ``encode_sequence`` (handcrafted.py:78-97) as two int64 arrays."""
    frame_length = max(1, int(round(size_seconds * sample_rate)))
    frame_step = max(1, int(round(stride_seconds * sample_rate)))
    starts = np.arange(0, n_samples, frame_step, dtype=np.int64)
    ends = np.minimum(starts + frame_length, n_samples)
    return starts, ends


def _at_least(arrays, nbytes: int) -> bool:
    """Do the arrays hold ``nbytes`` or more between them?  (Stops counting as soon as they do.)"""
    total = 0
    for a in arrays:
        total += getattr(a, "nbytes", 0)
        if total >= nbytes:
            return True
    return False


class HandcraftedBackend:
    """Handcrafted (MFCC / chroma / mel / contrast) backend running on gfx950."""

    def __init__(
        self,
        *,
        frame_size_seconds: int = 3,
        frame_stride_seconds: int = 1,
        feature_flags: FeatureFlags | None = None,
    ) -> None:
        if frame_size_seconds <= 0:
            raise ValueError("frame_size_seconds must be greater than zero.")
        if frame_stride_seconds <= 0:
            raise ValueError("frame_stride_seconds must be greater than zero.")
        self._frame_size_seconds = frame_size_seconds
        self._frame_stride_seconds = frame_stride_seconds
        # None is the reference's default (tonnetz on, feature_dim 193, handcrafted.py:38): extraction then
        # raises NotImplementedError unless SERHIP_DEFAULT_FLAGS=fast187 (types.default_feature_flags)
        self._feature_flags = coerce_feature_flags(feature_flags)

    @property
    def backend_id(self) -> str:
        return "handcrafted"

    @property
    def feature_dim(self) -> int:
        return self._feature_flags.width()

    @property
    def feature_flags(self) -> FeatureFlags:
        return self._feature_flags

    def prepare_runtime(self) -> None:
        """Warm-up hook: creates the HIP context and uploads nothing else (plans are per sample rate)."""
        get_engine()

    # ------------------------------------------------------------------ single recording
    def encode_sequence(self, audio: NDArray[np.float32], sample_rate: int) -> EncodedSequence:
        """Frame-level features of one recording: all sliding windows in one GPU batch."""
        if sample_rate <= 0:
            raise ValueError("sample_rate must be a positive integer.")
        if audio.ndim != 1:
            raise ValueError("audio must be mono (1D array).")
        if audio.size == 0:
            raise ValueError("audio must contain at least one sample.")
        return self.encode_sequences([audio], sample_rate)[0]

    def encode_sequence_device(self, audio_dev, sample_rate: int) -> EncodedSequence:
        """:meth:`encode_sequence` for a recording that already sits on the GPU as prepared float32 samples
        (``ingest.prepared_file_on_device``: finite by construction -- the device prepare scrubs): the sliding
        windows are descriptors into it."""
        if sample_rate <= 0:
            raise ValueError("sample_rate must be a positive integer.")
        if audio_dev.dim() != 1:
            raise ValueError("audio must be mono (1D array).")
        n = int(audio_dev.numel())
        if n == 0:
            raise ValueError("audio must contain at least one sample.")
        require_accelerated(self._feature_flags)
        dsp.check_flags_for_rate(self._feature_flags, sample_rate)
        starts, ends = sliding_window_bounds(n, sample_rate, self._frame_size_seconds, self._frame_stride_seconds)
        if starts.size == 0:
            raise ValueError("Could not extract handcrafted features from provided audio.")
        if self._feature_flags.width():
            emb = get_engine(audio_dev.device.index).features(audio_dev, starts, ends - starts, sample_rate,
                                                              self._feature_flags).cpu().numpy()
        else:
            emb = np.empty((starts.size, 0), dtype=np.float32)
        return EncodedSequence(
            embeddings=np.ascontiguousarray(emb, dtype=np.float32),
            frame_start_seconds=starts.astype(np.float64) / float(sample_rate),
            frame_end_seconds=ends.astype(np.float64) / float(sample_rate),
            backend_id=self.backend_id,
        )

    def extract_vector(self, audio: NDArray[np.float32], sample_rate: int) -> FeatureVector:
        """One feature vector for the whole signal (training path)."""
        return np.asarray(
            dsp.extract_feature_from_signal(audio, sample_rate, feature_flags=self._feature_flags),
            dtype=np.float64,
        )

    def pool(self, encoded: EncodedSequence, windows: Sequence[PoolingWindow]) -> FeatureMatrix:
        """Mean-pools encoded frames for each window (float32 accumulation, float64 result)."""
        dim = int(encoded.embeddings.shape[1])
        if not windows:
            return np.empty((0, dim), dtype=np.float64)
        from .pool import window_frame_ranges

        lo, hi = window_frame_ranges(encoded, windows)
        out = get_engine().pool_windows(np.asarray(encoded.embeddings, dtype=np.float32), lo, hi, mode=1)
        return out.cpu().numpy()

    # ------------------------------------------------------------------ batches
    def extract_vectors(self, audios: Sequence[NDArray[np.float32]], sample_rate: int) -> FeatureMatrix:
        """``[len(audios), D]`` float64: one vector per whole clip, one GPU batch for all clips."""
        dim = self._feature_flags.width()
        if (dim and len(audios) > 1 and os.environ.get("SERHIP_INGEST", "pipelined") != "sequential"
                and _at_least(audios, FeatureEngine.STREAM_FROM_HOST_BYTES)):
            # a large batch: the clips are neither concatenated nor tested for non-finite samples on the host -- the
            # staging threads gather them chunk by chunk and test them in the same pass (ingest.features_from_host_arrays)
            from .ingest import features_from_host_arrays

            if sample_rate <= 0:
                raise ValueError("Sample rate must be a positive integer.")
            for a in audios:   # validate_signal's shape checks, inline: 12 500 calls of it cost 12 ms
                if a.ndim != 1:
                    raise ValueError("Audio must be mono (1D array).")
                if a.size == 0:
                    raise ValueError("Audio contains no samples.")
            require_accelerated(self._feature_flags)
            dsp.check_flags_for_rate(self._feature_flags, sample_rate)
            engine = get_engine()
            return engine.to_host_float64(features_from_host_arrays(engine, audios, sample_rate, self._feature_flags))
        prepared = [dsp.validate_signal(a, sample_rate) for a in audios]
        require_accelerated(self._feature_flags)
        if dim == 0 or not prepared:
            return np.empty((len(prepared), dim), dtype=np.float64)
        dsp.check_flags_for_rate(self._feature_flags, sample_rate)
        lengths = np.asarray([a.size for a in prepared], dtype=np.int64)
        offsets = np.concatenate(([0], np.cumsum(lengths)[:-1])).astype(np.int64)
        packed = np.concatenate(prepared)
        out = get_engine().features(packed, offsets, lengths, sample_rate, self._feature_flags)
        return out.cpu().numpy().astype(np.float64)

    def encode_sequences(self, audios: Sequence[NDArray[np.float32]], sample_rate: int) -> list[EncodedSequence]:
        """Sliding-window features of many recordings in one GPU batch."""
        if sample_rate <= 0:
            raise ValueError("sample_rate must be a positive integer.")
        prepared, bounds = [], []
        dim = self._feature_flags.width()
        # one long recording: its samples are tested for NaN / infinity by the threads that stage them for the GPU
        # (engine.features(check_finite=True)), not by a pass over the buffer in front of everything else
        defer = (len(audios) == 1 and dim > 0 and isinstance(audios[0], np.ndarray)
                 and audios[0].nbytes >= FeatureEngine.STREAM_FROM_HOST_BYTES)
        for audio in audios:
            if audio.ndim != 1:
                raise ValueError("audio must be mono (1D array).")
            if audio.size == 0:
                raise ValueError("audio must contain at least one sample.")
            a = dsp.validate_signal(audio, sample_rate, finite_check=not defer)
            prepared.append(a)
            bounds.append(sliding_window_bounds(a.size, sample_rate, self._frame_size_seconds,
                                                self._frame_stride_seconds))
        if not prepared:
            return []
        require_accelerated(self._feature_flags)
        dsp.check_flags_for_rate(self._feature_flags, sample_rate)
        base = np.concatenate(([0], np.cumsum([a.size for a in prepared])[:-1])).astype(np.int64)
        offsets = np.concatenate([b + s for b, (s, _) in zip(base, bounds)])
        lengths = np.concatenate([e - s for s, e in bounds])
        if dim:
            packed = prepared[0] if len(prepared) == 1 else np.concatenate(prepared)
            emb = get_engine().features(packed, offsets, lengths, sample_rate, self._feature_flags,
                                        check_finite=defer).cpu().numpy()
        else:
            emb = np.empty((offsets.size, 0), dtype=np.float32)
        result, row = [], 0
        for starts, ends in bounds:
            n = starts.size
            if n == 0:
                raise ValueError("Could not extract handcrafted features from provided audio.")
            result.append(
                EncodedSequence(
                    embeddings=np.ascontiguousarray(emb[row : row + n], dtype=np.float32),
                    frame_start_seconds=starts.astype(np.float64) / float(sample_rate),
                    frame_end_seconds=ends.astype(np.float64) / float(sample_rate),
                    backend_id=self.backend_id,
                )
            )
            row += n
        return result
