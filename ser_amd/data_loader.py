"""Dataset featurisation for fast-profile training -- the GPU-batched counterpart of
``ser/_internal/data/data_loader.py``.

The reference extracts one file at a time, either in a ``multiprocessing.Pool`` of up to 8
workers (``_load_data_for_settings``, data_loader.py:335-444) or serially over
readiness-checked utterances (``load_checked_fast_data``, :467-535).  Here files are decoded
on host threads, packed into one buffer and featurised in ONE GPU batch per sample rate
(``HandcraftedBackend.extract_vectors``); across GPUs each rank takes a contiguous shard of the
sorted file list and one all-gather assembles the matrix (``ser_amd.distributed``).

Kept from the reference: RAVDESS file-name label parsing, the observed-emotion filter, per-file
error strings, the failure-ratio abort and its message, the >= 2 classes rule, and the
``train_test_split(test_size, random_state, stratify)`` call with the non-stratified retry.
The reference's ``imap_unordered`` returns rows in arbitrary order before the split; this
loader keeps sorted-file order, so its split is deterministic.
"""

from __future__ import annotations

import glob
import logging
import os
from collections.abc import Callable, Collection, Mapping, Sequence
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from pathlib import Path
from types import MappingProxyType
from typing import NamedTuple

import numpy as np
from numpy.typing import NDArray

from .audio_io import AudioIntegrityError, read_audio_file, read_raw_clip
from .handcrafted import HandcraftedBackend
from .types import FAST_PATH_FLAGS, AudioReadConfig, FeatureFlags

logger = logging.getLogger(__name__)

FeatureVector = NDArray[np.float64]
FeatureMatrix = NDArray[np.float64]
DataSplit = tuple[FeatureMatrix, FeatureMatrix, list[str], list[str]]

#: RAVDESS emotion codes (ser/_internal/config/settings_builder.py:16-27)
RAVDESS_EMOTIONS: Mapping[str, str] = MappingProxyType({
    "01": "neutral", "02": "calm", "03": "happy", "04": "sad",
    "05": "angry", "06": "fearful", "07": "disgust", "08": "surprised",
})


@dataclass(frozen=True)
class LoaderSettings:
    """The settings this path reads from the reference's ``AppConfig`` (same defaults):
    ``dataset.glob_pattern``, ``emotions``, ``training.{test_size,random_state,stratify_split}``,
    ``data_loader.{max_workers,max_failed_file_ratio}``, ``models.num_cores``, ``feature_flags``,
    ``audio_read`` (config/schema.py:243-289)."""

    dataset_folder: Path = Path("ser/dataset/ravdess")
    subfolder_prefix: str = "Actor_*"
    extension: str = "*.wav"
    emotions: Mapping[str, str] = RAVDESS_EMOTIONS
    test_size: float = 0.25
    random_state: int = 42
    stratify_split: bool = True
    max_workers: int = 8
    num_cores: int = field(default_factory=lambda: os.cpu_count() or 1)
    max_failed_file_ratio: float = 0.01
    feature_flags: FeatureFlags = FAST_PATH_FLAGS
    audio_read: AudioReadConfig = field(default_factory=AudioReadConfig)

    @property
    def glob_pattern(self) -> str:
        return str(Path(self.dataset_folder) / self.subfolder_prefix / self.extension)


class ProcessFileResult(NamedTuple):
    """Per-file extraction success or failure (same shape as the reference's)."""

    sample: tuple[FeatureVector, str] | None
    error: str | None


def _extract_emotion_code(file_name: str) -> str | None:
    parts = file_name.split("-")
    return None if len(parts) < 3 else parts[2]


def extract_ravdess_speaker_id_from_path(file_path: str) -> str | None:
    parts = os.path.basename(file_path).split("-")
    if len(parts) < 7:
        return None
    return parts[6].split(".")[0].strip() or None


def _resolve_worker_count(max_cores: int, max_workers: int, file_count: int) -> int:
    if file_count <= 0:
        return 1
    return max(1, min(max_cores, max_workers, file_count))


def label_for_file(file: str, observed_emotions: Collection[str], emotion_map: Mapping[str, str]):
    """``(label, error)``: label None + error None means "not a target emotion" (silently skipped)."""
    file_name = os.path.basename(file)
    code = _extract_emotion_code(file_name)
    if code is None:
        return None, f"Skipping file with unexpected name format (missing emotion code): {file_name}"
    emotion = emotion_map.get(code)
    if not emotion or emotion not in observed_emotions:
        return None, None
    return emotion, None


def process_file(file: str, observed_emotions: Collection[str], emotion_map: Mapping[str, str], *,
                 settings=None, feature_flags: FeatureFlags | None = None,
                 audio_read_config: AudioReadConfig | None = None) -> ProcessFileResult:
    """One-file form of the loader (reference signature); batches should use :func:`featurize_files`."""
    try:
        emotion, error = label_for_file(file, observed_emotions, emotion_map)
        if emotion is None:
            return ProcessFileResult(sample=None, error=error)
        if settings is not None:
            feature_flags, audio_read_config = settings.feature_flags, settings.audio_read
        if feature_flags is None or audio_read_config is None:
            raise ValueError("process_file requires either settings or explicit feature extraction config.")
        audio, sample_rate = read_audio_file(file, audio_read_config=audio_read_config)
        vector = HandcraftedBackend(feature_flags=feature_flags).extract_vector(audio, sample_rate)
        return ProcessFileResult(sample=(np.asarray(vector, dtype=np.float64), emotion), error=None)
    except Exception as err:
        return ProcessFileResult(sample=None, error=f"Failed to process file {file}: {err}")


def _read_one(job):
    """Read one job.  Returns ``(kind, value, error)``: kind "raw" (a RawClip whose samples the GPU
    converts and normalises), "host" (``(samples, rate)`` decoded and prepared on the host: encodings
    the device path does not take) or None with the error."""
    path, start, duration, config, device_prepare = job
    try:
        if device_prepare:
            try:
                raw = read_raw_clip(path, start_seconds=start, duration_seconds=duration)
            except (FileNotFoundError, AudioIntegrityError, ValueError):
                raise
            except Exception:
                raw = None   # let the host reader produce the reference's retries and error
            if raw is not None and raw.frames > 0:
                return "raw", raw, None
        return "host", read_audio_file(path, start_seconds=start, duration_seconds=duration,
                                       audio_read_config=config), None
    except Exception as err:  # counted per file by the caller, like the reference
        return None, None, err


def _featurize_raw_group(clips, rate: int, sample_format: str, feature_flags: FeatureFlags):
    """One GPU batch for raw clips of one (sample rate, sample format): payload bytes are packed
    back to back (16-byte aligned), converted / mixed down / peak-normalised on the device
    (``serhip_prepare_audio``) and featurised from there -- the samples never exist on the host."""
    from . import dsp
    from .engine import get_engine

    dsp.check_flags_for_rate(feature_flags, rate)
    offsets, cursor = [], 0
    for clip in clips:
        offsets.append(cursor)
        cursor += (len(clip.payload) + 15) // 16 * 16
    packed = np.zeros(cursor, dtype=np.uint8)
    for clip, at in zip(clips, offsets):
        packed[at : at + len(clip.payload)] = np.frombuffer(clip.payload, dtype=np.uint8)
    engine = get_engine()
    audio, out_off, frames = engine.prepare_audio(packed, sample_format, offsets, [c.frames for c in clips],
                                                  [c.channels for c in clips])
    return engine.features(audio, out_off, frames, rate, feature_flags).cpu().numpy().astype(np.float64)


def featurize_audio_jobs(jobs: Sequence[tuple[str, float | None, float | None]], *, feature_flags: FeatureFlags,
                         audio_read_config: AudioReadConfig, workers: int = 8, device_prepare: bool = True):
    """Read ``(path, start_seconds, duration_seconds)`` jobs on host threads and featurise every
    readable clip in one GPU batch per (sample rate, sample format).

    With ``device_prepare`` (default) PCM16 / float32 WAVE payloads go to the GPU as raw bytes and
    ``_prepare_audio_buffer`` (audio_utils.py:28-60) runs there; other encodings are decoded and
    prepared on the host as before.  Returns ``(matrix [n_jobs, D] float64, errors)`` where
    ``errors[i]`` is None or the exception of job i (its matrix row is NaN)."""
    n = len(jobs)
    dim = feature_flags.width()
    matrix = np.full((n, dim), np.nan, dtype=np.float64)
    errors: list[Exception | None] = [None] * n
    if n == 0:
        return matrix, errors
    if feature_flags.tonnetz:
        raise NotImplementedError("tonnetz features are outside the accelerated path; use FeatureFlags(tonnetz=False)")
    packed = [(p, s, d, audio_read_config, device_prepare) for p, s, d in jobs]
    if workers > 1 and n > 1:
        with ThreadPoolExecutor(max_workers=workers) as pool:
            decoded = list(pool.map(_read_one, packed))
    else:
        decoded = [_read_one(j) for j in packed]
    raw_groups: dict[tuple[int, str], list[int]] = {}
    by_rate: dict[int, list[int]] = {}
    for i, (kind, value, err) in enumerate(decoded):
        if kind is None:
            errors[i] = err
        elif kind == "raw":
            raw_groups.setdefault((value.sample_rate, value.sample_format), []).append(i)
        else:
            by_rate.setdefault(int(value[1]), []).append(i)
    for (rate, sample_format), idxs in raw_groups.items():
        try:
            matrix[idxs] = _featurize_raw_group([decoded[i][1] for i in idxs], rate, sample_format, feature_flags)
        except Exception as err:
            for i in idxs:
                errors[i] = err
    backend = HandcraftedBackend(feature_flags=feature_flags)
    for rate, idxs in by_rate.items():
        clips, good = [], []
        for i in idxs:
            clip = decoded[i][1][0]
            try:  # per-clip validation errors must not poison the batch
                from .dsp import validate_signal

                validate_signal(clip, rate)
                clips.append(clip)
                good.append(i)
            except Exception as err:
                errors[i] = err
        if not clips:
            continue
        try:
            matrix[good] = backend.extract_vectors(clips, rate)
        except Exception as err:
            for i in good:
                errors[i] = err
    return matrix, errors


def featurize_files(files: Sequence[str], *, settings: LoaderSettings | None = None, distributed: bool = False):
    """Feature matrix for whole files (training vectors).  With ``distributed=True`` and an initialised
    ``torch.distributed`` group, each rank featurises its contiguous shard and the rows are all-gathered."""
    active = settings if settings is not None else LoaderSettings()
    files = list(files)
    workers = _resolve_worker_count(active.num_cores, active.max_workers, len(files))
    if not distributed:
        return featurize_audio_jobs([(f, None, None) for f in files], feature_flags=active.feature_flags,
                                    audio_read_config=active.audio_read, workers=workers)
    import torch
    import torch.distributed as dist

    from .distributed import all_gather_rows, shard_range

    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(len(files), rank, world)
    local, local_err = featurize_audio_jobs([(f, None, None) for f in files[lo:hi]],
                                            feature_flags=active.feature_flags,
                                            audio_read_config=active.audio_read, workers=workers)
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    full = all_gather_rows(torch.from_numpy(local).to(device), len(files)).cpu().numpy()
    errors: list[Exception | None] = [None] * len(files)
    for i in np.flatnonzero(np.isnan(full).any(axis=1)):
        errors[int(i)] = local_err[int(i) - lo] if lo <= i < hi else RuntimeError("extraction failed on another rank")
    return full, errors


def load_data(test_size: float | None = None, *, settings: LoaderSettings | None = None) -> DataSplit | None:
    """Glob the dataset, featurise on the GPU and split -> ``(x_train, x_test, y_train, y_test)`` or None."""
    from sklearn.model_selection import train_test_split

    active = settings if settings is not None else LoaderSettings()
    observed = set(active.emotions.values())
    resolved_test_size = active.test_size if test_size is None else test_size
    if not 0.0 < resolved_test_size < 1.0:
        raise ValueError("test_size must be between 0 and 1.")
    files = sorted(glob.glob(active.glob_pattern))
    if not files:
        logger.warning("No dataset files found for pattern: %s", active.glob_pattern)
        return None

    errors: list[str] = []
    wanted: list[tuple[str, str]] = []
    for file in files:
        label, error = label_for_file(file, observed, active.emotions)
        if error is not None:
            errors.append(error)
        elif label is not None:
            wanted.append((file, label))
    matrix, failures = featurize_files([f for f, _ in wanted], settings=active)
    rows, labels = [], []
    for (file, label), row, failure in zip(wanted, matrix, failures):
        if failure is not None:
            errors.append(f"Failed to process file {file}: {failure}")
        else:
            rows.append(row)
            labels.append(label)
    if errors:
        logger.warning("Skipped %s/%s files during feature extraction.", len(errors), len(files))
        for error in errors[:5]:
            logger.warning("%s", error)
    failure_ratio = len(errors) / float(len(files))
    if failure_ratio > active.max_failed_file_ratio:
        raise RuntimeError(
            "Aborting data load: "
            f"{failure_ratio * 100.0:.1f}% file failures exceeded configured "
            "limit "
            f"{active.max_failed_file_ratio * 100.0:.1f}%. "
            "You can relax this limit by increasing the SER_MAX_FAILED_FILE_RATIO "
            "environment variable."
        )
    if not rows:
        logger.warning("No data found or processed.")
        return None
    if len(set(labels)) < 2:
        logger.warning("At least two emotion classes are required to train the model.")
        return None
    features = np.asarray(rows, dtype=np.float64)
    try:
        split = train_test_split(features, labels, test_size=resolved_test_size, random_state=active.random_state,
                                 stratify=labels if active.stratify_split else None)
    except ValueError as err:
        logger.warning("Stratified split failed (%s). Falling back to non-stratified split.", err)
        split = train_test_split(features, labels, test_size=resolved_test_size, random_state=active.random_state,
                                 stratify=None)
    return (np.asarray(split[0], dtype=np.float64), np.asarray(split[1], dtype=np.float64),
            [str(v) for v in split[2]], [str(v) for v in split[3]])


def load_checked_fast_data(*, train_utterances: Sequence, test_utterances: Sequence, settings: LoaderSettings,
                           handle_sample_failure: Callable[[object, Exception], bool] | None = None
                           ) -> DataSplit | None:
    """Fast features for readiness-checked utterances (reference: data_loader.py:467-535).

    The reference splits inside (``split_utterances``, dataset governance, out of scope); here the
    caller passes the two partitions.  An utterance is any object with ``audio_path``,
    ``start_seconds``, ``duration_seconds``, ``sample_id`` and ``require_label()``.  Each partition is
    one GPU batch; per-sample failures go through ``handle_sample_failure`` exactly as in the reference
    (True = skip the sample, otherwise the error is raised)."""
    if not train_utterances and not test_utterances:
        return None

    def extract(partition: Sequence):
        jobs = [(str(u.audio_path), u.start_seconds, u.duration_seconds) for u in partition]
        matrix, failures = featurize_audio_jobs(jobs, feature_flags=settings.feature_flags,
                                                audio_read_config=settings.audio_read,
                                                workers=_resolve_worker_count(settings.num_cores, settings.max_workers,
                                                                              len(jobs)))
        rows, labels = [], []
        for utterance, row, failure in zip(partition, matrix, failures):
            if failure is not None:
                if handle_sample_failure is not None and handle_sample_failure(utterance, failure):
                    continue
                raise failure
            if row.ndim != 1 or row.size <= 0 or not np.all(np.isfinite(row)):
                raise ValueError(f"Fast feature contract failed for sample {utterance.sample_id!r}.")
            rows.append(row)
            labels.append(utterance.require_label())
        if not rows:
            raise RuntimeError("Fast checked preparation produced an empty split partition.")
        return np.vstack(rows).astype(np.float64, copy=False), labels

    x_train, y_train = extract(train_utterances)
    x_test, y_test = extract(test_utterances)
    if len(set(y_train)) < 2:
        raise RuntimeError("Fast checked preparation left fewer than two training classes.")
    return x_train, x_test, y_train, y_test
