"""Audio file reading for the fast path (host side, stdlib + numpy only).

Behaviour of ``read_audio_file`` / ``_prepare_audio_buffer``
(``ser/_internal/utils/audio_utils.py:28-162``): native sample rate (never
resampled), NaN/inf scrubbed to 0, channel mean in float32, peak normalisation to
[-1, 1], Git-LFS pointer detection, ``AudioReadConfig`` retries for failures that can be
transient (I/O errors, empty payloads) with ``AudioDecodeError`` after the last one; the
RIFF parser's verdict on the bytes is deterministic and is reported at once, unretried.  The reference decodes through librosa/soundfile, which are not part
of this package; RIFF/WAVE (PCM 8/16/24/32-bit, IEEE float 32/64) is parsed here,
which covers the reference's datasets and its synthetic generator
(``scripts/build_synthetic_ravdess_dataset.py``).
"""

from __future__ import annotations

import logging
import struct
import time
from pathlib import Path
from typing import NamedTuple

import numpy as np
from numpy.typing import NDArray

from .types import AudioReadConfig

logger = logging.getLogger(__name__)
_GIT_LFS_POINTER_PREFIX = b"version https://git-lfs.github.com/spec/v1"


class AudioIntegrityError(OSError):
    """Raised when a path contains metadata in place of audio bytes."""


class AudioDecodeError(OSError):
    """Raised when one otherwise regular media file cannot be decoded locally."""


def prepare_audio_buffer(raw_audio) -> NDArray[np.float32]:
    """The reference's ``_prepare_audio_buffer`` (audio_utils.py:53-60), same contract and same order:
    float32 cast, non-finite -> 0, THEN mono (float32 mean over the channel axis of ``[frames, channels]``),
    peak-normalise.  A NaN or inf in one channel of a 2-D input is therefore dropped from that frame's mean
    (the soundfile-fallback behaviour, audio_utils.py:135-138).  File reads do not hand 2-D data to this
    function: like the reference's primary path they mix down first (:func:`mix_down_like_librosa_load`)."""
    buf = np.array(raw_audio, dtype=np.float32, copy=True)
    np.nan_to_num(buf, copy=False, nan=0.0, posinf=0.0, neginf=0.0)
    if buf.ndim == 2:
        if buf.shape[1] == 0:
            buf = np.empty(0, dtype=np.float32)
        else:
            buf = np.asarray(np.mean(buf, axis=1, dtype=np.float32), dtype=np.float32)
    elif buf.ndim != 1:
        raise OSError(f"Unsupported audio shape: {buf.shape}")
    if buf.size == 0:
        raise OSError("Audio file contains no samples.")
    peak = float(np.max(np.abs(buf)))
    if peak == 0:
        return np.zeros_like(buf)
    return buf / peak


def mix_down_like_librosa_load(frames) -> NDArray[np.float32]:
    """The mono mix-down ``librosa.load(path, sr=None)`` performs before ``read_audio_file`` sees the samples
    (audio_utils.py:104-110; ``mono=True`` is librosa's default): the float32 mean over channels of the decoded
    ``[frames, channels]`` block, with NO scrub in front -- a NaN or inf in one channel makes the whole frame
    non-finite, and ``_prepare_audio_buffer`` then zeroes it.  This is the order ``read_audio_file`` and the
    device kernel ``k_prepare_convert`` follow (both stand in for ``librosa.load`` + ``_prepare_audio_buffer``)."""
    buf = np.asarray(frames, dtype=np.float32)
    if buf.ndim == 2 and buf.shape[1] > 0:
        with np.errstate(invalid="ignore", over="ignore"):
            buf = np.asarray(np.mean(buf, axis=1, dtype=np.float32), dtype=np.float32)
    return buf


def _parse_wave(data: bytes):
    """Minimal RIFF/WAVE parser -> (float32 array [frames] or [frames, channels], sample_rate)."""
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise AudioDecodeError("not a RIFF/WAVE file")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos : pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8 : pos + 8 + size]
        if tag == b"fmt ":
            fmt = body
        elif tag == b"data":
            payload = body
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None or len(fmt) < 16:
        raise AudioDecodeError("WAVE file lacks a fmt or data chunk")
    code, channels, rate, _, block_align, bits = struct.unpack_from("<HHIIHH", fmt, 0)
    if code == 0xFFFE and len(fmt) >= 26:  # WAVE_FORMAT_EXTENSIBLE: real code is in the sub-format GUID
        code = struct.unpack_from("<H", fmt, 24)[0]
    if channels == 0 or rate == 0 or block_align == 0:
        raise AudioDecodeError("WAVE header has zero channels / rate / block size")
    n_frames = len(payload) // block_align
    raw = payload[: n_frames * block_align]
    if code == 1 and bits == 16:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / np.float32(32768.0)
    elif code == 1 and bits == 8:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    elif code == 1 and bits == 24:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = (v.astype(np.float64) / float(1 << 23)).astype(np.float32)
    elif code == 1 and bits == 32:
        x = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / float(1 << 31)).astype(np.float32)
    elif code == 3 and bits == 32:
        x = np.frombuffer(raw, dtype="<f4").astype(np.float32)
    elif code == 3 and bits == 64:
        x = np.frombuffer(raw, dtype="<f8").astype(np.float32)
    else:
        raise AudioDecodeError(f"unsupported WAVE encoding (format {code}, {bits} bits)")
    if channels > 1:
        x = x.reshape(-1, channels)
    return x, int(rate)


class RawClip(NamedTuple):
    """Undecoded sample payload of one file (or segment), for ``FeatureEngine.prepare_audio``."""

    payload: memoryview      # interleaved little-endian samples, `frames * channels` of them
    sample_format: str       # "s16" | "f32"
    channels: int
    frames: int
    sample_rate: int


def read_raw_clip(file_path: str, *, start_seconds: float | None = None,
                  duration_seconds: float | None = None) -> RawClip | None:
    """Header-only read for device-side preparation: PCM16 and float32 RIFF/WAVE payloads are
    returned as bytes (no conversion, no normalisation on the host).  Returns None for encodings
    that need a host decode (8/24/32-bit PCM, float64): callers then use :func:`read_audio_file`.
    File-level errors are the same as :func:`read_audio_file`'s."""
    if start_seconds is not None and start_seconds < 0.0:
        raise ValueError("start_seconds must be >= 0")
    if duration_seconds is not None and duration_seconds <= 0.0:
        raise ValueError("duration_seconds must be > 0")
    path = Path(file_path)
    if not path.exists():
        raise FileNotFoundError(f"Audio file not found: {file_path}")
    if not path.is_file():
        raise OSError(f"Path is not a regular file: {file_path}")
    data = path.read_bytes()
    if data.startswith(_GIT_LFS_POINTER_PREFIX):
        raise AudioIntegrityError(
            f"Audio file is an unmaterialized Git LFS pointer: {file_path}. "
            "Install Git LFS, then run `git lfs pull` and `git lfs checkout` "
            "in the dataset checkout."
        )
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise AudioDecodeError(f"Error reading {file_path}")
    pos, fmt, body_at, body_len = 12, None, -1, 0
    while pos + 8 <= len(data):
        tag, size = data[pos : pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        if tag == b"fmt ":
            fmt = data[pos + 8 : pos + 8 + size]
        elif tag == b"data":
            body_at, body_len = pos + 8, min(size, len(data) - pos - 8)
        pos += 8 + size + (size & 1)
    if fmt is None or body_at < 0 or len(fmt) < 16:
        raise AudioDecodeError(f"Error reading {file_path}")
    code, channels, rate, _, block_align, bits = struct.unpack_from("<HHIIHH", fmt, 0)
    if code == 0xFFFE and len(fmt) >= 26:
        code = struct.unpack_from("<H", fmt, 24)[0]
    if channels == 0 or rate == 0 or block_align == 0:
        raise AudioDecodeError(f"Error reading {file_path}")
    if code == 1 and bits == 16 and block_align == 2 * channels:
        sample_format = "s16"
    elif code == 3 and bits == 32 and block_align == 4 * channels:
        sample_format = "f32"
    else:
        return None
    if channels > 64:
        return None
    n_frames = body_len // block_align
    first = int(float(start_seconds or 0.0) * rate)
    last = n_frames if duration_seconds is None else min(n_frames, first + int(float(duration_seconds) * rate))
    first = min(first, n_frames)
    view = memoryview(data)[body_at + first * block_align : body_at + max(first, last) * block_align]
    return RawClip(view, sample_format, int(channels), max(0, last - first), int(rate))


def decode_file(path: Path, *, start_seconds: float = 0.0, duration_seconds: float | None = None):
    """Decode a file (or a segment of it: frame ``int(start*sr)``, ``int(duration*sr)`` frames)."""
    samples, rate = _parse_wave(path.read_bytes())
    first = int(float(start_seconds) * rate)
    if duration_seconds is None:
        return samples[first:], rate
    return samples[first : first + int(float(duration_seconds) * rate)], rate


def read_audio_file(
    file_path: str,
    *,
    start_seconds: float | None = None,
    duration_seconds: float | None = None,
    audio_read_config: AudioReadConfig | None = None,
) -> tuple[NDArray[np.float32], int]:
    """Read an audio file (or segment) and normalise amplitude to [-1, 1] -> ``(samples, sample_rate)``."""
    if start_seconds is not None and start_seconds < 0.0:
        raise ValueError("start_seconds must be >= 0")
    if duration_seconds is not None and duration_seconds <= 0.0:
        raise ValueError("duration_seconds must be > 0")
    config = audio_read_config if audio_read_config is not None else AudioReadConfig()
    path = Path(file_path)
    if not path.exists():
        raise FileNotFoundError(f"Audio file not found: {file_path}")
    if not path.is_file():
        raise OSError(f"Path is not a regular file: {file_path}")
    with path.open("rb") as handle:
        if handle.read(len(_GIT_LFS_POINTER_PREFIX)) == _GIT_LFS_POINTER_PREFIX:
            raise AudioIntegrityError(
                f"Audio file is an unmaterialized Git LFS pointer: {file_path}. "
                "Install Git LFS, then run `git lfs pull` and `git lfs checkout` "
                "in the dataset checkout."
            )
    last_error: Exception | None = None
    for attempt in range(config.max_retries):
        try:
            samples, rate = decode_file(
                path,
                start_seconds=float(start_seconds or 0.0),
                duration_seconds=float(duration_seconds) if duration_seconds is not None else None,
            )
            return prepare_audio_buffer(mix_down_like_librosa_load(samples)), int(rate)
        except AudioDecodeError as err:
            # the parser's verdict on these bytes is deterministic: retrying (and sleeping) cannot change
            # it.  Only WAVE is decoded here (module docstring); name the container / encoding at once.
            logger.error("Failed to read audio file %s: %s", file_path, err)
            raise AudioDecodeError(f"Error reading {file_path}: {err}") from err
        except Exception as err:
            last_error = err
            detail = str(err).strip() or type(err).__name__
            logger.warning("Failed to read audio file: %s", detail)
            if attempt < config.max_retries - 1:
                logger.info("Retrying in %s seconds...", config.retry_delay_seconds)
                time.sleep(config.retry_delay_seconds)
            elif start_seconds is not None or duration_seconds is not None:
                raise
    logger.error("Failed to read audio file %s after %s retries.", file_path, config.max_retries)
    error = AudioDecodeError(f"Error reading {file_path}")
    if last_error is None:
        raise error
    raise error from last_error
