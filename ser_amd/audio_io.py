"""Audio file reading for the fast path (host side, stdlib + numpy only).

Behaviour of ``read_audio_file`` / ``_prepare_audio_buffer``
(``ser/_internal/utils/audio_utils.py:28-162``): native sample rate (never
resampled), NaN/inf scrubbed to 0, channel mean in float32, peak normalisation to
[-1, 1], Git-LFS pointer detection, ``AudioReadConfig`` retries for failures that can be
transient (I/O errors, empty payloads) with ``AudioDecodeError`` after the last one; a
parser's verdict on the bytes is deterministic and is reported at once, unretried.  The reference decodes
through librosa/soundfile (libsndfile), which are not part of this package.  Decoded here, by container magic
(never by file extension): RIFF/WAVE and RF64 (PCM 8/16/24/32-bit, IEEE float 32/64, A-law, mu-law,
WAVE_FORMAT_EXTENSIBLE), FLAC (native decoder, ``serhip_flac_decode``), AIFF / AIFF-C (big- and little-endian
PCM, float 32/64, A-law, mu-law) and Sun/NeXT AU (PCM 8/16/24/32, float 32/64, A-law, mu-law) -- the lossless
and uncompressed formats libsndfile reads, scaled as libsndfile scales them for a float read (integer PCM by
``2**(bits-1)``, companded formats through their 16-bit expansion).  Ogg/Vorbis, Opus and MP3 are lossy codecs
whose decoders are not here: ``AudioDecodeError``.
"""

from __future__ import annotations

import logging
import struct
import threading
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


def _g711_tables():
    """16-bit expansions of the 256 A-law and mu-law codes (ITU-T G.711), the tables libsndfile decodes with."""
    codes = np.arange(256, dtype=np.int32)
    u = ~codes & 0xFF
    t = (((u & 0x0F) << 3) + 0x84) << ((u & 0x70) >> 4)
    ulaw = np.where(u & 0x80, 0x84 - t, t - 0x84).astype(np.int16)
    a = codes ^ 0x55
    seg = (a & 0x70) >> 4
    t = (a & 0x0F) << 4
    t = np.where(seg == 0, t + 8, (t + 0x108) << np.maximum(seg - 1, 0))
    alaw = np.where(a & 0x80, t, -t).astype(np.int16)
    return alaw, ulaw


_ALAW16, _ULAW16 = _g711_tables()


def _int_pcm_to_float(raw: bytes, bits: int, *, big_endian: bool, unsigned8: bool = False) -> NDArray[np.float32]:
    """Integer PCM -> float32 in [-1, 1) the way libsndfile's float reads scale it: ``x / 2**(bits-1)``."""
    order = ">" if big_endian else "<"
    if bits == 8:
        if unsigned8:
            return (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
        return np.frombuffer(raw, dtype=np.int8).astype(np.float32) / np.float32(128.0)
    if bits == 16:
        return np.frombuffer(raw, dtype=order + "i2").astype(np.float32) / np.float32(32768.0)
    if bits == 24:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = (b[:, 0] << 16) | (b[:, 1] << 8) | b[:, 2] if big_endian else b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        return (v.astype(np.float64) / float(1 << 23)).astype(np.float32)
    if bits == 32:
        return (np.frombuffer(raw, dtype=order + "i4").astype(np.float64) / float(1 << 31)).astype(np.float32)
    raise AudioDecodeError(f"unsupported PCM width ({bits} bits)")


def _float_payload(raw: bytes, dtype: str) -> NDArray[np.float32]:
    """IEEE float samples -> float32; a float64 beyond float32's range becomes inf (then scrubbed like any other
    non-finite sample, audio_utils.py:53-55) without a numpy warning per file."""
    with np.errstate(over="ignore", invalid="ignore"):
        return np.frombuffer(raw, dtype=dtype).astype(np.float32)


def _companded_to_float(raw: bytes, table: NDArray[np.int16]) -> NDArray[np.float32]:
    return table[np.frombuffer(raw, dtype=np.uint8)].astype(np.float32) / np.float32(32768.0)


def _parse_flac(data: bytes):
    """FLAC -> (float32 [frames] or [frames, channels], sample_rate) through the native decoder (frame CRCs and the
    STREAMINFO MD5 verified there); integer samples scaled by ``2**(bits-1)`` like libsndfile's float read."""
    import ctypes
    from ctypes import POINTER, byref, c_int32, c_int64, c_void_p

    from . import _native

    lib = _native.load_library()
    rate, channels, bits, total = c_int32(), c_int32(), c_int32(), c_int64()
    buf = (ctypes.c_char * len(data)).from_buffer_copy(data) if not isinstance(data, bytes) else data
    try:
        _native.check(lib.serhip_flac_probe(buf, len(data), byref(rate), byref(channels), byref(bits), byref(total)))
        done = c_int64()
        if total.value == 0:   # STREAMINFO does not say: count first
            _native.check(lib.serhip_flac_decode(buf, len(data), None, 0, 0, byref(done)))
            total.value = done.value
        pcm = np.empty((int(total.value), int(channels.value)), dtype=np.int32)
        _native.check(lib.serhip_flac_decode(buf, len(data), pcm.ctypes.data_as(c_void_p), int(total.value), 1, byref(done)))
    except _native.SerhipError as err:
        raise AudioDecodeError(str(err).split(": ", 2)[-1]) from err
    x = (pcm[: int(done.value)].astype(np.float64) / float(1 << (int(bits.value) - 1))).astype(np.float32)
    return (x[:, 0] if channels.value == 1 else x), int(rate.value)


def _ieee_extended(b: bytes) -> float:
    """80-bit IEEE 754 extended (AIFF's sample-rate field) -> float."""
    exponent = ((b[0] & 0x7F) << 8) | b[1]
    mantissa = int.from_bytes(b[2:10], "big")
    if exponent == 0 and mantissa == 0:
        return 0.0
    value = mantissa * 2.0 ** (exponent - 16383 - 63)
    return -value if b[0] & 0x80 else value


def _parse_aiff(data: bytes):
    """AIFF / AIFF-C -> (float32 [frames] or [frames, channels], sample_rate)."""
    if len(data) < 12 or data[:4] != b"FORM" or data[8:12] not in (b"AIFF", b"AIFC"):
        raise AudioDecodeError("not an AIFF file")
    compressed = data[8:12] == b"AIFC"
    pos, comm, sound = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos : pos + 4], struct.unpack_from(">I", data, pos + 4)[0]
        body = data[pos + 8 : pos + 8 + size]
        if tag == b"COMM":
            comm = body
        elif tag == b"SSND" and len(body) >= 8:
            offset = struct.unpack_from(">I", body, 0)[0]
            sound = body[8 + offset :]
        pos += 8 + size + (size & 1)
    if comm is None or len(comm) < 18:
        raise AudioDecodeError("AIFF file lacks a COMM chunk")
    channels, n_frames, bits = struct.unpack_from(">hIh", comm, 0)
    rate = int(round(_ieee_extended(comm[8:18])))
    kind = comm[18:22] if compressed and len(comm) >= 22 else b"NONE"
    if channels <= 0 or rate <= 0:
        raise AudioDecodeError("AIFF header has zero channels or rate")
    if sound is None:
        if n_frames == 0:
            return np.empty(0, dtype=np.float32), rate
        raise AudioDecodeError("AIFF file lacks an SSND chunk")
    width = (bits + 7) // 8
    if kind in (b"NONE", b"twos", b"sowt", b"in24", b"in32", b"raw "):
        big = kind != b"sowt"
        if kind == b"raw ":
            frame_bytes = channels
        else:
            frame_bytes = channels * width
        usable = min(len(sound), n_frames * frame_bytes) // frame_bytes * frame_bytes
        if kind == b"raw ":
            x = _int_pcm_to_float(sound[:usable], 8, big_endian=True, unsigned8=True)
        else:
            if bits not in (8, 16, 24, 32):
                raise AudioDecodeError(f"unsupported AIFF sample size ({bits} bits)")
            x = _int_pcm_to_float(sound[:usable], bits, big_endian=big)
    elif kind in (b"fl32", b"FL32", b"fl64", b"FL64"):
        item = 4 if kind.lower() == b"fl32" else 8
        usable = min(len(sound), n_frames * channels * item) // (channels * item) * (channels * item)
        x = _float_payload(sound[:usable], ">f4" if item == 4 else ">f8")
    elif kind in (b"alaw", b"ALAW", b"ulaw", b"ULAW"):
        usable = min(len(sound), n_frames * channels) // channels * channels
        x = _companded_to_float(sound[:usable], _ALAW16 if kind.lower() == b"alaw" else _ULAW16)
    else:
        raise AudioDecodeError(f"unsupported AIFF-C compression {kind!r}")
    if channels > 1:
        x = x.reshape(-1, channels)
    return x, rate


def _parse_au(data: bytes):
    """Sun/NeXT AU -> (float32 [frames] or [frames, channels], sample_rate)."""
    if len(data) < 24 or data[:4] != b".snd":
        raise AudioDecodeError("not an AU file")
    offset, size, encoding, rate, channels = struct.unpack_from(">IIIII", data, 4)
    if channels == 0 or rate == 0 or offset < 24:
        raise AudioDecodeError("AU header has zero channels / rate or a bad data offset")
    body = data[offset:] if size == 0xFFFFFFFF else data[offset : offset + size]
    if encoding == 1:
        x = _companded_to_float(body[: len(body) // channels * channels], _ULAW16)
    elif encoding == 27:
        x = _companded_to_float(body[: len(body) // channels * channels], _ALAW16)
    elif encoding in (2, 3, 4, 5):
        bits = {2: 8, 3: 16, 4: 24, 5: 32}[encoding]
        frame = channels * bits // 8
        x = _int_pcm_to_float(body[: len(body) // frame * frame], bits, big_endian=True)
    elif encoding in (6, 7):
        item = 4 if encoding == 6 else 8
        frame = channels * item
        x = _float_payload(body[: len(body) // frame * frame], ">f4" if item == 4 else ">f8")
    else:
        raise AudioDecodeError(f"unsupported AU encoding {encoding}")
    if channels > 1:
        x = x.reshape(-1, channels)
    return x, int(rate)


_LOSSY_MAGIC = ((b"OggS", "Ogg (Vorbis / Opus / Ogg-FLAC)"), (b"\xff\xfb", "MP3"), (b"\xff\xf3", "MP3"), (b"\xff\xf2", "MP3"))


def decode_bytes(data: bytes):
    """Decode one file's bytes by container magic -> (float32 [frames] or [frames, channels], sample_rate).
    Whatever a parser trips over in damaged bytes (a short header, a size that does not divide, an absurd count) is
    a deterministic verdict on those bytes and leaves as ``AudioDecodeError`` -- never as an exception the caller's
    retry loop would take for a transient failure."""
    try:
        return _decode_bytes(data)
    except AudioDecodeError:
        raise
    except MemoryError:
        raise
    except (struct.error, ValueError, IndexError, OverflowError, ZeroDivisionError, TypeError) as err:
        raise AudioDecodeError(f"damaged audio file ({type(err).__name__}: {err})") from err


def _decode_bytes(data: bytes):
    head = bytes(data[:12])
    if head[:4] in (b"RIFF", b"RF64", b"BW64") and head[8:12] == b"WAVE":
        return _parse_wave(data)
    if head[:4] == b"fLaC":
        return _parse_flac(data)
    if head[:4] == b"FORM" and head[8:12] in (b"AIFF", b"AIFC"):
        return _parse_aiff(data)
    if head[:4] == b".snd":
        return _parse_au(data)
    if head[:3] == b"ID3":   # a tag in front of FLAC (decoded) or MP3 (not)
        size = ((head[6] & 0x7F) << 21) | ((head[7] & 0x7F) << 14) | ((head[8] & 0x7F) << 7) | (head[9] & 0x7F)
        at = 10 + size + (10 if head[5] & 0x10 else 0)
        if bytes(data[at : at + 4]) == b"fLaC":
            return _parse_flac(data)
        raise AudioDecodeError("unsupported container (MP3: a lossy codec this package does not decode)")
    for magic, name in _LOSSY_MAGIC:
        if head.startswith(magic):
            raise AudioDecodeError(f"unsupported container ({name}: a lossy codec this package does not decode)")
    raise AudioDecodeError("not a RIFF/WAVE, FLAC, AIFF or AU file")


def _parse_wave(data: bytes):
    """RIFF/WAVE (and RF64/BW64) parser -> (float32 array [frames] or [frames, channels], sample_rate)."""
    if len(data) < 12 or data[:4] not in (b"RIFF", b"RF64", b"BW64") or data[8:12] != b"WAVE":
        raise AudioDecodeError("not a RIFF/WAVE file")
    rf64 = data[:4] != b"RIFF"
    pos, fmt, payload, data_size64 = 12, None, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos : pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        if tag == b"ds64" and rf64 and size >= 24:   # 64-bit sizes of RF64: riff size, data size, sample count
            data_size64 = struct.unpack_from("<Q", data, pos + 16)[0]
        if tag == b"data" and size == 0xFFFFFFFF and data_size64 is not None:
            size = data_size64
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
        x = _float_payload(raw, "<f8")
    elif code == 6 and bits == 8:
        x = _companded_to_float(raw, _ALAW16)
    elif code == 7 and bits == 8:
        x = _companded_to_float(raw, _ULAW16)
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


class _DecodedFiles:
    """The last few files decoded for SEGMENT reads, kept whole: a manifest that cuts one long recording into
    thousands of utterances (``start_seconds`` / ``duration_seconds`` per utterance, data/manifest.py:175-230) asks for
    the same file again and again, usually in order.  The reference seeks inside the file (``librosa.load(offset=,
    duration=)``, audio_utils.py:104-109); decoding the whole file per utterance instead would be quadratic.  Keyed by
    (path, size, mtime); bounded by bytes; concurrent requests for one file wait for the first decode."""

    def __init__(self, max_bytes: int = 1 << 30, max_files: int = 4) -> None:
        self.max_bytes, self.max_files = max_bytes, max_files
        self._lock = threading.Lock()
        self._done: dict[tuple, tuple] = {}          # key -> (samples, rate), insertion order = age
        self._pending: dict[tuple, threading.Event] = {}

    def get(self, path: Path):
        st = path.stat()
        key = (str(path), st.st_size, st.st_mtime_ns)
        while True:
            with self._lock:
                hit = self._done.pop(key, None)
                if hit is not None:
                    self._done[key] = hit              # youngest again
                    return hit
                waiter = self._pending.get(key)
                if waiter is None:
                    self._pending[key] = threading.Event()
                    break
            waiter.wait()
        try:
            value = decode_bytes(path.read_bytes())
            with self._lock:
                self._done[key] = value
                total = sum(v[0].nbytes for v in self._done.values())
                while len(self._done) > 1 and (total > self.max_bytes or len(self._done) > self.max_files):
                    oldest = next(iter(self._done))
                    total -= self._done.pop(oldest)[0].nbytes
            return value
        finally:
            with self._lock:
                self._pending.pop(key).set()

    def clear(self) -> None:
        with self._lock:
            self._done.clear()


_decoded_files = _DecodedFiles()


def decode_file(path: Path, *, start_seconds: float = 0.0, duration_seconds: float | None = None):
    """Decode a file (or a segment of it: frame ``int(start*sr)``, ``int(duration*sr)`` frames)."""
    if float(start_seconds) == 0.0 and duration_seconds is None:
        return decode_bytes(path.read_bytes())
    samples, rate = _decoded_files.get(path)
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
            # it.  Name the container / encoding at once (module docstring: what is decoded here).
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
