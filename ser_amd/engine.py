"""Batched GPU feature engine: the host side of ``serhip_fast_features``.

One :class:`FeatureEngine` per (process, GPU).  Audio lives in HBM as a torch
tensor (torch is used for device memory and streams only); segments are
(offset, length) descriptors into it, so overlapping sliding windows need no
copies.  Every call goes through the C ABI; there is no CPU fallback.
"""

from __future__ import annotations

import ctypes
import os
import threading
from ctypes import POINTER, c_int32, c_int64, c_void_p

import numpy as np

from . import _native
from .types import FeatureFlags, require_accelerated

_engines: dict[tuple[int, int], "FeatureEngine"] = {}
_engines_lock = threading.Lock()


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError(
            "ser_amd needs an AMD GPU (torch.cuda.is_available() is False) -- there is no CPU fallback; "
            "the CPU restatement under oracle/ is test infrastructure only"
        )
    return torch


def get_engine(device_index: int | None = None) -> "FeatureEngine":
    """Process-local engine for ``device_index`` (default: torch's current device).

    Created lazily and keyed by PID so a forked ``multiprocessing`` worker builds
    its own HIP context instead of reusing the parent's.
    """
    torch = _torch()
    if device_index is None:
        device_index = torch.cuda.current_device()
    key = (os.getpid(), int(device_index))
    with _engines_lock:
        eng = _engines.get(key)
        if eng is None:
            eng = FeatureEngine(int(device_index))
            _engines[key] = eng
        return eng


def _as_i64(values) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(values, dtype=np.int64))


class FeatureEngine:
    """Owns one ``serhip_ctx`` and its plans; serialises calls with a lock."""

    def __init__(self, device_index: int = 0) -> None:
        self._torch = _torch()
        self.device_index = int(device_index)
        self.device = self._torch.device("cuda", self.device_index)
        with self._torch.cuda.device(self.device):
            self._ctx = _native.Context(self.device_index)
        self._lib = _native.load_library()
        self._lock = threading.Lock()

    # ------------------------------------------------------------------ buffers
    def to_device(self, audio) -> "object":
        """float32 contiguous 1-D (or 2-D for whisper) tensor on this engine's GPU."""
        torch = self._torch
        if isinstance(audio, torch.Tensor):
            t = audio.to(device=self.device, dtype=torch.float32)
        else:
            arr = np.ascontiguousarray(np.asarray(audio, dtype=np.float32))
            t = torch.from_numpy(arr).to(self.device)
        return t.contiguous()

    def to_device_staged(self, array: np.ndarray):
        """A C-contiguous numpy array of any fixed-size dtype -> a device tensor of the same shape and dtype.  Small
        arrays take torch's pageable copy; from ``STAGED_TO_HOST_BYTES`` up the bytes travel like the ingest pipeline's:
        the memcpy team fills the pinned ring chunk by chunk while the chunk before is on the wire (a 69 MB float64
        matrix: 2.3 ms pageable, 1.6 ms this way -- pageable uploads are far less of a detour than pageable downloads,
        ``to_host_float64``; ``scripts/experiments/staged_upload.py``)."""
        torch = self._torch
        array = np.ascontiguousarray(array)
        nbytes = int(array.nbytes)
        if nbytes < self.STAGED_TO_HOST_BYTES:
            return torch.from_numpy(array).to(self.device)
        from .ingest import PINNED_BUFFERS, ingest_buffers, parallel_memcpy

        out = torch.empty(tuple(array.shape), dtype=torch.from_numpy(array[:0]).dtype, device=self.device)
        flat = out.view(torch.uint8).reshape(-1)
        buffers = ingest_buffers(self)
        with buffers.lock, torch.cuda.device(self.device):
            buffers.ensure(min(nbytes, self.STAGED_CHUNK_BYTES))
            pinned = buffers.pinned
            chunk = min(self.STAGED_CHUNK_BYTES, *(int(p.numel()) for p in pinned))
            stream = torch.cuda.current_stream(self.device)
            events: dict[int, object] = {}
            try:
                for k in range(-(-nbytes // chunk)):
                    lo, hi = k * chunk, min(nbytes, (k + 1) * chunk)
                    if k - PINNED_BUFFERS in events:
                        events.pop(k - PINNED_BUFFERS).synchronize()   # the copy that last read this pinned buffer
                    buf = pinned[k % PINNED_BUFFERS]
                    parallel_memcpy(buf.data_ptr(), array.ctypes.data + lo, hi - lo)
                    flat[lo:hi].copy_(buf[: hi - lo], non_blocking=True)
                    events[k] = torch.cuda.Event()
                    events[k].record(stream)
            finally:
                for ev in events.values():   # the ring is shared: nothing may still be reading it when we leave
                    ev.synchronize()
        return out

    def _stream(self) -> c_void_p:
        return c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    #: device matrices at least this large come back through pinned staging (below)
    STAGED_TO_HOST_BYTES = 8 << 20
    STAGED_CHUNK_BYTES = 8 << 20

    def to_host_float64(self, matrix) -> np.ndarray:
        """``[n, D]`` device tensor -> a freshly allocated float64 numpy array (the dtype of the reference's API,
        SURVEY 8b).  The widening runs on the device (exact).  A large matrix is not handed to a pageable
        ``.cpu()`` (the driver stages that through small bounce buffers: 69 MB took 16.6 ms, a fifth of a
        46 080-file ingest): it is copied chunk by chunk into the ingest ring's pinned buffers and moved from there
        into the numpy array by the memcpy team while the next chunk is in flight."""
        torch = self._torch
        wide = matrix.to(device=self.device, dtype=torch.float64).contiguous()
        nbytes = wide.numel() * 8
        if nbytes < self.STAGED_TO_HOST_BYTES:
            return wide.cpu().numpy()
        from .ingest import ingest_buffers

        out = np.empty(tuple(wide.shape), dtype=np.float64)
        buffers = ingest_buffers(self)
        with buffers.lock:
            return self._staged_to_host(buffers, wide, out, nbytes)

    def _staged_to_host(self, buffers, wide, out, nbytes: int) -> np.ndarray:
        from .ingest import PINNED_BUFFERS, parallel_memcpy

        torch = self._torch
        buffers.ensure(min(nbytes, self.STAGED_CHUNK_BYTES))
        pinned = buffers.pinned
        chunk = min(self.STAGED_CHUNK_BYTES, *(int(p.numel()) for p in pinned))
        flat = wide.view(torch.uint8).reshape(-1)
        n_chunks = -(-nbytes // chunk)
        stream = torch.cuda.current_stream(self.device)
        events: dict[int, object] = {}

        def issue(k: int) -> None:
            lo, hi = k * chunk, min(nbytes, (k + 1) * chunk)
            pinned[k % PINNED_BUFFERS][: hi - lo].copy_(flat[lo:hi], non_blocking=True)
            events[k] = torch.cuda.Event()
            events[k].record(stream)

        with torch.cuda.device(self.device):
            try:
                for k in range(min(PINNED_BUFFERS, n_chunks)):
                    issue(k)
                for k in range(n_chunks):
                    events.pop(k).synchronize()
                    lo, hi = k * chunk, min(nbytes, (k + 1) * chunk)
                    parallel_memcpy(out.ctypes.data + lo, pinned[k % PINNED_BUFFERS].data_ptr(), hi - lo)
                    if k + PINNED_BUFFERS < n_chunks:
                        issue(k + PINNED_BUFFERS)   # into the buffer that has just been drained
            finally:
                for ev in events.values():          # on an error: no copy may still be writing the shared ring
                    ev.synchronize()
        return out

    # ------------------------------------------------------------------ hot path
    def features_into(self, audio_dev, seg_offset, seg_len, sample_rate: int, flags: FeatureFlags, out, *,
                      tuning=None, plan=None):
        """Allocation-free form (``serhip_fast_features_noalloc``): ``audio_dev`` and ``out`` (and ``tuning``)
        are float32 device tensors the caller already owns, ``[n_samples]`` and ``[n_seg, D]``; the library's
        scratch is a static device array, so nothing is allocated on the GPU by the call.  Bounded: 256
        segments / 6 144 frame slots per call (``include/serhip.h``).  ``plan`` may be passed to skip the
        plan lookup (it must have been created before entering a region that forbids allocation)."""
        require_accelerated(flags)
        torch = self._torch
        off, length = _as_i64(seg_offset), _as_i64(seg_len)
        n_seg = int(off.shape[0])
        for t, shape in ((audio_dev, None), (out, (n_seg, flags.width()))):
            if not isinstance(t, torch.Tensor) or t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError("features_into takes contiguous float32 tensors on the engine's device")
            if shape is not None and tuple(t.shape) != shape:
                raise ValueError(f"out must be {shape}, got {tuple(t.shape)}")
        with self._torch.cuda.device(self.device), self._lock:
            if plan is None:
                plan = self._ctx.plan(int(sample_rate), flags.native_bits())
            _native.check(
                self._lib.serhip_fast_features_noalloc(
                    plan.handle, c_void_p(audio_dev.data_ptr()), int(audio_dev.numel()),
                    off.ctypes.data_as(POINTER(c_int64)), length.ctypes.data_as(POINTER(c_int64)), n_seg,
                    c_void_p(out.data_ptr()), c_void_p(tuning.data_ptr()) if tuning is not None else None,
                    self._stream(),
                )
            )
        return out

    #: host buffers at least this large go to the GPU through the staged, overlapped path (ser_amd.ingest)
    STREAM_FROM_HOST_BYTES = 32 << 20

    def features(self, audio, seg_offset, seg_len, sample_rate: int, flags: FeatureFlags, *,
                 return_tuning: bool = False, out=None, check_finite: bool = False):
        """Feature matrix ``[n_seg, D]`` (float32, on device) for segments of ``audio``.

        ``audio`` is a 1-D float32 numpy array or torch tensor (moved to the GPU if
        needed); ``seg_offset`` / ``seg_len`` are host integer sequences.  ``out`` may name the
        contiguous float32 ``[n_seg, D]`` device tensor (e.g. a row range of a larger matrix) the kernels
        write into.  A large numpy buffer whose segments come in non-decreasing order is not copied whole
        first: it is staged chunk by chunk while the kernels run (``ingest.features_from_host``).
        ``check_finite`` (numpy ``audio`` only): raise the reference's ``ValueError`` when a sample of the buffer is
        not finite (dsp.py:93-95) -- tested by the staging threads as they copy on the staged path, by one BLAS pass
        (``dsp.samples_all_finite``) otherwise.
        """
        require_accelerated(flags)
        torch = self._torch
        off = _as_i64(seg_offset)
        length = _as_i64(seg_len)
        if off.shape != length.shape or off.ndim != 1:
            raise ValueError("seg_offset and seg_len must be 1-D and the same length")
        n_seg = int(off.shape[0])
        dim = flags.width()
        if out is not None and (not isinstance(out, torch.Tensor) or out.device != self.device or out.dtype != torch.float32
                                or not out.is_contiguous() or tuple(out.shape) != (n_seg, dim)):
            raise ValueError(f"out must be a contiguous float32 ({n_seg}, {dim}) tensor on {self.device}")
        if (not return_tuning and isinstance(audio, np.ndarray) and audio.ndim == 1 and n_seg > 1
                and audio.nbytes >= self.STREAM_FROM_HOST_BYTES and os.environ.get("SERHIP_INGEST", "pipelined") != "sequential"
                and bool((np.diff(off) >= 0).all())):
            from .ingest import features_from_host

            return features_from_host(self, audio, off, length, int(sample_rate), flags, out=out, check_finite=check_finite)
        if check_finite and isinstance(audio, np.ndarray):
            from .dsp import samples_all_finite

            if not samples_all_finite(np.asarray(audio, dtype=np.float32)):
                raise ValueError("Audio buffer is not finite everywhere.")
        audio_dev = self.to_device(audio)
        if audio_dev.dim() != 1:
            raise ValueError("audio must be a 1-D sample buffer")
        with self._torch.cuda.device(self.device), self._lock:
            plan = self._ctx.plan(int(sample_rate), flags.native_bits())
            if out is None:
                out = torch.empty((n_seg, dim), dtype=torch.float32, device=self.device)
            tuning = torch.empty((n_seg,), dtype=torch.float32, device=self.device) if return_tuning else None
            if n_seg and dim:
                _native.check(
                    self._lib.serhip_fast_features(
                        plan.handle,
                        c_void_p(audio_dev.data_ptr()),
                        int(audio_dev.numel()),
                        off.ctypes.data_as(POINTER(c_int64)),
                        length.ctypes.data_as(POINTER(c_int64)),
                        n_seg,
                        c_void_p(out.data_ptr()),
                        c_void_p(tuning.data_ptr()) if tuning is not None else None,
                        self._stream(),
                    )
                )
            # the kernels read audio_dev asynchronously: keep it alive until they are enqueued
            # on the same stream the caller will synchronise (torch's caching allocator is
            # stream-ordered, so releasing the tensor after this point is safe)
        if return_tuning:
            return out, tuning
        return out

    def tuning_from_peaks(self, frames, sample_rate: int = 16000) -> float:
        """Diagnostic: the device tuning estimate for per-frame lists of ``(pitch_hz, magnitude)`` peaks
        (``serhip_tuning_from_peaks``); returns the float32 estimate."""
        from ctypes import c_float

        counts = np.ascontiguousarray([len(f) for f in frames], dtype=np.int32)
        flat = np.ascontiguousarray([pm for f in frames for pm in f], dtype=np.float32).reshape(-1, 2)
        out = c_float(0.0)
        with self._torch.cuda.device(self.device), self._lock:
            plan = self._ctx.plan(int(sample_rate), FeatureFlags(tonnetz=False).native_bits())
            _native.check(
                self._lib.serhip_tuning_from_peaks(
                    plan.handle, flat.ctypes.data_as(POINTER(c_float)) if flat.size else None,
                    counts.ctypes.data_as(POINTER(c_int32)), int(counts.size), ctypes.byref(out),
                )
            )
        return np.float32(out.value)

    # ------------------------------------------------------------------ audio preparation
    def prepare_audio(self, raw, sample_format: str, clip_raw_offset, clip_frames, clip_channels):
        """Device form of ``_prepare_audio_buffer`` for a batch of raw clips.

        ``raw`` holds the interleaved sample payloads (bytes-like, uint8 numpy array or uint8 torch
        tensor); ``sample_format`` is ``"s16"`` (little-endian PCM16) or ``"f32"``.  Returns
        ``(audio, offsets, lengths)``: one float32 device buffer with every clip mono,
        scrubbed and peak-normalised, and the int64 host descriptors ``features`` takes.
        """
        torch = self._torch
        fmt = {"s16": 0, "f32": 1}.get(sample_format)
        if fmt is None:
            raise ValueError("sample_format must be 's16' or 'f32'")
        if isinstance(raw, torch.Tensor):
            raw_dev = raw.to(device=self.device).contiguous().view(torch.uint8).reshape(-1)
        else:
            raw_dev = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()
                                       if not isinstance(raw, np.ndarray)
                                       else np.ascontiguousarray(raw).view(np.uint8).reshape(-1)).to(self.device)
        off = _as_i64(clip_raw_offset)
        frames = _as_i64(clip_frames)
        chans = np.ascontiguousarray(np.asarray(clip_channels, dtype=np.int32))
        if not (off.shape == frames.shape == chans.shape) or off.ndim != 1:
            raise ValueError("clip descriptors must be 1-D and the same length")
        n = int(off.shape[0])
        out_off = np.zeros(n, dtype=np.int64)
        if n:
            np.cumsum(frames[:-1], out=out_off[1:])
        total = int(frames.sum())
        with torch.cuda.device(self.device), self._lock:
            out = torch.empty((total,), dtype=torch.float32, device=self.device)
            if n:
                _native.check(
                    self._lib.serhip_prepare_audio(
                        self._ctx.handle,
                        c_void_p(raw_dev.data_ptr()),
                        int(raw_dev.numel()),
                        fmt,
                        off.ctypes.data_as(POINTER(c_int64)),
                        frames.ctypes.data_as(POINTER(c_int64)),
                        chans.ctypes.data_as(POINTER(c_int32)),
                        out_off.ctypes.data_as(POINTER(c_int64)),
                        n,
                        c_void_p(out.data_ptr()),
                        total,
                        self._stream(),
                    )
                )
        return out, out_off, frames

    # ------------------------------------------------------------------ pooling
    def pool_windows(self, embeddings, win_lo, win_hi, *, mode: int):
        """Pool rows ``[lo, hi)`` of ``embeddings`` per window; returns a float64 device tensor."""
        torch = self._torch
        emb = self.to_device(embeddings)
        if emb.dim() != 2:
            raise ValueError("embeddings must be 2-D (frames, features)")
        lo = np.ascontiguousarray(np.asarray(win_lo, dtype=np.int32))
        hi = np.ascontiguousarray(np.asarray(win_hi, dtype=np.int32))
        n_win, (n_frames, dim) = int(lo.shape[0]), emb.shape
        width = dim * 2 if mode == 0 else dim
        with self._torch.cuda.device(self.device), self._lock:
            out = torch.empty((n_win, width), dtype=torch.float64, device=self.device)
            for start in range(0, n_win, 65535):
                stop = min(n_win, start + 65535)
                _native.check(
                    self._lib.serhip_pool_windows(
                        self._ctx.handle,
                        c_void_p(emb.data_ptr()),
                        int(n_frames),
                        int(dim),
                        lo[start:stop].ctypes.data_as(POINTER(c_int32)),
                        hi[start:stop].ctypes.data_as(POINTER(c_int32)),
                        stop - start,
                        int(mode),
                        c_void_p(out[start:stop].data_ptr()),
                        self._stream(),
                    )
                )
        return out

    # ------------------------------------------------------------------ whisper
    def whisper_logmel(self, audio, n_mels: int = 80):
        """``[n_clips, n_mels, n_samples // 160]`` float32 device tensor (clips are rows of ``audio``)."""
        torch = self._torch
        a = self.to_device(audio)
        if a.dim() == 1:
            a = a.unsqueeze(0)
        if a.dim() != 2:
            raise ValueError("audio must be [n_samples] or [n_clips, n_samples]")
        n_clips, n_samples = int(a.shape[0]), int(a.shape[1])
        n_frames = n_samples // 160
        with self._torch.cuda.device(self.device), self._lock:
            out = torch.empty((n_clips, int(n_mels), n_frames), dtype=torch.float32, device=self.device)
            _native.check(
                self._lib.serhip_whisper_logmel(
                    self._ctx.handle, c_void_p(a.data_ptr()), n_clips, n_samples, int(n_mels),
                    c_void_p(out.data_ptr()), self._stream(),
                )
            )
        return out

    def close(self) -> None:
        self._ctx.close()
