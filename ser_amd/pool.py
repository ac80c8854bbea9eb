"""Segment pooling: window generation on the host, statistics on the GPU.

``temporal_pooling_windows`` keeps the scalar walk of
``ser/_internal/pool/windowing.py:10-71``; ``mean_std_pool`` has the signature of
``ser/_internal/pool/stats_pool.py:15-43`` and runs ``serhip_pool_windows`` over
the contiguous frame range each window overlaps.
"""

from __future__ import annotations

from collections.abc import Sequence

import numpy as np
from numpy.typing import NDArray

from .engine import get_engine
from .types import EncodedSequence, PoolingWindow, overlap_frame_range

_EPS = 1e-9


def temporal_pooling_windows(
    encoded: EncodedSequence,
    *,
    window_size_seconds: float,
    window_stride_seconds: float,
) -> list[PoolingWindow]:
    """Deterministic temporal pooling windows that cover the encoded timeline."""
    for label, value in (("window_size_seconds", window_size_seconds),
                         ("window_stride_seconds", window_stride_seconds)):
        if value <= 0.0 or not np.isfinite(value):
            raise ValueError(f"{label} must be a positive finite float.")
    t_begin = float(encoded.frame_start_seconds[0])
    t_end = float(encoded.frame_end_seconds[-1])
    duration = t_end - t_begin
    if duration <= 0.0:
        raise ValueError("Encoded sequence duration must be positive.")
    span = min(window_size_seconds, duration)
    if np.isclose(span, duration):
        return [PoolingWindow(start_seconds=t_begin, end_seconds=t_end)]

    windows: list[PoolingWindow] = []
    cursor = t_begin
    while cursor + span <= t_end + _EPS:
        windows.append(PoolingWindow(start_seconds=cursor, end_seconds=min(t_end, cursor + span)))
        cursor += window_stride_seconds
    tail = PoolingWindow(start_seconds=max(t_begin, t_end - span), end_seconds=t_end)
    if not windows:
        return [tail]
    last = windows[-1]
    if last.end_seconds < t_end - _EPS and not (
        np.isclose(last.start_seconds, tail.start_seconds) and np.isclose(last.end_seconds, tail.end_seconds)
    ):
        windows.append(tail)
    return windows


def window_frame_ranges(encoded: EncodedSequence, windows: Sequence[PoolingWindow]):
    """``(lo, hi)`` int32 arrays of the frame range each window overlaps.

    All windows in two ``searchsorted`` calls; validates like the reference's per-window walk
    (``overlap_frame_mask``, repr/backend.py:94-110): the first offending window, in order, raises its
    error through :func:`overlap_frame_range`."""
    if len(windows) == 0:
        return np.empty(0, dtype=np.int32), np.empty(0, dtype=np.int32)
    t0, t1 = encoded.frame_start_seconds, encoded.frame_end_seconds
    ws = np.asarray([w.start_seconds for w in windows], dtype=np.float64)
    we = np.asarray([w.end_seconds for w in windows], dtype=np.float64)
    lo = np.searchsorted(t1, ws, side="right")   # first frame with end > window start
    hi = np.searchsorted(t0, we, side="left")    # first frame with start >= window end
    bad = (ws < float(t0[0])) | (we > float(t1[-1])) | (hi <= lo)
    if bad.any():
        overlap_frame_range(encoded, windows[int(np.argmax(bad))])   # raises the reference's message
    return lo.astype(np.int32), hi.astype(np.int32)


def mean_std_pool(encoded: EncodedSequence, windows: Sequence[PoolingWindow]) -> NDArray[np.float64]:
    """``[len(windows), 2*D]`` float64: per-window mean and population std of the frames."""
    dim = int(encoded.embeddings.shape[1])
    if not windows:
        return np.empty((0, dim * 2), dtype=np.float64)
    lo, hi = window_frame_ranges(encoded, windows)
    out = get_engine().pool_windows(np.asarray(encoded.embeddings, dtype=np.float32), lo, hi, mode=0)
    return out.cpu().numpy()
