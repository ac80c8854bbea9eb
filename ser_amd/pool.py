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
    """``(lo, hi)`` int32 arrays of the frame range each window overlaps (validates like the reference)."""
    ranges = [overlap_frame_range(encoded, w) for w in windows]
    lo = np.asarray([r[0] for r in ranges], dtype=np.int32)
    hi = np.asarray([r[1] for r in ranges], dtype=np.int32)
    return lo, hi


def mean_std_pool(encoded: EncodedSequence, windows: Sequence[PoolingWindow]) -> NDArray[np.float64]:
    """``[len(windows), 2*D]`` float64: per-window mean and population std of the frames."""
    dim = int(encoded.embeddings.shape[1])
    if not windows:
        return np.empty((0, dim * 2), dtype=np.float64)
    lo, hi = window_frame_ranges(encoded, windows)
    out = get_engine().pool_windows(np.asarray(encoded.embeddings, dtype=np.float32), lo, hi, mode=0)
    return out.cpu().numpy()
