"""Multi-GPU sharding of the featurisation: one process per GPU, contiguous shards by
segment index, one all-gather (RCCL over xGMI; gloo in the CPU tests) that assembles
the training feature matrix on every rank.

The reference has no distributed code; its only fan-out is the ``multiprocessing.Pool``
over files of ``ser/_internal/data/data_loader.py:368-379``, which proves the units are
independent (``imap_unordered``).  Segments never exchange data, so the gather is the
single collective of the path.

Two forms of that gather: :func:`all_gather_rows` on ``torch.distributed`` (backend ``nccl`` = RCCL; gloo in the
CPU tests) and :class:`NativeRowGather` on the C ABI's own RCCL communicator (``serhip_comm_*``,
``serhip_allgather_rows``) for hosts that do not run ``torch.distributed``.
"""

from __future__ import annotations

import numpy as np


def shard_bounds(n_items: int, world_size: int) -> np.ndarray:
    """``world_size + 1`` boundaries of contiguous shards: rank r owns [b[r], b[r+1]).

    Every rank gets ``ceil(n / world)`` items except the trailing ranks, which may get
    fewer (or none) -- the gather pads to the common row count.
    """
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    per = -(-int(n_items) // int(world_size)) if n_items > 0 else 0
    return np.minimum(np.arange(world_size + 1, dtype=np.int64) * per, int(n_items))


def shard_range(n_items: int, rank: int, world_size: int) -> tuple[int, int]:
    b = shard_bounds(n_items, world_size)
    return int(b[rank]), int(b[rank + 1])


def all_gather_rows_async(local, n_total: int, *, group=None):
    """Non-blocking form for evenly divisible shards: returns ``(full, work)``; ``full`` is valid after
    ``work.wait()`` (``work`` is None when nothing is in flight: single process, or uneven shards, which
    take the blocking path because their padding rows have to be cut out afterwards)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return all_gather_rows(local, n_total, group=group), None
    world = dist.get_world_size(group)
    if int(local.shape[0]) * world != int(n_total):
        return all_gather_rows(local, n_total, group=group), None
    send = local.contiguous()
    full = torch.empty((int(n_total), int(local.shape[1])), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(full, send, group=group, async_op=True)
    return full, work


def all_gather_rows(local, n_total: int, *, group=None):
    """Gather row shards ``[rows_r, D]`` (torch tensors, same device/dtype on all ranks) into
    the full ``[n_total, D]`` matrix on every rank, in rank order.

    Shards are padded to ``ceil(n_total / world)`` rows so a single
    ``all_gather_into_tensor`` (one RCCL collective) moves them; padding rows are dropped.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if int(local.shape[0]) != int(n_total):
            raise ValueError("single-process gather: local rows must equal n_total")
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bounds = shard_bounds(n_total, world)
    per = int(bounds[1] - bounds[0]) if world > 0 else 0
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    if int(local.shape[0]) != hi - lo:
        raise ValueError(f"rank {rank}: expected {hi - lo} rows, got {int(local.shape[0])}")
    dim = int(local.shape[1])
    send = local
    if hi - lo != per:
        send = torch.zeros((per, dim), dtype=local.dtype, device=local.device)
        send[: hi - lo] = local
    send = send.contiguous()
    full = torch.empty((world * per, dim), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, send, group=group)
    if world * per == n_total:
        return full
    pieces = [full[r * per : r * per + int(bounds[r + 1] - bounds[r])] for r in range(world)]
    return torch.cat(pieces, dim=0)


COMM_ID_BYTES = 128


def new_comm_id() -> bytes:
    """The opaque id of a new native communicator (``serhip_comm_unique_id``).  Rank 0 creates it and hands
    the bytes to the other ranks by whatever transport the host has."""
    import ctypes

    from ._native import check, load_library

    buf = (ctypes.c_ubyte * COMM_ID_BYTES)()
    check(load_library().serhip_comm_unique_id(buf))
    return bytes(buf)


class NativeRowGather:
    """The fan-in of the feature shards on the C ABI's RCCL communicator: one instance per process / GPU.

    ``NativeRowGather(comm_id, rank, world)`` is collective (every rank constructs it with the same id);
    ``gather(local, rows_per_rank)`` returns the full ``[sum(rows_per_rank), D]`` float32 matrix on this
    rank's GPU, shards in rank order (uneven and empty shards allowed), enqueued on torch's current stream.
    """

    def __init__(self, comm_id: bytes, rank: int, world: int, *, device_index: int | None = None) -> None:
        import ctypes

        from ._native import check, load_library
        from .engine import get_engine

        if len(comm_id) != COMM_ID_BYTES:
            raise ValueError(f"comm_id must be {COMM_ID_BYTES} bytes")
        self._lib = load_library()
        self._engine = get_engine(device_index)
        self.rank, self.world = int(rank), int(world)
        handle = ctypes.c_void_p()
        buf = (ctypes.c_ubyte * COMM_ID_BYTES).from_buffer_copy(comm_id)
        check(self._lib.serhip_comm_create(self._engine._ctx.handle, buf, self.rank, self.world, ctypes.byref(handle)))
        self._handle = handle

    def gather(self, local, rows_per_rank):
        import ctypes

        import torch

        from ._native import check

        rows = np.ascontiguousarray(np.asarray(rows_per_rank, dtype=np.int64))
        if rows.shape != (self.world,):
            raise ValueError(f"rows_per_rank must have {self.world} entries")
        if local.dtype != torch.float32 or local.dim() != 2 or int(local.shape[0]) != int(rows[self.rank]):
            raise ValueError(f"rank {self.rank}: expected a float32 [{int(rows[self.rank])}, D] shard")
        local = local.contiguous()
        dim = int(local.shape[1])
        full = torch.empty((int(rows.sum()), dim), dtype=torch.float32, device=local.device)
        if full.numel() == 0:   # nothing to move on any rank (every rank sees the same rows_per_rank)
            return full
        stream = torch.cuda.current_stream(local.device).cuda_stream
        check(self._lib.serhip_allgather_rows(
            self._handle, ctypes.c_void_p(local.data_ptr()), rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
            dim, ctypes.c_void_p(full.data_ptr()), ctypes.c_void_p(stream)))
        return full

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self._lib.serhip_comm_destroy(self._handle)
            self._handle = None

    def __del__(self) -> None:  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass
