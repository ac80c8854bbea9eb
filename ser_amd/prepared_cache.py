"""Prepared-feature cache: the ``.npz`` file that ``ser --train --prepare-only`` publishes and
``ser --train --prepared-plan`` consumes, so GPU-built feature matrices
can feed the reference's trainer.

Written from the FORMAT, which is what the reference's payload functions define
(``ser/_internal/models/training_orchestration.py:842-973``; file name from
``training_readiness.py:880-882``, cache namespace ``"fast_features"``):

=================  =======================================================================
entry              content
=================  =======================================================================
``x_train``        float64 ``[n_train, D]``
``x_dev``          float64 ``[n_dev, D]``   (may have 0 rows)
``x_test``         float64 ``[n_test, D]``
``y_train`` ...    numpy unicode ``[n]``, one non-blank label per row of the matching matrix
``metadata_json``  0-d numpy unicode: compact JSON object with sorted keys
=================  =======================================================================

compressed ``.npz`` (``np.savez_compressed``), no pickled objects, replaced atomically.  The
reference's reader additionally checks the file against its digest-protected plan ledger
(governance, out of scope); here ``feature_dtype`` / ``feature_shape`` stand in for the plan.

The writer produces the same container -- a zip archive of deflated ``.npy`` members that ``np.load`` and the
reference's reader open -- but deflates it with a team of threads (:func:`write_npz_deflated`): ``np.savez_compressed``
is one thread of zlib at level 6, 12 s for the 69 MB float64 matrix of a 46 080-file corpus that the GPU featurises
in 62 ms.

The module is table-driven: :data:`SCHEMA` lists the entries once and both directions walk it.
"""

from __future__ import annotations

import contextlib
import io
import json
import os
import struct
import tempfile
import time
import zlib
from collections.abc import Mapping, Sequence
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import NamedTuple

import numpy as np
from numpy.typing import NDArray

DEFAULT_FILE_NAME = "prepared-training-fast.npz"
PARTITIONS = ("train", "dev", "test")
#: entry name -> kind; the order is the order entries are written in
SCHEMA: tuple[tuple[str, str], ...] = (
    *((f"x_{p}", "matrix") for p in PARTITIONS),
    *((f"y_{p}", "labels") for p in PARTITIONS),
    ("metadata_json", "json"),
)
PAYLOAD_KEYS = frozenset(name for name, _ in SCHEMA)


class PreparedPlanError(ValueError):
    """The payload file is missing, malformed or inconsistent with what was planned."""


class PreparedFeaturePayload(NamedTuple):
    x_train: NDArray[np.float64]
    x_dev: NDArray[np.float64]
    x_test: NDArray[np.float64]
    y_train: list[str]
    y_dev: list[str]
    y_test: list[str]
    metadata: dict


def serialize_metadata(value):
    """Plan metadata as plain JSON types: mappings -> dicts with string keys, paths -> strings,
    sequences -> lists; anything else that JSON cannot hold is a ``TypeError``."""
    if value is None or isinstance(value, (bool, int, float, str)):
        return value
    if isinstance(value, os.PathLike):
        return os.fspath(value)
    if isinstance(value, Mapping):
        return {str(key): serialize_metadata(item) for key, item in value.items()}
    if isinstance(value, (list, tuple)):
        return [serialize_metadata(item) for item in value]
    raise TypeError(f"Unsupported prepared metadata type: {type(value).__name__}")


def _encode(kind: str, value) -> np.ndarray:
    if kind == "matrix":
        return np.asarray(value, dtype=np.float64)
    if kind == "labels":
        return np.asarray([str(label) for label in value], dtype=np.str_)
    return np.asarray(json.dumps(serialize_metadata(value), sort_keys=True, separators=(",", ":")), dtype=np.str_)


@contextlib.contextmanager
def _replacing(destination: Path):
    """Yield a binary handle on a sibling temp file; on a clean exit its bytes are flushed to disk
    and renamed over ``destination`` (readers see the old file or the new one, never a torso); on
    an exception the temp file is removed and ``destination`` is untouched."""
    destination.parent.mkdir(parents=True, exist_ok=True)
    handle = tempfile.NamedTemporaryFile("wb", dir=destination.parent, prefix=f".{destination.name}.",
                                         suffix=".part", delete=False)
    try:
        with handle:
            yield handle
            handle.flush()
            os.fsync(handle.fileno())
        os.replace(handle.name, destination)
    except BaseException:
        with contextlib.suppress(FileNotFoundError):
            os.unlink(handle.name)
        raise


#: bytes of a member deflated as one independent piece (pieces are joined at full-flush points into ONE deflate stream)
DEFLATE_PIECE_BYTES = 1 << 20
#: zlib level of the payload file: 1 (23 MB/s and thread, 0.60 of the size on float32 values widened to float64) where
#: numpy's writer uses 6 (5.6 MB/s, 0.55); ``SERHIP_CACHE_DEFLATE_LEVEL`` overrides
DEFAULT_DEFLATE_LEVEL = 1
_ZIP32_LIMIT = 0xFFFFFFFF


def _deflate_threads() -> int:
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:  # pragma: no cover
        return max(1, min(16, os.cpu_count() or 1))


def _npy_pieces(arr: np.ndarray, piece_bytes: int) -> list:
    """The bytes of ``np.save(arr)`` as a list of buffers of at most ``piece_bytes`` (the header rides in front of the
    first one); a large matrix is sliced in place, never copied."""
    arr = np.asarray(arr)
    if arr.dtype.hasobject:
        raise ValueError("object arrays are not written (no pickles in a payload file)")
    if arr.ndim == 0 or arr.dtype.kind in "US" or arr.nbytes <= piece_bytes or not arr.flags.c_contiguous:
        buf = io.BytesIO()
        np.lib.format.write_array(buf, arr, allow_pickle=False)   # (numpy's own layout choices: Fortran order is kept)
        whole = memoryview(buf.getvalue())
        return [whole[i:i + piece_bytes] for i in range(0, max(1, len(whole)), piece_bytes)]
    head = io.BytesIO()
    np.lib.format.write_array_header_1_0(head, np.lib.format.header_data_from_array_1_0(arr))
    data = memoryview(arr).cast("B")
    first = piece_bytes - len(head.getvalue()) % piece_bytes
    pieces = [head.getvalue() + bytes(data[:first])]
    pieces += [data[i:i + piece_bytes] for i in range(first, len(data), piece_bytes)]
    return pieces


def write_npz_deflated(handle, arrays: Mapping[str, np.ndarray], *, level: int | None = None, threads: int | None = None,
                       piece_bytes: int | None = None) -> None:
    """``np.savez_compressed(handle, **arrays)`` with the deflating done by ``threads`` threads: every member's ``.npy``
    image is cut into pieces, each piece is deflated on its own (zlib releases the GIL) and ended with a full flush, and
    the pieces in order ARE one raw deflate stream -- the member any zip reader expects.  CRC-32 of each member runs
    beside the deflating.  Members or archives of 4 GiB and more (zip64) are left to ``np.savez_compressed`` itself."""
    if level is None:
        level = int(os.environ.get("SERHIP_CACHE_DEFLATE_LEVEL", DEFAULT_DEFLATE_LEVEL))
    piece_bytes = int(piece_bytes or DEFLATE_PIECE_BYTES)
    threads = int(threads or _deflate_threads())
    prepared = {name: np.asarray(value) for name, value in arrays.items()}
    if sum(a.nbytes for a in prepared.values()) + 4096 * len(prepared) >= _ZIP32_LIMIT or len(prepared) >= 0xFFFF:
        np.savez_compressed(handle, **prepared)
        return
    members = [(f"{name}.npy".encode("utf-8"), _npy_pieces(arr, piece_bytes)) for name, arr in prepared.items()]

    def deflate(job):
        piece, last = job
        comp = zlib.compressobj(level, zlib.DEFLATED, -15)
        return comp.compress(piece) + comp.flush(zlib.Z_FINISH if last else zlib.Z_FULL_FLUSH)

    def checksum(pieces):
        crc = 0
        for piece in pieces:
            crc = zlib.crc32(piece, crc)
        return crc & 0xFFFFFFFF

    now = time.localtime()
    dos_date = (max(now.tm_year, 1980) - 1980) << 9 | now.tm_mon << 5 | now.tm_mday
    dos_time = now.tm_hour << 11 | now.tm_min << 5 | now.tm_sec // 2
    directory = []
    start = handle.tell()
    with ThreadPoolExecutor(max_workers=threads, thread_name_prefix="serhip-deflate") as team:
        crcs = [team.submit(checksum, pieces) for _, pieces in members]
        streams = [team.map(deflate, [(piece, k == len(pieces) - 1) for k, piece in enumerate(pieces)]) for _, pieces in members]
        for (name, pieces), crc_job, stream in zip(members, crcs, streams):
            deflated = list(stream)
            crc = crc_job.result()
            c_size, u_size, at = sum(map(len, deflated)), sum(map(len, pieces)), handle.tell() - start
            fixed = struct.pack("<HHHHHIIIHH", 20, 0, 8, dos_time, dos_date, crc, c_size, u_size, len(name), 0)
            handle.write(b"PK\x03\x04" + fixed + name)
            for block in deflated:
                handle.write(block)
            directory.append(b"PK\x01\x02" + struct.pack("<H", 20 | 3 << 8) + fixed + struct.pack("<HHHII", 0, 0, 0, 0o600 << 16, at)
                             + name)
    dir_at = handle.tell() - start
    for entry in directory:
        handle.write(entry)
    dir_size = handle.tell() - start - dir_at
    handle.write(b"PK\x05\x06" + struct.pack("<HHHHIIH", 0, 0, len(directory), len(directory), dir_size, dir_at, 0))


def default_prepared_payload_path(settings) -> Path:
    """Where the reference publishes the fast profile's payload: ``settings.tmp_folder /
    "prepared-training-fast.npz"`` (training_readiness.py:880-882 with the profile name this package
    implements)."""
    folder = getattr(settings, "tmp_folder", None)
    if folder is None:
        raise ValueError("write_prepared_feature_payload needs `path` or settings with a `tmp_folder`.")
    return Path(folder) / DEFAULT_FILE_NAME


def write_prepared_feature_payload(path=None, *, settings=None, x_train, x_dev, x_test, y_train: Sequence[str],
                                   y_dev: Sequence[str], y_test: Sequence[str],
                                   metadata: Mapping[str, object]) -> Path:
    """Publish one payload file; returns its path.  Keyword surface of the reference's writer
    (training_orchestration.py:842-853: ``settings``, the six arrays, ``metadata``, ``path=None``): without
    ``path`` the file goes to the reference's default location under ``settings.tmp_folder``.  ``path`` is
    also accepted as the first positional argument."""
    given = {"x_train": x_train, "x_dev": x_dev, "x_test": x_test, "y_train": y_train, "y_dev": y_dev,
             "y_test": y_test, "metadata_json": metadata}
    arrays = {name: _encode(kind, given[name]) for name, kind in SCHEMA}
    destination = Path(path) if path is not None else default_prepared_payload_path(settings)
    with _replacing(destination) as handle:
        write_npz_deflated(handle, arrays)   # the container of np.savez_compressed, deflated by a team of threads
    return destination


def _decode_matrix(name: str, arr: np.ndarray, feature_dtype: str) -> NDArray[np.float64]:
    if arr.dtype.name != feature_dtype:
        raise PreparedPlanError(f"Prepared payload entry {name} has dtype {arr.dtype.name}, the plan says {feature_dtype}.")
    if arr.ndim != 2:
        raise PreparedPlanError(f"Prepared payload entry {name} has {arr.ndim} dimensions, expected a 2-D matrix.")
    out = np.asarray(arr, dtype=np.float64)
    if not np.isfinite(out).all():
        raise PreparedPlanError(f"Prepared payload entry {name} contains non-finite values.")
    return out


def _decode_labels(name: str, arr: np.ndarray) -> list[str]:
    if arr.ndim != 1 or arr.dtype.kind not in "US":
        raise PreparedPlanError(f"Prepared payload entry {name} must be a 1-D array of strings.")
    labels = [str(item) for item in arr.tolist()]
    blank = sum(1 for label in labels if not label.strip())
    if blank:
        raise PreparedPlanError(f"Prepared payload entry {name} holds {blank} blank label(s).")
    return labels


def _decode_json(name: str, arr: np.ndarray) -> dict:
    if arr.ndim != 0 or arr.dtype.kind not in "US":
        raise PreparedPlanError(f"Prepared payload entry {name} must be one scalar JSON string.")
    try:
        parsed = json.loads(str(arr[()]))
    except json.JSONDecodeError as error:
        raise PreparedPlanError(f"Prepared payload entry {name} is not valid JSON: {error}") from error
    if not isinstance(parsed, dict):
        raise PreparedPlanError(f"Prepared payload entry {name} must decode to a JSON object.")
    return parsed


def read_prepared_feature_payload(path, *, feature_dtype: str = "float64",
                                  feature_shape: tuple[int, int] | None = None) -> PreparedFeaturePayload:
    """Load and validate one payload.  ``feature_shape`` is the planned ``(total rows, D)``."""
    try:
        with np.load(Path(path), allow_pickle=False) as archive:
            stored = set(archive.files)
            if stored != PAYLOAD_KEYS:
                problem = ", ".join(sorted(stored ^ PAYLOAD_KEYS))
                raise PreparedPlanError(f"Prepared payload entries differ from the format (missing or unknown: {problem}).")
            raw = {name: np.asarray(archive[name]) for name, _ in SCHEMA}
    except PreparedPlanError:
        raise
    except Exception as error:   # unreadable zip, object arrays refused by allow_pickle=False, I/O errors
        raise PreparedPlanError(f"Prepared payload {path} cannot be read: {error}") from error

    decoded: dict[str, object] = {}
    for name, kind in SCHEMA:
        if kind == "matrix":
            decoded[name] = _decode_matrix(name, raw[name], feature_dtype)
        elif kind == "labels":
            decoded[name] = _decode_labels(name, raw[name])
        else:
            decoded["metadata"] = _decode_json(name, raw[name])
    widths = {decoded[f"x_{p}"].shape[1] for p in PARTITIONS}
    rows = sum(decoded[f"x_{p}"].shape[0] for p in PARTITIONS)
    if len(widths) != 1:
        raise PreparedPlanError(f"Prepared feature matrices differ in width ({sorted(widths)}): not the planned shape.")
    if feature_shape is not None and (rows, widths.copy().pop()) != tuple(int(v) for v in feature_shape):
        raise PreparedPlanError(
            f"Prepared feature matrices hold {rows} x {widths.copy().pop()} values, the planned shape is "
            f"{tuple(feature_shape)}.")
    for p in PARTITIONS:
        if decoded[f"x_{p}"].shape[0] != len(decoded[f"y_{p}"]):
            raise PreparedPlanError(
                f"Prepared payload partition {p}: {decoded[f'x_{p}'].shape[0]} feature rows but "
                f"{len(decoded[f'y_{p}'])} labels.")
    return PreparedFeaturePayload(**decoded)
