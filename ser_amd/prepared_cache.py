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

The module is table-driven: :data:`SCHEMA` lists the entries once and both directions walk it.
"""

from __future__ import annotations

import contextlib
import json
import os
import tempfile
from collections.abc import Mapping, Sequence
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
        np.savez_compressed(handle, **arrays)
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
