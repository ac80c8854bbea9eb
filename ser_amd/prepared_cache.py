"""Prepared-feature cache: the ``.npz`` payload ``ser --train --prepare-only`` publishes and
``--prepared-plan`` consumes.

Same container as ``write_prepared_feature_payload`` / ``read_prepared_feature_payload``
(``ser/_internal/models/training_orchestration.py:842-973``): ``np.savez_compressed`` with keys
``x_train, x_dev, x_test`` (float64, 2-D), ``y_train, y_dev, y_test`` (``np.str_`` 1-D) and
``metadata_json`` (0-d string, compact sorted-key JSON); written to a temp file in the destination
directory, fsynced and ``os.replace``d; read with ``allow_pickle=False``.  The reference's reader also
cross-checks the payload against its digest-protected plan ledger (governance, out of scope); the
structural checks are reproduced here with the reference's messages.
"""

from __future__ import annotations

import json
import os
import tempfile
from collections.abc import Mapping, Sequence
from pathlib import Path
from typing import NamedTuple

import numpy as np
from numpy.typing import NDArray

PAYLOAD_KEYS = frozenset({"x_train", "x_dev", "x_test", "y_train", "y_dev", "y_test", "metadata_json"})
DEFAULT_FILE_NAME = "prepared-training-fast.npz"   # training_readiness.py:880-882, namespace "fast_features"


class PreparedPlanError(ValueError):
    """The payload is missing, malformed or inconsistent."""


class PreparedFeaturePayload(NamedTuple):
    x_train: NDArray[np.float64]
    x_dev: NDArray[np.float64]
    x_test: NDArray[np.float64]
    y_train: list[str]
    y_dev: list[str]
    y_test: list[str]
    metadata: dict


def serialize_metadata(value):
    """JSON-safe form of plan metadata (paths -> str, tuples -> lists), like the reference's helper."""
    if isinstance(value, Mapping):
        return {str(k): serialize_metadata(v) for k, v in value.items()}
    if isinstance(value, Path):
        return str(value)
    if isinstance(value, (tuple, list)):
        return [serialize_metadata(v) for v in value]
    if value is None or isinstance(value, (str, int, float, bool)):
        return value
    raise TypeError(f"Unsupported prepared metadata type: {type(value).__name__}")


def write_prepared_feature_payload(path, *, x_train, x_dev, x_test, y_train: Sequence[str], y_dev: Sequence[str],
                                   y_test: Sequence[str], metadata: Mapping[str, object]) -> Path:
    """Atomically persist reusable features without pickle/object arrays."""
    destination = Path(path)
    destination.parent.mkdir(parents=True, exist_ok=True)
    descriptor, raw_tmp = tempfile.mkstemp(prefix=f".{destination.name}.", dir=destination.parent)
    tmp_path = Path(raw_tmp)
    try:
        with os.fdopen(descriptor, "wb") as handle:
            np.savez_compressed(
                handle,
                x_train=np.asarray(x_train, dtype=np.float64),
                x_dev=np.asarray(x_dev, dtype=np.float64),
                x_test=np.asarray(x_test, dtype=np.float64),
                y_train=np.asarray(list(y_train), dtype=np.str_),
                y_dev=np.asarray(list(y_dev), dtype=np.str_),
                y_test=np.asarray(list(y_test), dtype=np.str_),
                metadata_json=np.asarray(
                    json.dumps(serialize_metadata(metadata), sort_keys=True, separators=(",", ":")), dtype=np.str_),
            )
            handle.flush()
            os.fsync(handle.fileno())
        os.replace(tmp_path, destination)
    finally:
        tmp_path.unlink(missing_ok=True)
    return destination


def read_prepared_feature_payload(path, *, feature_dtype: str = "float64",
                                  feature_shape: tuple[int, int] | None = None) -> PreparedFeaturePayload:
    """Read and structurally validate one payload (dtype/shape arguments play the plan's role)."""
    try:
        with np.load(Path(path), allow_pickle=False) as payload:
            if set(payload.files) != PAYLOAD_KEYS:
                raise PreparedPlanError("Prepared feature payload keys are incomplete or unknown.")
            raw = {k: np.asarray(payload[k]) for k in PAYLOAD_KEYS}
    except PreparedPlanError:
        raise
    except (OSError, ValueError, KeyError) as error:
        raise PreparedPlanError(f"Prepared feature payload is invalid: {error}") from error
    mats = [raw["x_train"], raw["x_dev"], raw["x_test"]]
    if any(str(m.dtype) != feature_dtype for m in mats):
        raise PreparedPlanError("Prepared feature payload dtype differs from the plan.")
    if any(m.ndim != 2 for m in mats):
        raise PreparedPlanError("Prepared feature matrices must all be two-dimensional.")
    combined = (int(sum(m.shape[0] for m in mats)), int(mats[0].shape[1]))
    if len({m.shape[1] for m in mats}) != 1 or (feature_shape is not None and combined != tuple(feature_shape)):
        raise PreparedPlanError("Prepared feature matrices differ from the planned shape.")
    x_train, x_dev, x_test = (np.asarray(m, dtype=np.float64) for m in mats)
    if not all(np.all(np.isfinite(m)) for m in (x_train, x_dev, x_test)):
        raise PreparedPlanError("Prepared feature matrices contain non-finite values.")
    labels = [raw["y_train"], raw["y_dev"], raw["y_test"]]
    if any(v.ndim != 1 for v in labels):
        raise PreparedPlanError("Prepared labels must be one-dimensional arrays.")
    y_train, y_dev, y_test = ([str(item) for item in v] for v in labels)
    if any(not label.strip() for label in (*y_train, *y_dev, *y_test)):
        raise PreparedPlanError("Prepared labels must be non-empty strings.")
    if x_train.shape[0] != len(y_train) or x_dev.shape[0] != len(y_dev) or x_test.shape[0] != len(y_test):
        raise PreparedPlanError("Prepared feature/label row counts do not match.")
    if raw["metadata_json"].ndim != 0:
        raise PreparedPlanError("Prepared feature metadata payload must be scalar JSON.")
    try:
        parsed = json.loads(str(raw["metadata_json"].item()))
    except json.JSONDecodeError as error:
        raise PreparedPlanError("Prepared feature metadata is invalid JSON.") from error
    if not isinstance(parsed, dict):
        raise PreparedPlanError("Prepared feature metadata must be a JSON object.")
    return PreparedFeaturePayload(x_train, x_dev, x_test, y_train, y_dev, y_test, parsed)
