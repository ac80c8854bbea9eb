"""Frame labels -> merged emotion segments, as arrays (the step after the classifier on the fast
inference path).

Contract (pinned by the reference's known answers,
``tests/suites/unit/runtime/test_timestamp_semantics.py:32-110``): consecutive
frames with the same label form one segment that starts where its first frame starts and ends where
its last frame ends (windows overlap, so segments may too: frames (0,2,angry), (1,3,happy),
(2,4.4,happy) give angry 0-2 and happy 1-4.4); a segment's confidence and class probabilities are
the plain means over its frames.

The reference does this with per-frame Python objects inside ``models/fast_path.py``, which stays the
caller's code: pass :class:`ser_amd.GpuEmotionClassifier` as its ``model=`` and
``ser_amd.extract_feature_frames`` as its ``extract_feature_frames_fn=``.  This module is the batch
form for callers that stay on arrays (``classify_sequence``: device-resident embeddings -> GPU MLP ->
one run-length pass), e.g. an hour-long recording with 3 600 windows.
"""

from __future__ import annotations

from typing import NamedTuple

import numpy as np
from numpy.typing import NDArray

from .types import EncodedSequence


class LabelRuns(NamedTuple):
    """Run-length form of a label sequence.  Run r covers frames ``first[r] .. last[r]`` inclusive."""

    label_index: NDArray[np.int64]      # [R] index into ``classes``
    first: NDArray[np.int64]            # [R]
    last: NDArray[np.int64]             # [R]
    start_seconds: NDArray[np.float64]  # [R] start of the run's first frame
    end_seconds: NDArray[np.float64]    # [R] end of the run's last frame
    confidence: NDArray[np.float64]     # [R] mean frame confidence
    probabilities: NDArray[np.float64] | None   # [R, C] mean class probabilities, or None

    def as_records(self, classes) -> list[dict]:
        """``[{emotion, start_seconds, end_seconds, confidence, probabilities}]`` -- the fields of the
        reference's ``SegmentPrediction`` (ser/runtime/schema.py:13-41) as plain dicts."""
        names = [str(c) for c in classes]
        out = []
        for r in range(self.label_index.size):
            proba = None
            if self.probabilities is not None:
                proba = {names[c]: float(self.probabilities[r, c]) for c in range(len(names))}
            out.append({"emotion": names[int(self.label_index[r])], "start_seconds": float(self.start_seconds[r]),
                        "end_seconds": float(self.end_seconds[r]), "confidence": float(self.confidence[r]),
                        "probabilities": proba})
        return out


def merge_label_runs(label_index, start_seconds, end_seconds, confidence=None, probabilities=None) -> LabelRuns:
    """One vectorised run-length pass over per-frame label indices.

    ``confidence`` defaults to the row maximum of ``probabilities`` when those are given, else to 1.0
    (what the reference reports for a model without ``predict_proba``)."""
    labels = np.asarray(label_index, dtype=np.int64).reshape(-1)
    t0 = np.asarray(start_seconds, dtype=np.float64).reshape(-1)
    t1 = np.asarray(end_seconds, dtype=np.float64).reshape(-1)
    n = labels.size
    if t0.size != n or t1.size != n:
        raise RuntimeError(f"Frame/prediction length mismatch. Got {t0.size} frames and {n} predictions.")
    proba = None
    if probabilities is not None:
        proba = np.asarray(probabilities, dtype=np.float64)
        if proba.ndim != 2 or proba.shape[0] != n:
            raise ValueError(f"probabilities must be [{n}, classes], got {proba.shape}")
    if confidence is None:
        conf = proba.max(axis=1) if proba is not None and proba.shape[1] else np.ones(n, dtype=np.float64)
    else:
        conf = np.asarray(confidence, dtype=np.float64).reshape(-1)
        if conf.size != n:
            raise ValueError(f"confidence must have {n} entries, got {conf.size}")
    if n == 0:
        empty_i, empty_f = np.empty(0, np.int64), np.empty(0, np.float64)
        return LabelRuns(empty_i, empty_i, empty_i, empty_f, empty_f, empty_f,
                         None if proba is None else np.empty((0, proba.shape[1]), np.float64))
    first = np.flatnonzero(np.concatenate(([True], labels[1:] != labels[:-1])))
    last = np.concatenate((first[1:] - 1, [n - 1]))
    counts = (last - first + 1).astype(np.float64)
    mean_conf = np.add.reduceat(conf, first) / counts
    mean_proba = None if proba is None else np.add.reduceat(proba, first, axis=0) / counts[:, None]
    return LabelRuns(labels[first], first, last, t0[first], t1[last], mean_conf, mean_proba)


def classify_sequence(model, encoded: EncodedSequence, *, embeddings_device=None):
    """Labels, probabilities and merged runs for one encoded recording.

    ``model`` is a :class:`ser_amd.GpuEmotionClassifier` (``predict_device``: features may stay in HBM --
    pass the device tensor as ``embeddings_device``) or any fitted estimator with ``predict`` and
    optionally ``predict_proba`` / ``classes_``.  Returns ``(classes, label_index [n], probabilities
    [n, C] or None, LabelRuns)``."""
    n = int(encoded.embeddings.shape[0])
    if hasattr(model, "predict_device"):
        source = embeddings_device if embeddings_device is not None else encoded.embeddings
        label_dev, proba_dev = model.predict_device(source)
        labels = label_dev.cpu().numpy().astype(np.int64)
        proba = proba_dev.cpu().numpy()
        classes = np.asarray(model.classes_)
    else:
        matrix = np.asarray(encoded.embeddings, dtype=np.float64)
        predicted = np.asarray(model.predict(matrix))
        if predicted.shape[0] != n:
            raise RuntimeError(f"Frame/prediction length mismatch. Got {n} frames and {predicted.shape[0]} predictions.")
        known = getattr(model, "classes_", None)
        classes = np.asarray(known) if known is not None else np.unique(predicted)
        lookup = {str(c): i for i, c in enumerate(classes.tolist())}
        labels = np.asarray([lookup[str(p)] for p in predicted.tolist()], dtype=np.int64)
        proba = None
        if callable(getattr(model, "predict_proba", None)) and known is not None:
            candidate = np.asarray(model.predict_proba(matrix), dtype=np.float64)
            if candidate.shape == (n, classes.size):
                proba = candidate
    runs = merge_label_runs(labels, encoded.frame_start_seconds, encoded.frame_end_seconds, probabilities=proba)
    return classes, labels, proba, runs
