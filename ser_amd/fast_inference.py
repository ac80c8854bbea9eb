"""Fast-profile inference helpers with the reference's signatures
(``ser/_internal/models/fast_path.py:19-226``, result types of ``ser/runtime/schema.py:13-41``).

``predict_emotions_detailed_with_model`` is the reference's function: feature frames from
``extract_feature_frames_fn`` (pass ``ser_amd.extract_feature_frames``), labels and probabilities
from ``model`` -- a fitted sklearn estimator or a :class:`ser_amd.classifier.GpuEmotionClassifier`,
which runs the MLP on the GPU -- and adjacent equal labels merged into segments on the host.
"""

from __future__ import annotations

import logging
from collections.abc import Callable, Sequence
from dataclasses import dataclass
from statistics import fmean

import numpy as np

from .types import FeatureFrame


@dataclass(frozen=True)
class FramePrediction:
    """One frame-level inference prediction."""

    start_seconds: float
    end_seconds: float
    emotion: str
    confidence: float
    probabilities: dict[str, float] | None


@dataclass(frozen=True)
class SegmentPrediction:
    """Merged segment-level inference prediction."""

    emotion: str
    start_seconds: float
    end_seconds: float
    confidence: float
    probabilities: dict[str, float] | None = None


@dataclass(frozen=True)
class InferenceResult:
    """Full inference payload with frame and segment predictions."""

    schema_version: str
    segments: list[SegmentPrediction]
    frames: list[FramePrediction]


def frame_confidence_and_probabilities(model, feature_matrix: np.ndarray, frame_count: int, *,
                                       logger: logging.Logger):
    """Per-frame confidence (max probability) and class-probability maps; the reference's fallbacks
    (confidence 1.0, probabilities None) when the model cannot supply consistent probabilities."""
    fallback_confidence = [1.0] * frame_count
    fallback_probabilities: list[dict[str, float] | None] = [None] * frame_count
    predict_proba = getattr(model, "predict_proba", None)
    if not callable(predict_proba):
        logger.warning("Loaded model does not expose predict_proba; using confidence=1.0 fallback.")
        return fallback_confidence, fallback_probabilities
    classes_attr = getattr(model, "classes_", None)
    class_values = None
    if isinstance(classes_attr, np.ndarray):
        class_values = list(classes_attr.tolist())
    elif isinstance(classes_attr, (list, tuple)):
        class_values = list(classes_attr)
    if class_values is None:
        logger.warning("Loaded model predict_proba path missing classes_; using confidence fallback.")
        return fallback_confidence, fallback_probabilities
    class_labels = [str(item) for item in class_values]
    raw = np.asarray(predict_proba(feature_matrix), dtype=np.float64)
    if raw.ndim != 2:
        logger.warning("Unexpected predict_proba output shape %s; using confidence fallback.", raw.shape)
        return fallback_confidence, fallback_probabilities
    if raw.shape[0] != frame_count:
        logger.warning("predict_proba frame count mismatch (expected=%s, got=%s); using fallback.",
                       frame_count, raw.shape[0])
        return fallback_confidence, fallback_probabilities
    if raw.shape[1] != len(class_labels):
        logger.warning("predict_proba class count mismatch (classes=%s, probs=%s); using fallback.",
                       len(class_labels), raw.shape[1])
        return fallback_confidence, fallback_probabilities
    confidences = [float(np.max(row)) for row in raw]
    probabilities = [{class_labels[i]: float(row[i]) for i in range(len(class_labels))} for row in raw]
    return confidences, probabilities


def aggregate_probabilities(probabilities):
    """Mean of per-frame probability maps when every frame has a full map over the same labels."""
    if not probabilities or any(item is None for item in probabilities):
        return None
    labels = list(probabilities[0].keys())
    if any(set(item.keys()) != set(labels) for item in probabilities[1:]):
        return None
    return {label: float(fmean([item[label] for item in probabilities])) for label in labels}


def segment_predictions(frame_predictions: list[FramePrediction]) -> list[SegmentPrediction]:
    """Merge adjacent frames with equal labels into segments (confidence and probabilities averaged)."""
    if not frame_predictions:
        return []
    segments: list[SegmentPrediction] = []
    run = [frame_predictions[0]]

    def close() -> None:
        segments.append(SegmentPrediction(
            emotion=run[0].emotion, start_seconds=run[0].start_seconds, end_seconds=run[-1].end_seconds,
            confidence=float(fmean([f.confidence for f in run])),
            probabilities=aggregate_probabilities([f.probabilities for f in run])))

    for frame in frame_predictions[1:]:
        if frame.emotion == run[0].emotion:
            run.append(frame)
            continue
        close()
        run = [frame]
    close()
    return segments


def predict_emotions_detailed_with_model(file: str, *, model, expected_feature_size: int | None,
                                         output_schema_version: str,
                                         extract_feature_frames_fn: Callable[[str], Sequence[FeatureFrame]],
                                         logger: logging.Logger) -> InferenceResult:
    """Fast-path inference with a preloaded model: frames -> labels/probabilities -> merged segments."""
    feature_frames = list(extract_feature_frames_fn(file))
    if not feature_frames:
        logger.warning("No features extracted for file %s.", file)
        return InferenceResult(schema_version=output_schema_version, segments=[], frames=[])
    vectors = [frame.features for frame in feature_frames]
    if expected_feature_size is not None:
        invalid = {v.shape[0] for v in vectors if v.shape[0] != expected_feature_size}
        if invalid:
            raise ValueError("Feature vector size mismatch for loaded model. "
                             f"Expected {expected_feature_size}, got {sorted(invalid)}.")
    matrix = np.asarray(vectors, dtype=np.float64)
    predicted = [str(item) for item in model.predict(matrix)]
    if len(predicted) != len(feature_frames):
        raise RuntimeError("Frame/prediction length mismatch. "
                           f"Got {len(feature_frames)} frames and {len(predicted)} predictions.")
    confidences, probabilities = frame_confidence_and_probabilities(model, matrix, len(feature_frames), logger=logger)
    logger.debug("Emotion model prediction completed for %d frames.", len(predicted))
    frames = [FramePrediction(start_seconds=feature_frames[i].start_seconds, end_seconds=feature_frames[i].end_seconds,
                              emotion=predicted[i], confidence=confidences[i], probabilities=probabilities[i])
              for i in range(len(feature_frames))]
    logger.debug("Timestamp extraction started.")
    segments = segment_predictions(frames)
    logger.debug("Timestamp extraction completed for %d segments.", len(segments))
    return InferenceResult(schema_version=output_schema_version, segments=segments, frames=frames)
