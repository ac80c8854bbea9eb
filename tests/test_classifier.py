"""Fast-profile classifier on the GPU and the inference helpers around it.

Oracle for the forward pass: the fitted sklearn estimator itself (the reference calls exactly
``model.predict`` / ``model.predict_proba``, ser/_internal/models/fast_path.py:180-192).
Gates: identical labels, probabilities within 1e-10 (the reference gate is 1e-4).
"""

import logging
import warnings

import numpy as np
import pytest

LOG = logging.getLogger("test")


class _PredictOnlyModel:
    """Stand-in used by the reference's contract tests (test_emotion_model_contracts.py)."""

    def __init__(self, labels):
        self._labels = labels

    def predict(self, matrix):
        return np.asarray(self._labels[: len(matrix)])


def _seq(n, dim=3, ends=None):
    from ser_amd import EncodedSequence

    starts = np.arange(n, dtype=np.float64)
    return EncodedSequence(np.full((n, dim), 0.1, np.float32), starts,
                           np.asarray(ends, np.float64) if ends is not None else starts + 3.0, "handcrafted")


def test_label_run_merge_known_answers():
    """Segment-merge semantics pinned by the reference's known answers
    (tests/suites/unit/runtime/test_timestamp_semantics.py:32-110)."""
    from ser_amd import classify_sequence, merge_label_runs

    # frames (0,2,angry), (1,3,happy), (2,4.4,happy) -> angry 0-2, happy 1-4.4
    runs = merge_label_runs([0, 1, 1], [0.0, 1.0, 2.0], [2.0, 3.0, 4.4])
    assert runs.as_records(["angry", "happy"]) == [
        {"emotion": "angry", "start_seconds": 0.0, "end_seconds": 2.0, "confidence": 1.0, "probabilities": None},
        {"emotion": "happy", "start_seconds": 1.0, "end_seconds": 4.4, "confidence": 1.0, "probabilities": None}]
    assert runs.first.tolist() == [0, 1] and runs.last.tolist() == [0, 2]
    # length mismatch -> RuntimeError "Frame/prediction length mismatch"
    with pytest.raises(RuntimeError, match="Frame/prediction length mismatch"):
        merge_label_runs([0, 1], [0.0], [1.0])
    # a label that comes back is a new segment; confidence and probabilities are per-run means
    proba = np.asarray([[0.5, 0.5], [0.7, 0.3], [0.2, 0.8], [0.9, 0.1]])
    runs = merge_label_runs([0, 0, 1, 0], [0, 1, 2, 3], [3, 4, 5, 6], probabilities=proba)
    assert runs.label_index.tolist() == [0, 1, 0]
    assert runs.start_seconds.tolist() == [0.0, 2.0, 3.0] and runs.end_seconds.tolist() == [4.0, 5.0, 6.0]
    np.testing.assert_allclose(runs.confidence, [0.6, 0.8, 0.9])
    np.testing.assert_allclose(runs.probabilities, [[0.6, 0.4], [0.2, 0.8], [0.9, 0.1]])
    empty = merge_label_runs([], [], [])
    assert empty.label_index.size == 0 and empty.as_records([]) == []
    # a model without predict_proba: confidence 1.0, no probabilities; adjacent equal labels merge
    classes, labels, p, runs = classify_sequence(_PredictOnlyModel(["calm", "calm", "angry", "calm"]), _seq(4))
    assert p is None and [classes[i] for i in labels] == ["calm", "calm", "angry", "calm"]
    assert [(r["emotion"], r["start_seconds"], r["end_seconds"], r["confidence"]) for r in runs.as_records(classes)] == [
        ("calm", 0.0, 4.0, 1.0), ("angry", 2.0, 5.0, 1.0), ("calm", 3.0, 6.0, 1.0)]
    with pytest.raises(RuntimeError, match="Frame/prediction length mismatch"):
        classify_sequence(_PredictOnlyModel(["calm"]), _seq(2))
    # 3 600 windows of an hour in one pass
    rng = np.random.default_rng(0)
    lab = rng.integers(0, 3, size=3600)
    runs = merge_label_runs(lab, np.arange(3600.0), np.minimum(np.arange(3600.0) + 3, 3600.0))
    want_first = [0] + [i for i in range(1, 3600) if lab[i] != lab[i - 1]]
    assert runs.first.tolist() == want_first and int((runs.last - runs.first + 1).sum()) == 3600


def _reference_style_inference(frames, model):
    """What the reference's fast_path.py does with the injected pieces (frames -> model.predict /
    predict_proba -> merge equal neighbours), written down only as far as this test needs it."""
    matrix = np.asarray([f.features for f in frames], dtype=np.float64)
    labels = [str(v) for v in model.predict(matrix)]
    proba = np.asarray(model.predict_proba(matrix), dtype=np.float64)
    segments, i = [], 0
    while i < len(frames):
        j = i
        while j + 1 < len(frames) and labels[j + 1] == labels[i]:
            j += 1
        segments.append((labels[i], frames[i].start_seconds, frames[j].end_seconds))
        i = j + 1
    return labels, proba, segments


def _fit(n_classes, hidden=(300,), activation="relu", scaler=True, seed=0, dim=187, n=400):
    from sklearn.neural_network import MLPClassifier
    from sklearn.pipeline import Pipeline
    from sklearn.preprocessing import StandardScaler

    rng = np.random.default_rng(seed)
    centres = rng.standard_normal((n_classes, dim)) * 3.0
    y_idx = rng.integers(0, n_classes, size=n)
    names = np.asarray(["angry", "calm", "disgust", "fearful", "happy", "neutral", "sad", "surprised"])[:n_classes]
    x = (centres[y_idx] + rng.standard_normal((n, dim))) * rng.uniform(0.1, 50.0, size=dim) + rng.uniform(-100, 100, size=dim)
    mlp = MLPClassifier(alpha=0.01, batch_size=256, epsilon=1e-8, hidden_layer_sizes=hidden, learning_rate="adaptive",
                        max_iter=40, random_state=42, activation=activation)
    model = Pipeline([("scaler", StandardScaler()), ("classifier", mlp)]) if scaler else mlp
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.fit(x, names[y_idx])
    test = (centres[rng.integers(0, n_classes, size=257)] + 2.0 * rng.standard_normal((257, dim))) \
        * rng.uniform(0.1, 50.0, size=dim)
    return model, test


@pytest.mark.gpu
@pytest.mark.parametrize("n_classes,hidden,activation,scaler", [
    (8, (300,), "relu", True),          # the reference's configuration
    (2, (300,), "relu", True),          # binary: logistic output, {1-p, p}
    (5, (64, 32), "tanh", False),       # bare classifier, two hidden layers
    (3, (17,), "logistic", True),
])
def test_gpu_classifier_matches_sklearn(n_classes, hidden, activation, scaler):
    import torch

    from ser_amd import GpuEmotionClassifier

    model, x = _fit(n_classes, hidden, activation, scaler)
    gpu = GpuEmotionClassifier.from_sklearn(model)
    assert list(gpu.classes_) == list(model.classes_)
    np.testing.assert_array_equal(gpu.predict(x), model.predict(x))
    np.testing.assert_allclose(gpu.predict_proba(x), model.predict_proba(x), rtol=0, atol=1e-10)
    # float32 features already on the device (what FeatureEngine.features returns)
    x32 = x.astype(np.float32)
    label, proba = gpu.predict_device(torch.from_numpy(x32).cuda())
    np.testing.assert_array_equal(model.classes_[label.cpu().numpy()], model.predict(x32.astype(np.float64)))
    np.testing.assert_allclose(proba.cpu().numpy(), model.predict_proba(x32.astype(np.float64)), rtol=0, atol=1e-10)
    assert gpu.predict_proba(x[:0]).shape == (0, n_classes)
    with pytest.raises(ValueError, match="features"):
        gpu.predict(x[:, :-1])
    # a matrix large enough for the staged upload (pinned ring, memcpy team): the same labels and probabilities
    big = np.tile(x, (-(-12000 // len(x)), 1))[:12000]
    assert big.nbytes >= gpu._engine.STAGED_TO_HOST_BYTES
    np.testing.assert_array_equal(gpu.predict(big)[: len(x)], model.predict(x))
    np.testing.assert_array_equal(gpu.predict_proba(big)[: len(x)], gpu.predict_proba(x))


@pytest.mark.gpu
def test_end_to_end_file_inference_same_labels(tmp_path):
    """Config-2 style: synthetic RAVDESS clips -> GPU features -> sklearn training -> inference on a
    longer file with the sklearn model and with the GPU classifier injected where the reference takes its
    ``model=``: identical frame labels and segments, probabilities within 1e-10."""
    from ser_amd import (
        FeatureFlags,
        GpuEmotionClassifier,
        HandcraftedBackend,
        classify_sequence,
        extract_feature_frames,
    )
    from ser_amd.audio_io import prepare_audio_buffer
    from ser_amd.feature_extractor import FeatureSettings
    from ser_amd.synth import ravdess_sine_clip, write_wav_pcm16
    from sklearn.neural_network import MLPClassifier
    from sklearn.pipeline import Pipeline
    from sklearn.preprocessing import StandardScaler

    flags = FeatureFlags(tonnetz=False)
    emotions = ["neutral", "calm", "happy", "sad", "angry", "fearful", "disgust", "surprised"]
    clips, labels = [], []
    for actor in range(1, 7):
        for e in range(1, 9):
            clips.append(prepare_audio_buffer(ravdess_sine_clip(actor, e)))
            labels.append(emotions[e - 1])
    x = HandcraftedBackend(feature_flags=flags).extract_vectors(clips, 16000)
    model = Pipeline([("scaler", StandardScaler()),
                      ("classifier", MLPClassifier(alpha=0.01, batch_size=256, epsilon=1e-8, hidden_layer_sizes=(300,),
                                                   learning_rate="adaptive", max_iter=60, random_state=42))])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.fit(x, labels)
    gpu = GpuEmotionClassifier.from_sklearn(model)
    np.testing.assert_array_equal(gpu.predict(x), model.predict(x))
    # a 6-second file made of three different clips
    long_clip = np.concatenate([ravdess_sine_clip(1, 1), ravdess_sine_clip(2, 5), ravdess_sine_clip(3, 8),
                                ravdess_sine_clip(1, 3)])
    path = tmp_path / "long.wav"
    write_wav_pcm16(path, long_clip, 16000)
    settings = FeatureSettings(feature_flags=flags)
    frames = extract_feature_frames(str(path), settings=settings)
    want_labels, want_proba, want_segments = _reference_style_inference(frames, model)
    got_labels, got_proba, got_segments = _reference_style_inference(frames, gpu)   # the drop-in `model=`
    assert len(frames) == 6 and got_labels == want_labels and got_segments == want_segments
    np.testing.assert_allclose(got_proba, want_proba, rtol=0, atol=1e-10)
    # the array form: embeddings stay on the device between the feature kernels and the MLP
    backend = HandcraftedBackend(feature_flags=flags)
    from ser_amd.audio_io import read_audio_file

    audio, sr = read_audio_file(str(path))
    enc = backend.encode_sequence(audio, sr)
    classes, labels, proba, runs = classify_sequence(gpu, enc)
    assert [str(classes[i]) for i in labels] == want_labels
    assert [(r["emotion"], r["start_seconds"], r["end_seconds"]) for r in runs.as_records(classes)] == want_segments
    np.testing.assert_allclose(proba, want_proba, rtol=0, atol=1e-10)
    np.testing.assert_allclose(runs.confidence,
                               [want_proba[a : b + 1].max(axis=1).mean() for a, b in zip(runs.first, runs.last)], atol=1e-10)
