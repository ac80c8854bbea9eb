"""Dataset loader host logic and the prepared-feature cache container (CPU) + the GPU loader run."""

import numpy as np
import pytest

from ser_amd.data_loader import (
    LoaderSettings,
    _resolve_worker_count,
    extract_ravdess_speaker_id_from_path,
    label_for_file,
    process_file,
)
from ser_amd.prepared_cache import (
    PreparedPlanError,
    read_prepared_feature_payload,
    write_prepared_feature_payload,
)


def test_worker_count_and_name_parsing():
    # tests/suites/unit/data/test_data_loader.py (worker count, name format)
    assert _resolve_worker_count(16, 8, 100) == 8
    assert _resolve_worker_count(4, 8, 100) == 4
    assert _resolve_worker_count(16, 8, 3) == 3
    assert _resolve_worker_count(16, 8, 0) == 1
    s = LoaderSettings()
    observed = set(s.emotions.values())
    assert label_for_file("/x/Actor_01/03-01-05-01-01-01-01.wav", observed, s.emotions) == ("angry", None)
    assert label_for_file("/x/03-01-99-01.wav", observed, s.emotions) == (None, None)
    assert label_for_file("/x/03-01-05-01.wav", {"happy"}, s.emotions) == (None, None)
    label, err = label_for_file("/x/weird.wav", observed, s.emotions)
    assert label is None and "unexpected name format" in err
    assert extract_ravdess_speaker_id_from_path("/x/03-01-05-01-01-01-17.wav") == "17"
    assert extract_ravdess_speaker_id_from_path("/x/a-b.wav") is None
    assert s.glob_pattern.endswith("Actor_*/*.wav")
    r = process_file("/x/weird.wav", observed, s.emotions, settings=s)
    assert r.sample is None and "unexpected name format" in r.error
    r = process_file("/x/03-01-05-01-01-01-01.wav", observed, s.emotions, settings=s)   # missing file
    assert r.sample is None and r.error.startswith("Failed to process file")
    with pytest.raises(ValueError, match="test_size"):
        from ser_amd.data_loader import load_data

        load_data(1.5, settings=s)


def test_prepared_cache_round_trip_and_validation(tmp_path):
    rng = np.random.default_rng(0)
    xs = [rng.standard_normal((n, 187)) for n in (5, 2, 3)]
    ys = [["happy"] * 5, ["sad", "calm"], ["angry"] * 3]
    meta = {"cache_namespace": "fast_features", "feature_dim": 187, "paths": (tmp_path / "a",), "nested": {"k": 1}}
    path = write_prepared_feature_payload(tmp_path / "tmp" / "prepared-training-fast.npz", x_train=xs[0], x_dev=xs[1],
                                          x_test=xs[2], y_train=ys[0], y_dev=ys[1], y_test=ys[2], metadata=meta)
    assert path.exists() and not list(path.parent.glob(".prepared*"))        # temp file replaced atomically
    with np.load(path, allow_pickle=False) as z:                               # the reference's container
        assert set(z.files) == {"x_train", "x_dev", "x_test", "y_train", "y_dev", "y_test", "metadata_json"}
        assert z["x_train"].dtype == np.float64 and z["y_train"].dtype.kind == "U" and z["metadata_json"].ndim == 0
    got = read_prepared_feature_payload(path, feature_shape=(10, 187))
    for a, b in zip(xs, (got.x_train, got.x_dev, got.x_test)):
        np.testing.assert_array_equal(a, b)
    assert [got.y_train, got.y_dev, got.y_test] == ys
    assert got.metadata["paths"] == [str(tmp_path / "a")] and got.metadata["nested"] == {"k": 1}
    with pytest.raises(PreparedPlanError, match="planned shape"):
        read_prepared_feature_payload(path, feature_shape=(11, 187))
    with pytest.raises(PreparedPlanError, match="dtype differs"):
        read_prepared_feature_payload(path, feature_dtype="float32")
    bad = xs[0].copy()
    bad[0, 0] = np.nan
    p2 = write_prepared_feature_payload(tmp_path / "bad.npz", x_train=bad, x_dev=xs[1], x_test=xs[2], y_train=ys[0],
                                        y_dev=ys[1], y_test=ys[2], metadata={})
    with pytest.raises(PreparedPlanError, match="non-finite"):
        read_prepared_feature_payload(p2)
    p3 = write_prepared_feature_payload(tmp_path / "rows.npz", x_train=xs[0], x_dev=xs[1], x_test=xs[2],
                                        y_train=ys[0][:4], y_dev=ys[1], y_test=ys[2], metadata={})
    with pytest.raises(PreparedPlanError, match="row counts"):
        read_prepared_feature_payload(p3)
    np.savez_compressed(tmp_path / "keys.npz", x_train=xs[0])
    with pytest.raises(PreparedPlanError, match="incomplete or unknown"):
        read_prepared_feature_payload(tmp_path / "keys.npz")
    (tmp_path / "junk.npz").write_bytes(b"junk")
    with pytest.raises(PreparedPlanError, match="invalid"):
        read_prepared_feature_payload(tmp_path / "junk.npz")
    with pytest.raises(TypeError, match="Unsupported prepared metadata type"):
        write_prepared_feature_payload(tmp_path / "m.npz", x_train=xs[0], x_dev=xs[1], x_test=xs[2], y_train=ys[0],
                                       y_dev=ys[1], y_test=ys[2], metadata={"x": object()})


def _build_dataset(root, actors=(1, 2, 3, 4), sample_rate=16000):
    from ser_amd.synth import ravdess_sine_clip, write_wav_pcm16

    files = []
    for actor in actors:
        for emo in range(1, 9):
            clip = ravdess_sine_clip(actor, emo, sample_rate=sample_rate)
            path = root / f"Actor_{actor:02d}" / f"03-01-{emo:02d}-01-01-01-{actor:02d}.wav"
            path.parent.mkdir(parents=True, exist_ok=True)
            write_wav_pcm16(path, np.rint(clip * 32768).astype(np.int16), sample_rate)
            files.append(path)
    return files


@pytest.mark.gpu
def test_config2_training_featurisation_from_files(tmp_path):
    """Config 2 end to end: synthetic RAVDESS files -> GPU loader -> split -> feature cache,
    checked against the oracle run on the same decoded files with the same split."""
    from sklearn.model_selection import train_test_split

    from oracle import ser_fast_path as ofp
    from parity_utils import assert_features_close
    from ser_amd.audio_io import read_audio_file
    from ser_amd.data_loader import featurize_files, load_checked_fast_data, load_data

    files = _build_dataset(tmp_path / "ravdess")
    settings = LoaderSettings(dataset_folder=tmp_path / "ravdess")
    x_train, x_test, y_train, y_test = load_data(settings=settings)
    assert x_train.shape == (24, 187) and x_test.shape == (8, 187) and x_train.dtype == np.float64
    ordered = sorted(str(f) for f in files)
    want_rows = np.asarray([ofp.extract_vector(read_audio_file(f)[0], 16000) for f in ordered])
    want_labels = [settings.emotions[f.rsplit("/", 1)[1].split("-")[2]] for f in ordered]
    ref = train_test_split(want_rows, want_labels, test_size=0.25, random_state=42, stratify=want_labels)
    assert y_train == list(ref[2]) and y_test == list(ref[3])
    assert_features_close(x_train, ref[0])
    assert_features_close(x_test, ref[1])
    # failure-ratio guard (data_loader.py:391-400)
    (tmp_path / "ravdess" / "Actor_01" / "03-01-02-01-01-02-01.wav").write_bytes(b"junk")
    quick = LoaderSettings(dataset_folder=tmp_path / "ravdess",
                           audio_read=settings.audio_read.__class__(max_retries=1, retry_delay_seconds=0.0))
    with pytest.raises(RuntimeError, match=r"3\.0% file failures exceeded configured limit 1\.0%"):
        load_data(settings=quick)
    matrix, errors = featurize_files(ordered[:3] + [str(tmp_path / "missing.wav")], settings=quick)
    assert [e is None for e in errors] == [True, True, True, False] and np.isnan(matrix[3]).all()
    assert_features_close(matrix[:3], want_rows[:3])

    class Utt:
        def __init__(self, path, label, start=None, dur=None):
            self.audio_path, self.label, self.start_seconds, self.duration_seconds = path, label, start, dur
            self.sample_id = f"id:{path}"

        def require_label(self):
            return self.label

    utts = [Utt(f, lab) for f, lab in zip(ordered, want_labels)]
    utts[5] = Utt(ordered[5], want_labels[5], start=0.25, dur=1.0)
    broken = Utt(str(tmp_path / "gone.wav"), "happy")
    skipped = []
    out = load_checked_fast_data(train_utterances=utts[:16] + [broken], test_utterances=utts[16:], settings=quick,
                                 handle_sample_failure=lambda u, e: skipped.append(u.sample_id) or True)
    assert out[0].shape == (16, 187) and out[1].shape == (16, 187) and skipped == [broken.sample_id]
    seg_audio = read_audio_file(ordered[5], start_seconds=0.25, duration_seconds=1.0)[0]
    assert_features_close(out[0][5], ofp.extract_vector(seg_audio, 16000))
    with pytest.raises(FileNotFoundError):
        load_checked_fast_data(train_utterances=utts[:4] + [broken], test_utterances=utts[16:], settings=quick)
    # prepared-feature cache filled from the GPU features
    from ser_amd.prepared_cache import read_prepared_feature_payload, write_prepared_feature_payload

    p = write_prepared_feature_payload(tmp_path / "tmp" / "prepared-training-fast.npz", x_train=x_train,
                                       x_dev=np.empty((0, 187)), x_test=x_test, y_train=y_train, y_dev=[],
                                       y_test=y_test, metadata={"cache_namespace": "fast_features"})
    back = read_prepared_feature_payload(p, feature_shape=(32, 187))
    np.testing.assert_array_equal(back.x_train, x_train)
    assert back.y_test == y_test
