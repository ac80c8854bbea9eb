"""Dataset loader host logic and the prepared-feature cache container (CPU) + the GPU loader run."""

import json

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
    with pytest.raises(PreparedPlanError, match="x_train has dtype float64, the plan says float32"):
        read_prepared_feature_payload(path, feature_dtype="float32")
    bad = xs[0].copy()
    bad[0, 0] = np.nan
    p2 = write_prepared_feature_payload(tmp_path / "bad.npz", x_train=bad, x_dev=xs[1], x_test=xs[2], y_train=ys[0],
                                        y_dev=ys[1], y_test=ys[2], metadata={})
    with pytest.raises(PreparedPlanError, match="non-finite"):
        read_prepared_feature_payload(p2)
    p3 = write_prepared_feature_payload(tmp_path / "rows.npz", x_train=xs[0], x_dev=xs[1], x_test=xs[2],
                                        y_train=ys[0][:4], y_dev=ys[1], y_test=ys[2], metadata={})
    with pytest.raises(PreparedPlanError, match="partition train: 5 feature rows but 4 labels"):
        read_prepared_feature_payload(p3)
    np.savez_compressed(tmp_path / "keys.npz", x_train=xs[0])
    with pytest.raises(PreparedPlanError, match="missing or unknown: metadata_json, x_dev"):
        read_prepared_feature_payload(tmp_path / "keys.npz")
    (tmp_path / "junk.npz").write_bytes(b"junk")
    with pytest.raises(PreparedPlanError, match="cannot be read"):
        read_prepared_feature_payload(tmp_path / "junk.npz")
    assert not list(tmp_path.glob(".*.part")) and not list((tmp_path / "tmp").glob(".*.part"))   # nothing left behind
    # a failing write leaves the old file in place and no temp file
    before = path.read_bytes()
    with pytest.raises(TypeError):
        write_prepared_feature_payload(path, x_train=xs[0], x_dev=xs[1], x_test=xs[2], y_train=ys[0], y_dev=ys[1],
                                       y_test=ys[2], metadata={"x": {1, 2}})
    assert path.read_bytes() == before and not list(path.parent.glob(".*.part"))
    # blank labels, label arrays of the wrong rank, metadata that is not an object
    def raw_write(name, **over):
        arrays = dict(x_train=xs[0], x_dev=xs[1], x_test=xs[2], y_train=np.asarray(ys[0]), y_dev=np.asarray(ys[1]),
                      y_test=np.asarray(ys[2]), metadata_json=np.asarray("{}"))
        arrays.update(over)
        np.savez_compressed(tmp_path / name, **arrays)
        return tmp_path / name
    with pytest.raises(PreparedPlanError, match="y_dev holds 1 blank"):
        read_prepared_feature_payload(raw_write("blank.npz", y_dev=np.asarray(["sad", " "])))
    with pytest.raises(PreparedPlanError, match="y_test must be a 1-D array of strings"):
        read_prepared_feature_payload(raw_write("rank.npz", y_test=np.asarray([["a", "b", "c"]])))
    with pytest.raises(PreparedPlanError, match="metadata_json must decode to a JSON object"):
        read_prepared_feature_payload(raw_write("meta.npz", metadata_json=np.asarray("[1]")))
    with pytest.raises(PreparedPlanError, match="metadata_json must be one scalar"):
        read_prepared_feature_payload(raw_write("meta2.npz", metadata_json=np.asarray(["{}"])))
    with pytest.raises(PreparedPlanError, match="x_dev has 1 dimensions"):
        read_prepared_feature_payload(raw_write("dim.npz", x_dev=np.zeros(3)))
    with pytest.raises(PreparedPlanError, match="differ in width"):
        read_prepared_feature_payload(raw_write("width.npz", x_dev=np.zeros((2, 5))))
    with pytest.raises(TypeError, match="Unsupported prepared metadata type"):
        write_prepared_feature_payload(tmp_path / "m.npz", x_train=xs[0], x_dev=xs[1], x_test=xs[2], y_train=ys[0],
                                       y_dev=ys[1], y_test=ys[2], metadata={"x": object()})


def test_payload_file_is_the_container_numpy_writes(tmp_path, monkeypatch):
    """``write_npz_deflated`` (members deflated piece by piece on a team of threads): any zip reader accepts the file
    (CRCs, sizes, directory), ``np.load`` returns the arrays ``np.savez_compressed`` would have stored -- dtype, shape and
    bits --, for pieces much smaller than a member, a member that is exactly a multiple of the piece size, empty
    matrices, unicode labels and the 0-d JSON string; zip64-sized archives are left to numpy's own writer."""
    import io
    import zipfile

    from ser_amd import prepared_cache as pc

    rng = np.random.default_rng(41)
    arrays = {"x_train": rng.standard_normal((513, 187)).astype(np.float32).astype(np.float64), "x_dev": np.zeros((0, 187)),
              "x_test": rng.standard_normal((3, 187)), "y_train": np.asarray([f"e{i % 8}" for i in range(513)], dtype=np.str_),
              "y_dev": np.asarray([], dtype=np.str_), "y_test": np.asarray(["a", "é", "c"]), "metadata_json": np.asarray('{"a":1}'),
              "ints": np.arange(10, dtype=np.int32).reshape(2, 5), "fortran": np.asfortranarray(rng.standard_normal((300, 7))),
              "exact": np.zeros(4096 - 128 // 8 * 0, dtype=np.uint8)}
    want_buf = io.BytesIO()
    np.savez_compressed(want_buf, **arrays)
    want_buf.seek(0)
    want = np.load(want_buf, allow_pickle=False)
    for piece, threads, level in ((4096, 3, 1), (1000, 1, 6), (1 << 20, 8, None)):
        buf = io.BytesIO()
        buf.write(b"")   # (the archive starts where the handle stands)
        pc.write_npz_deflated(buf, arrays, piece_bytes=piece, threads=threads, level=level)
        buf.seek(0)
        with zipfile.ZipFile(buf) as z:
            assert z.testzip() is None
            assert z.namelist() == [f"{k}.npy" for k in arrays]
            assert all(i.compress_type == zipfile.ZIP_DEFLATED for i in z.infolist())
        buf.seek(0)
        got = np.load(buf, allow_pickle=False)
        assert set(got.files) == set(arrays)
        for k in arrays:
            assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape and np.array_equal(got[k], want[k]), (k, piece)
            assert got[k].flags.f_contiguous == want[k].flags.f_contiguous, k
    with pytest.raises(ValueError, match="object arrays"):
        pc.write_npz_deflated(io.BytesIO(), {"o": np.asarray([{"a": 1}], dtype=object)})
    # the published file goes through it, and a plain numpy reader opens it
    path = pc.write_prepared_feature_payload(tmp_path / "p.npz", x_train=arrays["x_train"], x_dev=arrays["x_dev"], x_test=arrays["x_test"],
                                             y_train=arrays["y_train"], y_dev=[], y_test=["a", "b", "c"], metadata={"k": [1, 2]})
    with np.load(path, allow_pickle=False) as z:
        assert np.array_equal(z["x_train"], arrays["x_train"]) and json.loads(str(z["metadata_json"][()])) == {"k": [1, 2]}
    assert pc.read_prepared_feature_payload(path, feature_shape=(516, 187)).y_test == ["a", "b", "c"]
    # an archive that would need zip64: numpy's writer is called instead
    called = {}
    monkeypatch.setattr(pc, "_ZIP32_LIMIT", 1000)
    monkeypatch.setattr(pc.np, "savez_compressed", lambda handle, **kw: called.update(kw))
    pc.write_npz_deflated(io.BytesIO(), {"x": arrays["x_train"]})
    assert list(called) == ["x"]


def test_prepared_cache_writer_takes_the_references_keywords(tmp_path):
    """The reference's call shape (training_orchestration.py:842-856): keyword-only ``settings``, the six
    arrays, ``metadata`` and ``path=None``; without a path the file goes to
    ``settings.tmp_folder / "prepared-training-fast.npz"`` (training_readiness.py:880-882)."""
    from types import SimpleNamespace as NS

    from ser_amd.prepared_cache import default_prepared_payload_path

    x = np.ones((2, 187))
    settings = NS(tmp_folder=tmp_path / "tmp")
    kw = dict(x_train=x, x_dev=x[:0], x_test=x, y_train=["a", "b"], y_dev=[], y_test=["a", "b"], metadata={"k": 1})
    p = write_prepared_feature_payload(settings=settings, **kw)
    assert p == tmp_path / "tmp" / "prepared-training-fast.npz" == default_prepared_payload_path(settings) and p.exists()
    q = write_prepared_feature_payload(settings=settings, path=tmp_path / "elsewhere.npz", **kw)
    assert q == tmp_path / "elsewhere.npz" and q.exists()
    assert read_prepared_feature_payload(q, feature_shape=(4, 187)).metadata == {"k": 1}
    with pytest.raises(ValueError, match="needs `path` or settings"):
        write_prepared_feature_payload(**kw)


class _Utt:
    """What the loader reads from the reference's ``Utterance`` (data/manifest types)."""

    def __init__(self, path, label, start=None, dur=None, split=None):
        self.audio_path, self.label, self.start_seconds, self.duration_seconds = path, label, start, dur
        self.sample_id = f"id:{path}"
        if split is not None:
            self.split = split

    def require_label(self):
        return self.label


def _app_config(tmp_path, flags):
    """A fake shaped like the reference's nested ``AppConfig`` (config/schema.py:243-312,629-660): only
    attribute access, none of this package's types."""
    from types import SimpleNamespace as NS

    return NS(dataset=NS(glob_pattern=str(tmp_path / "ravdess" / "Actor_*" / "*.wav")),
              emotions={"01": "neutral", "03": "happy", "05": "angry"},
              training=NS(test_size=0.4, random_state=7, stratify_split=True),
              data_loader=NS(max_workers=3, max_failed_file_ratio=0.5),
              models=NS(num_cores=2),
              feature_flags=NS(mfcc=True, chroma=True, mel=True, contrast=True, tonnetz=flags),
              audio_read=NS(max_retries=1, retry_delay_seconds=0.0))


def test_reference_shaped_settings_and_checked_seam(tmp_path, monkeypatch):
    """The training seams take what the reference passes them:
    ``load_checked_fast_data(*, utterances, settings, handle_sample_failure)`` with a nested AppConfig
    (data_loader.py:467-472; hook wiring models/training_entrypoints.py:69-80) and
    ``load_data(test_size, *, settings: AppConfig)`` (:335-339,447-464).  The GPU featuriser is replaced by a
    fake here (CPU suite); the GPU run of the same seams is test_config2_training_featurisation_from_files."""
    from ser_amd import data_loader as dl
    from ser_amd.types import AudioReadConfig, FeatureFlags

    cfg = _app_config(tmp_path, flags=False)
    view = dl.resolve_loader_settings(cfg)
    assert view.glob_pattern.endswith("Actor_*/*.wav") and view.test_size == 0.4 and view.random_state == 7
    assert view.max_workers == 3 and view.num_cores == 2 and view.max_failed_file_ratio == 0.5
    assert view.feature_flags == FeatureFlags(tonnetz=False)
    assert view.audio_read == AudioReadConfig(max_retries=1, retry_delay_seconds=0.0)
    flat = dl.resolve_loader_settings(dl.LoaderSettings(dataset_folder=tmp_path, feature_flags=FeatureFlags(tonnetz=False)))
    assert flat.glob_pattern == str(tmp_path / "Actor_*" / "*.wav") and flat.test_size == 0.25
    assert dl.resolve_loader_settings(view) is view

    seen = []

    def fake_jobs(jobs, *, feature_flags, audio_read_config, workers=8, device_prepare=True):
        seen.append((list(jobs), feature_flags, audio_read_config, workers))
        rows = np.asarray([[float(len(p)), float(s or 0.0), float(d or 0.0)] for p, s, d in jobs], dtype=np.float64)
        errs = [FileNotFoundError(p) if "gone" in p else None for p, _s, _d in jobs]
        rows[[e is not None for e in errs]] = np.nan
        return rows, errs

    monkeypatch.setattr(dl, "featurize_audio_jobs", fake_jobs)
    labels = ["happy", "angry", "neutral"] * 4
    utts = [_Utt(f"/d/u{i}.wav", lab, start=0.5 * i if i % 2 else None, dur=1.0 if i % 2 else None)
            for i, lab in enumerate(labels)]
    gone = _Utt("/d/gone.wav", "happy")
    skipped = []
    # keyword-only, the reference's names; the default splitter honours settings.training.*
    out = dl.load_checked_fast_data(utterances=utts + [gone], settings=cfg,
                                    handle_sample_failure=lambda u, e: skipped.append((u.sample_id, type(e))) or True)
    x_train, x_test, y_train, y_test = out
    assert skipped == [(gone.sample_id, FileNotFoundError)]
    assert x_train.dtype == np.float64 and x_train.shape[0] == len(y_train) and x_test.shape[0] == len(y_test)
    assert len(y_train) + len(y_test) == 12 and len(y_test) in (5, 6)            # test_size 0.4 of 13, one skipped
    assert sorted(y_train + y_test) == sorted(labels)
    assert all(fl == FeatureFlags(tonnetz=False) and ar.max_retries == 1 and w <= 2 for _j, fl, ar, w in seen)
    assert seen[0][0][1] == ("/d/u1.wav", 0.5, 1.0) or ("/d/u1.wav", 0.5, 1.0) in seen[0][0] + seen[1][0]
    # an injected splitter is called the way the reference calls its own
    calls = []

    def ref_splitter(*, samples, settings, logger):
        calls.append((len(samples), settings is cfg, logger is dl.logger))
        return samples[:9], samples[9:], {"strategy": "fake"}

    out2 = dl.load_checked_fast_data(utterances=utts, settings=cfg, split_utterances=ref_splitter)
    assert calls == [(12, True, True)] and out2[0].shape == (9, 3) and out2[3] == labels[9:]
    np.testing.assert_array_equal(out2[0][1], [len("/d/u1.wav"), 0.5, 1.0])
    # without a handler the sample's own exception propagates (data_loader.py:507-515)
    with pytest.raises(FileNotFoundError):
        dl.load_checked_fast_data(utterances=utts + [gone], settings=cfg, split_utterances=lambda **k: (k["samples"], k["samples"][:3], {}))
    assert dl.load_checked_fast_data(utterances=[], settings=cfg) is None
    with pytest.raises(RuntimeError, match="fewer than two training classes"):
        dl.load_checked_fast_data(utterances=utts, settings=cfg,
                                  split_utterances=lambda **k: ([u for u in k["samples"] if u.label == "happy"],
                                                                k["samples"][:2], {}))
    # manifest splits carried by the utterances win over the shuffle
    tagged = [_Utt(f"/d/m{i}.wav", lab, split="train" if i < 8 else ("dev" if i < 10 else "test"))
              for i, lab in enumerate(labels)]
    tr, te, info = dl.default_split_utterances(samples=tagged, settings=cfg, logger=None)
    assert len(tr) == 8 and len(te) == 4 and info["strategy"] == "manifest"
    # the two-partition helper is still there
    out3 = dl.load_checked_partitions(train_utterances=utts[:6], test_utterances=utts[6:], settings=cfg)
    assert out3[0].shape == (6, 3) and out3[1].shape == (6, 3)

    # load_data(test_size, settings=AppConfig): glob -> label filter -> featurise -> split
    for actor in (1, 2):
        for emo in (1, 2, 3, 5):                       # "02" (calm) is not in cfg.emotions: silently skipped
            for rep in (1, 2):
                f = tmp_path / "ravdess" / f"Actor_{actor:02d}" / f"03-01-{emo:02d}-01-01-{rep:02d}-{actor:02d}.wav"
                f.parent.mkdir(parents=True, exist_ok=True)
                f.write_bytes(b"x")
    (tmp_path / "ravdess" / "Actor_01" / "weird.wav").write_bytes(b"x")    # counted as a failure
    monkeypatch.setattr(dl, "featurize_audio_jobs_device", fake_jobs)
    x_train, x_test, y_train, y_test = dl.load_data(settings=cfg)
    assert len(y_train) + len(y_test) == 12 and len(y_test) == 5 and set(y_train) == {"neutral", "happy", "angry"}
    assert dl.load_data(0.25, settings=cfg)[1].shape[0] == 3
    strict = _app_config(tmp_path, flags=False)
    strict.data_loader.max_failed_file_ratio = 0.01
    with pytest.raises(RuntimeError, match=r"5\.9% file failures exceeded configured limit 1\.0%"):
        dl.load_data(settings=strict)
    # a reference-default AppConfig (tonnetz on) is loud, not silently 187-dim
    monkeypatch.undo()
    with pytest.raises(NotImplementedError, match="tonnetz"):
        dl.load_data(settings=_app_config(tmp_path, flags=True))
    # process_file with an AppConfig-shaped settings object (data_loader.py:279-281)
    r = dl.process_file("/x/03-01-05-01-01-01-01.wav", {"angry"}, cfg.emotions, settings=cfg)
    assert r.sample is None and r.error.startswith("Failed to process file") and "not found" in r.error


def test_bounded_slabs_and_bisecting_retry(tmp_path):
    """jobs are processed in slabs bounded by a byte budget, and a failing batch is retried in
    halves so only the offending item is reported."""
    from ser_amd.data_loader import _bisecting, _job_slabs

    files = []
    for i, size in enumerate((10, 20, 30, 40, 50)):
        f = tmp_path / f"f{i}.bin"
        f.write_bytes(bytes(size))
        files.append((str(f), None, None))
    files.insert(2, (str(tmp_path / "missing.bin"), None, None))
    assert list(_job_slabs(files, 1000)) == [(0, 6)]
    assert list(_job_slabs(files, 60)) == [(0, 4), (4, 5), (5, 6)]
    assert list(_job_slabs(files, 5)) == [(i, i + 1) for i in range(6)]
    assert list(_job_slabs([], 5)) == []
    done, failed, calls = [], {}, []

    def run(sub):
        calls.append(list(sub))
        if 5 in sub or 11 in sub:
            raise ValueError(f"bad item in {sub}")
        done.extend(sub)

    _bisecting(list(range(16)), run, lambda i, e: failed.__setitem__(i, e))
    assert sorted(done) == [i for i in range(16) if i not in (5, 11)] and sorted(failed) == [5, 11]
    assert len(calls) <= 15   # log-depth, not one call per item


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
    from ser_amd import FeatureFlags

    settings = LoaderSettings(dataset_folder=tmp_path / "ravdess", feature_flags=FeatureFlags(tonnetz=False))
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
    quick = LoaderSettings(dataset_folder=tmp_path / "ravdess", feature_flags=FeatureFlags(tonnetz=False),
                           audio_read=settings.audio_read.__class__(max_retries=1, retry_delay_seconds=0.0))
    with pytest.raises(RuntimeError, match=r"3\.0% file failures exceeded configured limit 1\.0%"):
        load_data(settings=quick)
    matrix, errors = featurize_files(ordered[:3] + [str(tmp_path / "missing.wav")], settings=quick)
    assert [e is None for e in errors] == [True, True, True, False] and np.isnan(matrix[3]).all()
    assert_features_close(matrix[:3], want_rows[:3])

    Utt = _Utt
    utts = [Utt(f, lab) for f, lab in zip(ordered, want_labels)]
    utts[5] = Utt(ordered[5], want_labels[5], start=0.25, dur=1.0)
    broken = Utt(str(tmp_path / "gone.wav"), "happy")
    skipped = []

    def halves(*, samples, settings, logger):       # the reference's splitter call shape (data_loader.py:478-482)
        return samples[:17], samples[17:], {}

    out = load_checked_fast_data(utterances=utts[:16] + [broken] + utts[16:], settings=quick, split_utterances=halves,
                                 handle_sample_failure=lambda u, e: skipped.append(u.sample_id) or True)
    assert out[0].shape == (16, 187) and out[1].shape == (16, 187) and skipped == [broken.sample_id]
    seg_audio = read_audio_file(ordered[5], start_seconds=0.25, duration_seconds=1.0)[0]
    assert_features_close(out[0][5], ofp.extract_vector(seg_audio, 16000))
    with pytest.raises(FileNotFoundError):
        load_checked_fast_data(utterances=utts[:4] + [broken] + utts[16:], settings=quick, split_utterances=halves)
    # the same seam driven by an AppConfig-shaped object and the default splitter
    cfg = _app_config(tmp_path, flags=False)
    cfg.emotions = dict(settings.emotions)
    got = load_checked_fast_data(utterances=utts, settings=cfg)
    assert got[0].shape[1] == 187 and got[0].shape[0] + got[1].shape[0] == 32 and got[1].shape[0] == 13
    got2 = load_data(settings=cfg)       # 33 files, one junk: under cfg's 50 % failure limit
    assert got2[0].shape[0] + got2[1].shape[0] == 32
    # prepared-feature cache filled from the GPU features
    from ser_amd.prepared_cache import read_prepared_feature_payload, write_prepared_feature_payload

    p = write_prepared_feature_payload(tmp_path / "tmp" / "prepared-training-fast.npz", x_train=x_train,
                                       x_dev=np.empty((0, 187)), x_test=x_test, y_train=y_train, y_dev=[],
                                       y_test=y_test, metadata={"cache_namespace": "fast_features"})
    back = read_prepared_feature_payload(p, feature_shape=(32, 187))
    np.testing.assert_array_equal(back.x_train, x_train)
    assert back.y_test == y_test


def test_sorted_glob_and_label_pass_equal_the_per_file_forms(tmp_path):
    """``sorted_glob`` / ``labels_for_files`` are the loop-free forms of ``sorted(glob.glob(...))`` and
    ``label_for_file`` (reference: data_loader.py:356-366, 234-262): same files, same order, same messages."""
    import glob

    from ser_amd.data_loader import RAVDESS_EMOTIONS, label_for_file, labels_for_files, sorted_glob

    for actor in (1, 2, 10):
        d = tmp_path / f"Actor_{actor:02d}"
        d.mkdir()
        for emo in range(1, 9):
            (d / f"03-01-{emo:02d}-01-01-01-{actor:02d}.wav").touch()
        for odd in (".hidden.wav", "notes.txt", "weird.wav", "a-b.wav", "03-01-99-01-01-01-01.wav"):
            (d / odd).touch()
    (tmp_path / "Actor_not_a_directory").touch()
    (tmp_path / "Other").mkdir()
    (tmp_path / "Other" / "03-01-03-01-01-01-01.wav").touch()
    patterns = [tmp_path / "Actor_*" / "*.wav", tmp_path / "Actor_*" / ".*", tmp_path / "*" / "*.wav",
                tmp_path / "Actor_01" / "*.wav", tmp_path / "Actor_0[12]" / "03-01-0?-*.wav", tmp_path / "none*" / "*.wav",
                tmp_path / "**" / "*.wav", tmp_path / "Actor_01" / "notes.txt", tmp_path / "Actor_*"]
    for pattern in patterns:
        assert sorted_glob(str(pattern)) == sorted(glob.glob(str(pattern))), pattern
    files = sorted_glob(str(tmp_path / "*" / "*.wav"))
    observed = set(RAVDESS_EMOTIONS.values()) - {"neutral"}
    wanted, errors = [], []
    for f in files:
        label, error = label_for_file(f, observed, RAVDESS_EMOTIONS)
        if error is not None:
            errors.append(error)
        elif label is not None:
            wanted.append((f, label))
    assert labels_for_files(files, observed, RAVDESS_EMOTIONS) == (wanted, errors)
    assert len(errors) == 6 and len(wanted) == 3 * 7 + 1     # weird.wav, a-b.wav per actor; code 99 and neutral dropped silently
