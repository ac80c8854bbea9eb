"""Host-side contract tests (no GPU): the reference's own known answers for the boundary
types, windowing, pooling windows, audio preparation and error strings."""

import wave

import numpy as np
import pytest

import ser_amd
from ser_amd import (
    AudioReadConfig,
    EncodedSequence,
    FeatureBackend,
    FeatureFlags,
    PoolingWindow,
    VectorFeatureBackend,
    overlap_frame_mask,
)
from ser_amd.audio_io import AudioDecodeError, AudioIntegrityError, prepare_audio_buffer, read_audio_file
from ser_amd.handcrafted import HandcraftedBackend, sliding_window_bounds
from ser_amd.pool import temporal_pooling_windows, window_frame_ranges


def seq(rows, starts, ends, backend_id="handcrafted"):
    return EncodedSequence(np.asarray(rows, np.float32), np.asarray(starts, np.float64),
                           np.asarray(ends, np.float64), backend_id)


def test_feature_dims_and_backend_identity():
    # tests/suites/unit/repr/test_handcrafted_backend.py:13-24, integration/test_fast_inference.py:27-46
    assert FeatureFlags().width() == 193
    assert FeatureFlags(tonnetz=False).width() == 187
    b = HandcraftedBackend(feature_flags=FeatureFlags())
    assert b.backend_id == "handcrafted" and b.feature_dim == 193
    assert HandcraftedBackend().feature_dim == 193          # None means the reference's default (tonnetz on)
    assert HandcraftedBackend(feature_flags=FeatureFlags(False, True, False, True, False)).feature_dim == 19
    assert isinstance(b, FeatureBackend) and isinstance(b, VectorFeatureBackend)
    assert FeatureFlags(tonnetz=False).native_bits() == 15
    assert AudioReadConfig() == AudioReadConfig(max_retries=3, retry_delay_seconds=1.0)


def test_default_flags_are_the_references_and_loud(monkeypatch):
    """VERDICT r1 / ADVICE r1: ``feature_flags=None`` / ``settings=None`` / ``HandcraftedBackend()`` mean
    what they mean in the reference -- FeatureFlags() with tonnetz on, 193 dims (dsp.py:92,
    repr/handcrafted.py:38) -- and therefore raise NotImplementedError naming the way out, instead of
    silently returning 187 values.  ``None`` and ``FeatureFlags()`` agree.  One explicit opt-in
    (SERHIP_DEFAULT_FLAGS=fast187) selects the accelerated flags as the default."""
    from ser_amd import dsp, feature_extractor
    from ser_amd.data_loader import LoaderSettings, featurize_audio_jobs, resolve_loader_settings
    from ser_amd.types import coerce_feature_flags, default_feature_flags

    monkeypatch.delenv("SERHIP_DEFAULT_FLAGS", raising=False)
    x = np.linspace(-1, 1, 4000, dtype=np.float32)
    assert default_feature_flags() == FeatureFlags() and coerce_feature_flags(None).tonnetz
    for flags in (None, FeatureFlags()):
        with pytest.raises(NotImplementedError, match=r"FeatureFlags\(tonnetz=False\).*SERHIP_DEFAULT_FLAGS=fast187"):
            dsp.extract_feature_from_signal(x, 16000, feature_flags=flags)
        b = HandcraftedBackend(feature_flags=flags)
        assert b.feature_dim == 193
        for call in (lambda: b.encode_sequence(x, 16000), lambda: b.extract_vector(x, 16000),
                     lambda: b.extract_vectors([x], 16000)):
            with pytest.raises(NotImplementedError, match="tonnetz"):
                call()
    with pytest.raises(NotImplementedError, match="tonnetz"):
        feature_extractor.extract_feature_from_signal(x, 16000)            # settings=None
    with pytest.raises(NotImplementedError, match="tonnetz"):
        feature_extractor.extract_feature_from_signal(x, 16000, settings=feature_extractor.FeatureSettings())
    assert LoaderSettings().feature_flags == FeatureFlags() and resolve_loader_settings(None).feature_flags.tonnetz
    with pytest.raises(NotImplementedError, match="tonnetz"):
        featurize_audio_jobs([("a.wav", None, None)], feature_flags=FeatureFlags(), audio_read_config=AudioReadConfig())
    # input validation still comes first, as in the reference
    with pytest.raises(ValueError, match="no samples"):
        dsp.extract_feature_from_signal(np.zeros(0, np.float32), 16000)
    # a foreign flags object (the reference's own dataclass) is read by attribute
    class RefFlags:
        mfcc = chroma = mel = contrast = True
        tonnetz = False
    assert coerce_feature_flags(RefFlags()) == FeatureFlags(tonnetz=False)
    # the opt-in
    monkeypatch.setenv("SERHIP_DEFAULT_FLAGS", "fast187")
    assert default_feature_flags() == FeatureFlags(tonnetz=False)
    assert HandcraftedBackend().feature_dim == 187 and LoaderSettings().feature_flags.width() == 187
    monkeypatch.setenv("SERHIP_DEFAULT_FLAGS", "bogus")
    with pytest.raises(ValueError, match="SERHIP_DEFAULT_FLAGS"):
        default_feature_flags()


def test_backend_constructor_validation():
    # test_handcrafted_backend.py:112-117
    with pytest.raises(ValueError, match="frame_size_seconds"):
        HandcraftedBackend(frame_size_seconds=0)
    with pytest.raises(ValueError, match="frame_stride_seconds"):
        HandcraftedBackend(frame_stride_seconds=0)


def test_sliding_window_bounds_known_answer():
    # 10 samples @ 2 Hz, size 2 s, stride 1 s -> sizes [4,4,4,4,2] (test_feature_extractor.py:79-163)
    starts, ends = sliding_window_bounds(10, 2, 2, 1)
    assert starts.tolist() == [0, 2, 4, 6, 8] and ends.tolist() == [4, 6, 8, 10, 10]
    assert (starts / 2.0).tolist() == [0, 1, 2, 3, 4] and (ends / 2.0).tolist() == [2, 3, 4, 5, 5]
    s, e = sliding_window_bounds(69937, 16000, 3, 1)
    assert (e - s).tolist() == [48000, 48000, 37937, 21937, 5937]
    s, e = sliding_window_bounds(57_600_000, 16000, 3, 1)
    assert s.size == 3600 and (e - s)[-2:].tolist() == [32000, 16000]


def test_encoded_sequence_validation_messages():
    # tests/suites/unit/repr/test_backend_contract.py:95-114
    with pytest.raises(ValueError, match="frame_end_seconds length"):
        seq([[1, 2], [3, 4]], [0.0, 1.0], [1.0])
    with pytest.raises(ValueError, match="end_seconds > start_seconds"):
        seq([[1, 2]], [1.0], [1.0])
    with pytest.raises(ValueError, match="non-empty string"):
        seq([[1, 2]], [0.0], [1.0], backend_id="")
    with pytest.raises(ValueError, match="must be 2D"):
        seq([1, 2], [0.0], [1.0])
    with pytest.raises(ValueError, match="non-finite"):
        seq([[np.nan, 2]], [0.0], [1.0])
    with pytest.raises(ValueError, match="non-decreasing"):
        seq([[1, 2], [3, 4]], [1.0, 0.0], [2.0, 3.0])
    with pytest.raises(ValueError, match="at least one frame"):
        seq(np.zeros((0, 2)), [], [])


def test_pooling_window_validation():
    with pytest.raises(ValueError, match="finite"):
        PoolingWindow(0.0, float("inf"))
    with pytest.raises(ValueError, match="non-negative"):
        PoolingWindow(-0.1, 1.0)
    with pytest.raises(ValueError, match="greater than start_seconds"):
        PoolingWindow(1.0, 1.0)


def test_overlap_mask_and_ranges():
    # test_backend_contract.py:142-165, test_handcrafted_backend.py:66-91
    enc = seq([[0, 3], [2, 5], [4, 7], [6, 9], [8, 9]], [0, 1, 2, 3, 4], [2, 3, 4, 5, 5])
    assert overlap_frame_mask(enc, PoolingWindow(0.0, 1.0)).tolist() == [True, False, False, False, False]
    assert overlap_frame_mask(enc, PoolingWindow(2.0, 4.0)).tolist() == [False, True, True, True, False]
    with pytest.raises(ValueError, match="outside encoded sequence range"):
        overlap_frame_mask(enc, PoolingWindow(0.0, 6.0))
    gap = seq([[1], [2]], [0.0, 2.0], [1.0, 3.0])
    with pytest.raises(ValueError, match="does not overlap any encoded frames"):
        overlap_frame_mask(gap, PoolingWindow(1.2, 1.8))
    lo, hi = window_frame_ranges(enc, [PoolingWindow(0.0, 1.0), PoolingWindow(2.0, 4.0)])
    assert lo.tolist() == [0, 1] and hi.tolist() == [1, 4]
    # mask semantics equal the reference's boolean expression on random monotone timelines
    rng = np.random.default_rng(0)
    starts = np.sort(rng.random(40) * 10)
    ends = starts + 0.5 + rng.random(40)
    ends = np.maximum.accumulate(ends)
    enc2 = seq(np.zeros((40, 1)), starts, ends)
    for _ in range(50):
        a = rng.uniform(starts[0], ends[-1] - 0.01)
        b = rng.uniform(a + 0.001, ends[-1])
        ref = np.logical_and(ends > a, starts < b)
        if ref.any():
            np.testing.assert_array_equal(overlap_frame_mask(enc2, PoolingWindow(a, b)), ref)


def test_window_frame_ranges_vectorised_equals_per_window_walk():
    """All windows at once (two searchsorted calls) against the reference-shaped walk, window by window;
    a bad window raises the error of the FIRST bad one, as the reference's loop does."""
    from ser_amd.types import overlap_frame_range

    rng = np.random.default_rng(1)
    starts = np.sort(rng.random(60) * 20)
    ends = np.maximum.accumulate(starts + 0.3 + rng.random(60))
    enc = seq(np.zeros((60, 1)), starts, ends)
    wins = []
    for _ in range(200):
        a = rng.uniform(starts[0], ends[-1] - 0.01)
        wins.append(PoolingWindow(a, rng.uniform(a + 0.001, ends[-1])))
    lo, hi = window_frame_ranges(enc, wins)
    assert lo.dtype == hi.dtype == np.int32
    assert [(int(a), int(b)) for a, b in zip(lo, hi)] == [overlap_frame_range(enc, w) for w in wins]
    lo0, hi0 = window_frame_ranges(enc, [])
    assert lo0.size == hi0.size == 0
    gap = seq([[1], [2]], [0.0, 2.0], [1.0, 3.0])
    inside, hole, outside = PoolingWindow(0.1, 0.9), PoolingWindow(1.2, 1.8), PoolingWindow(0.0, 6.0)
    with pytest.raises(ValueError, match="does not overlap any encoded frames"):
        window_frame_ranges(gap, [inside, hole, outside])
    with pytest.raises(ValueError, match="outside encoded sequence range"):
        window_frame_ranges(gap, [inside, outside, hole])


def test_temporal_pooling_windows_known_answers():
    # tests/suites/unit/repr/test_pooling.py:78-138
    enc = seq(np.zeros((4, 2)), [0, 1, 2, 3], [1, 2, 3, 4])
    w = temporal_pooling_windows(enc, window_size_seconds=2.0, window_stride_seconds=1.5)
    assert [(x.start_seconds, x.end_seconds) for x in w] == [(0.0, 2.0), (1.5, 3.5), (2.0, 4.0)]
    short = seq(np.zeros((1, 2)), [0.0], [1.0])
    w = temporal_pooling_windows(short, window_size_seconds=3.0, window_stride_seconds=1.0)
    assert [(x.start_seconds, x.end_seconds) for x in w] == [(0.0, 1.0)]
    with pytest.raises(ValueError, match="window_size_seconds"):
        temporal_pooling_windows(enc, window_size_seconds=0.0, window_stride_seconds=1.0)
    with pytest.raises(ValueError, match="window_stride_seconds"):
        temporal_pooling_windows(enc, window_size_seconds=1.0, window_stride_seconds=-0.1)
    hour = seq(np.zeros((3600, 1)), np.arange(3600.0), np.minimum(np.arange(3600.0) + 3, 3600.0))
    assert len(temporal_pooling_windows(hour, window_size_seconds=10.0, window_stride_seconds=5.0)) == 719


def test_prepare_audio_buffer_known_answers():
    # tests/suites/unit/utils/test_audio_utils.py:14-27
    out = prepare_audio_buffer(np.asarray([[0.2, 0.0], [0.4, 0.0]], np.float32))
    assert out.dtype == np.float32
    np.testing.assert_allclose(out, [0.5, 1.0])
    with pytest.raises(OSError, match="contains no samples"):
        prepare_audio_buffer(np.zeros(0, np.float32))
    np.testing.assert_array_equal(prepare_audio_buffer(np.asarray([np.nan, np.inf, -2.0, 1.0], np.float32)),
                                  [0.0, 0.0, -1.0, 0.5])
    assert not prepare_audio_buffer(np.zeros(5, np.float32)).any()
    src = np.asarray([0.5, -0.25], np.float32)
    prepare_audio_buffer(src)
    np.testing.assert_array_equal(src, [0.5, -0.25])          # inputs are never mutated


def test_read_audio_file(tmp_path):
    from ser_amd.synth import ravdess_sine_clip, write_wav_pcm16

    clip = ravdess_sine_clip(1, 3)
    path = tmp_path / "Actor_01" / "03-01-03-01-01-01-01.wav"
    path.parent.mkdir()
    write_wav_pcm16(path, np.rint(clip * 32768).astype(np.int16), 16000)
    audio, sr = read_audio_file(str(path))
    assert sr == 16000 and audio.dtype == np.float32 and audio.shape == (24000,)
    np.testing.assert_array_equal(audio, prepare_audio_buffer(clip))
    seg, _ = read_audio_file(str(path), start_seconds=0.5, duration_seconds=0.25)
    assert seg.shape == (4000,)
    stereo = tmp_path / "stereo.wav"
    with wave.open(str(stereo), "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(8000)
        w.writeframes(np.asarray([[1000, 3000], [-2000, -2000]], "<i2").tobytes())
    a2, sr2 = read_audio_file(str(stereo))
    assert sr2 == 8000
    np.testing.assert_allclose(a2, [1.0, -1.0])
    with pytest.raises(FileNotFoundError, match="Audio file not found"):
        read_audio_file(str(tmp_path / "missing.wav"))
    with pytest.raises(ValueError, match="start_seconds"):
        read_audio_file(str(path), start_seconds=-1.0)
    with pytest.raises(ValueError, match="duration_seconds"):
        read_audio_file(str(path), duration_seconds=0.0)
    lfs = tmp_path / "pointer.wav"
    lfs.write_bytes(b"version https://git-lfs.github.com/spec/v1\noid sha256:0\nsize 1\n")
    with pytest.raises(AudioIntegrityError, match="Git LFS pointer"):
        read_audio_file(str(lfs))
    junk = tmp_path / "junk.wav"
    junk.write_bytes(b"not audio at all")
    import time

    t0 = time.perf_counter()   # a deterministic parser verdict is not retried: no sleeping
    with pytest.raises(AudioDecodeError, match="Error reading .*not a RIFF/WAVE, FLAC, AIFF or AU file"):
        read_audio_file(str(junk), audio_read_config=AudioReadConfig(max_retries=3, retry_delay_seconds=5.0))
    assert time.perf_counter() - t0 < 1.0
    adpcm = tmp_path / "adpcm.wav"
    import struct

    fmt = struct.pack("<HHIIHH", 2, 1, 16000, 8000, 256, 4)
    adpcm.write_bytes(b"RIFF" + struct.pack("<I", 36 + 8) + b"WAVEfmt " + struct.pack("<I", 16) + fmt +
                      b"data" + struct.pack("<I", 8) + bytes(8))
    with pytest.raises(AudioDecodeError, match="unsupported WAVE encoding \\(format 2, 4 bits\\)"):
        read_audio_file(str(adpcm), audio_read_config=AudioReadConfig(max_retries=3, retry_delay_seconds=5.0))


def test_audio_read_config_retries_transient_failures(tmp_path, monkeypatch):
    """What ``AudioReadConfig`` does here (reference loop: audio_utils.py:96-162): every failure that is not the
    parser's deterministic verdict -- an I/O error while reading, an empty payload -- is retried up to
    ``max_retries`` times with ``retry_delay_seconds`` between attempts; the last failure becomes
    ``AudioDecodeError("Error reading <path>")`` chained to it, except for segment reads, which re-raise
    the original exception like the reference (:121-131)."""
    import ser_amd.audio_io as aio
    from ser_amd.synth import ravdess_sine_clip, write_wav_pcm16

    path = tmp_path / "clip.wav"
    write_wav_pcm16(path, np.rint(ravdess_sine_clip(1, 2) * 32768).astype(np.int16), 16000)
    real_decode, calls, naps = aio.decode_file, [], []

    def flaky(p, **kw):
        calls.append(p)
        if len(calls) <= 2:
            raise OSError(5, "Input/output error")
        return real_decode(p, **kw)

    monkeypatch.setattr(aio, "decode_file", flaky)
    monkeypatch.setattr(aio.time, "sleep", naps.append)
    audio, rate = aio.read_audio_file(str(path), audio_read_config=AudioReadConfig(max_retries=3, retry_delay_seconds=0.25))
    assert rate == 16000 and audio.shape == (24000,) and len(calls) == 3 and naps == [0.25, 0.25]
    # one attempt fewer than failures: the reference's terminal error, chained to the last failure, no sleep after it
    calls.clear()
    naps.clear()
    with pytest.raises(AudioDecodeError, match=f"Error reading {path}") as info:
        aio.read_audio_file(str(path), audio_read_config=AudioReadConfig(max_retries=2, retry_delay_seconds=0.5))
    assert isinstance(info.value.__cause__, OSError) and len(calls) == 2 and naps == [0.5]
    # segment reads re-raise the original exception on the last attempt
    calls.clear()
    naps.clear()
    with pytest.raises(OSError, match="Input/output error"):
        aio.read_audio_file(str(path), start_seconds=0.1, audio_read_config=AudioReadConfig(max_retries=2, retry_delay_seconds=0.5))
    assert len(calls) == 2 and naps == [0.5]


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a GPU the product path must raise; it never routes through the CPU oracle."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    x = np.linspace(-1, 1, 4000, dtype=np.float32)
    fast = FeatureFlags(tonnetz=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ser_amd.extract_feature_from_signal(x, 16000, feature_flags=fast)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HandcraftedBackend(feature_flags=fast).encode_sequence(x, 16000)
    # input validation still happens first, with the reference's messages
    with pytest.raises(ValueError, match="no samples"):
        ser_amd.extract_feature_from_signal(np.zeros(0, np.float32), 16000, feature_flags=fast)
    with pytest.raises(ValueError, match="at least one sample"):
        HandcraftedBackend(feature_flags=fast).encode_sequence(np.zeros(0, np.float32), 16000)


def test_product_never_imports_the_oracle():
    from pathlib import Path

    root = Path(ser_amd.__file__).resolve().parent
    for path in list(root.rglob("*.py")) + list(root.rglob("*.hpp")) + list(root.rglob("*.hip")):
        text = path.read_text()
        assert "import oracle" not in text and "from oracle" not in text, path


def test_validate_signal_messages_and_extreme_values():
    """Input checks of dsp.py:85-95 (order and messages); huge but finite samples pass even though
    their sum of squares overflows the fast finiteness test."""
    from ser_amd.dsp import validate_signal

    ok = np.array([1e30, -3e38, 0.5], dtype=np.float32)
    assert validate_signal(ok, 16000) is not None
    with pytest.raises(ValueError, match="Sample rate must be a positive integer."):
        validate_signal(ok, 0)
    with pytest.raises(ValueError, match="mono"):
        validate_signal(np.zeros((2, 2), dtype=np.float32), 16000)
    with pytest.raises(ValueError, match="no samples"):
        validate_signal(np.zeros(0, dtype=np.float32), 16000)
    for bad in (np.nan, np.inf, -np.inf):
        x = np.zeros(1000, dtype=np.float32)
        x[777] = bad
        with pytest.raises(ValueError, match="not finite everywhere"):
            validate_signal(x, 16000)
