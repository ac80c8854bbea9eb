"""The overlapped ingest path (ser_amd/ingest.py + serhip_read_wav_batch / serhip_stat_files /
serhip_parallel_memcpy): the GPU-side counterpart of the reference's ``Pool.imap_unordered`` over files
(ser/_internal/data/data_loader.py:368-379).

CPU tests: the native reader sees what the Python reader (``audio_io.read_raw_clip``) sees -- same payload bytes,
same frame selection, same verdict on what it does not take.  GPU tests: the pipelined loader returns the
sequential loader's bits, a failing file is reported alone, a host buffer streamed in chunks gives the whole-buffer
result."""

import os
import struct

import numpy as np
import pytest


def _wav_bytes(fmt_code: int, bits: int, channels: int, rate: int, payload: bytes, *, junk: bytes = b"abcd") -> bytes:
    block = channels * bits // 8
    fmt = struct.pack("<HHIIHH", fmt_code, channels, rate, rate * block, block, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if junk is not None:
        body += b"LIST" + struct.pack("<I", len(junk)) + junk + (b"\0" if len(junk) & 1 else b"")
    body += b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", len(body)) + body


def _corpus(tmp_path, rng):
    """[(path, start, duration)] over every kind of file the reader meets; returns jobs and the index of each kind"""
    files = {}

    def put(name, data):
        p = tmp_path / name
        p.write_bytes(data)
        files[name] = str(p)
        return str(p)

    put("mono16.wav", _wav_bytes(1, 16, 1, 16000, rng.integers(-30000, 30000, 5000, dtype=np.int16).astype("<i2").tobytes()))
    put("stereo16.wav", _wav_bytes(1, 16, 2, 8000, rng.integers(-30000, 30000, (999, 2), dtype=np.int16).astype("<i2").tobytes()))
    put("f32.wav", _wav_bytes(3, 32, 1, 22050, rng.standard_normal(777).astype("<f4").tobytes()))
    # a LIST chunk of 6 bytes puts the float32 payload on a 2-byte boundary: the reader slides it to a 4-byte one
    put("f32_misaligned.wav", _wav_bytes(3, 32, 2, 22050, rng.standard_normal((300, 2)).astype("<f4").tobytes(), junk=b"abcdef"))
    put("nolist.wav", _wav_bytes(1, 16, 1, 16000, rng.integers(-100, 100, 2500, dtype=np.int16).astype("<i2").tobytes(), junk=None))
    put("pcm24.wav", _wav_bytes(1, 24, 1, 16000, bytes(300)))
    put("lfs.wav", b"version https://git-lfs.github.com/spec/v1\noid sha256:0\nsize 1\n")
    put("junk.wav", b"not audio at all")
    put("truncated.wav", _wav_bytes(1, 16, 1, 16000, bytes(2000))[:-500])     # data chunk runs past the end of the file
    put("empty.wav", _wav_bytes(1, 16, 1, 16000, b""))
    files["missing.wav"] = str(tmp_path / "missing.wav")
    (tmp_path / "dir.wav").mkdir()
    files["dir.wav"] = str(tmp_path / "dir.wav")
    jobs = [(files[n], None, None) for n in files]
    jobs += [(files["stereo16.wav"], 0.05, 0.01), (files["mono16.wav"], 0.1, None), (files["mono16.wav"], None, 0.05),
             (files["mono16.wav"], 10.0, None), (files["f32.wav"], 0.01, 0.02)]
    return jobs


def test_native_reader_agrees_with_the_python_reader(tmp_path, native_lib):
    from ser_amd import ingest
    from ser_amd.audio_io import read_raw_clip

    rng = np.random.default_rng(5)
    jobs = _corpus(tmp_path, rng)
    paths = [j[0] for j in jobs]
    sizes = ingest.stat_files(paths, threads=4)
    for p, s in zip(paths, sizes):
        assert s == (os.path.getsize(p) if os.path.isfile(p) else -1)
    room = (np.maximum(sizes, 0) + 15) // 16 * 16
    offs = np.concatenate(([0], np.cumsum(room)[:-1]))
    slab = np.full(int(room.sum()) + 64, 0xEE, dtype=np.uint8)
    for threads in (1, 5):
        clips = ingest.read_wav_batch(paths, slab.ctypes.data, slab.nbytes, offs, room, start_seconds=[j[1] for j in jobs],
                                      duration_seconds=[j[2] for j in jobs], threads=threads)
        for (path, start, dur), c in zip(jobs, clips):
            try:
                want = read_raw_clip(path, start_seconds=start, duration_seconds=dur)
            except Exception:
                want = None
            if want is None or want.frames == 0:     # the Python loader falls back to the host reader for these
                assert c["status"] != 0, path
                continue
            assert c["status"] == 0, (path, c)
            assert (ingest.SAMPLE_FORMAT_NAMES[int(c["sample_format"])], int(c["channels"]), int(c["frames"]), int(c["sample_rate"])) == \
                (want.sample_format, want.channels, want.frames, want.sample_rate)
            ssize = 2 if want.sample_format == "s16" else 4
            assert c["raw_offset"] % ssize == 0
            got = slab[int(c["raw_offset"]): int(c["raw_offset"]) + len(want.payload)]
            assert got.tobytes() == bytes(want.payload), path
    status = {os.path.basename(p) + (f"@{s},{d}" if (s, d) != (None, None) else ""): int(c["status"]) for (p, s, d), c in zip(jobs, clips)}
    assert status["pcm24.wav"] == 1 and status["missing.wav"] == 2 and status["dir.wav"] == 2 and status["lfs.wav"] == 3
    assert status["junk.wav"] == 4 and status["empty.wav"] == 6 and status["mono16.wav@10.0,None"] == 6
    assert status["truncated.wav"] == 0
    # a region smaller than the file: reported, nothing written past the region
    small = np.full(64, 0xEE, dtype=np.uint8)
    c = ingest.read_wav_batch(paths[:1], small.ctypes.data, small.nbytes, np.zeros(1, np.int64), np.array([32]), threads=1)
    assert int(c["status"][0]) == 5 and (small == 0xEE).all()
    # regions outside dst are a caller error
    from ser_amd._native import SerhipError

    with pytest.raises(SerhipError, match="lies outside dst"):
        ingest.read_wav_batch(paths[:1], small.ctypes.data, small.nbytes, np.array([48]), np.array([32]), threads=1)


def test_parallel_memcpy_and_slab_plan(native_lib):
    from ser_amd import ingest

    rng = np.random.default_rng(1)
    src = rng.integers(0, 255, size=(9 << 20) + 13, dtype=np.uint8)
    for threads in (1, 3, 8):
        dst = np.zeros_like(src)
        ingest.parallel_memcpy(dst.ctypes.data, src.ctypes.data, src.nbytes, threads)
        assert np.array_equal(dst, src)
    sizes = np.array([100, 100, -1, 5000, 100, 100, 100], dtype=np.int64)     # 112-byte rooms; one missing; one oversize
    assert ingest.plan_slabs(sizes, 300) == [(0, 3), (3, 4), (4, 6), (6, 7)]
    assert ingest.plan_slabs(sizes, 1 << 20) == [(0, 7)]
    assert ingest.plan_slabs(np.zeros(0, np.int64), 300) == []
    blob, offsets = ingest.encode_paths(["a", "bcd", "éf"])
    assert blob == b"a\0bcd\0\xc3\xa9f\0" and offsets.tolist() == [0, 2, 6]


def test_gather_copies_arrays_and_finds_non_finite_samples(native_lib):
    """``serhip_gather_f32``: many small arrays, an array longer than one work item, empty arrays, NaN / infinity at the
    first and last sample of a piece and of a test block; regions outside the destination are refused."""
    from ser_amd import _native, ingest

    rng = np.random.default_rng(31)
    lens = [0, 1, 7, 8191, 8192, 8193, 48000, (1 << 20) + 5, 3, 0, (2 << 20) + 8192]
    arrays = [rng.standard_normal(n).astype(np.float32) for n in lens]
    off = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int64) + 3     # three floats of slack in front
    dst = np.full(int(sum(lens)) + 5, -7.0, dtype=np.float32)

    def run(threads):
        dst[:] = -7.0
        ptrs = np.asarray([a.ctypes.data if a.size else 0 for a in arrays], dtype=np.uint64)
        return ingest.gather_f32(dst.ctypes.data, dst.size, ptrs, np.asarray(lens), off, threads)

    for threads in (1, 5):
        assert run(threads) is True
        assert np.array_equal(dst[3:-2], np.concatenate(arrays)) and (dst[:3] == -7.0).all() and (dst[-2:] == -7.0).all()
    for which, at, value in ((1, 0, np.nan), (4, 8191, np.inf), (5, 8192, -np.inf), (7, 1 << 20, np.nan), (7, (1 << 20) - 1, np.inf),
                             (10, (2 << 20) + 8191, np.nan), (6, 17, np.float32(np.nan))):
        keep = arrays[which][at]
        arrays[which][at] = value
        assert run(4) is False, (which, at)
        assert np.array_equal(dst[3:-2], np.concatenate(arrays), equal_nan=True)   # the copy is complete either way
        arrays[which][at] = keep
    assert run(4) is True
    # the largest finite value and denormals are finite
    arrays[2][:3] = [np.finfo(np.float32).max, np.finfo(np.float32).tiny / 4, -0.0]
    assert run(3) is True
    with pytest.raises(_native.SerhipError, match="outside dst"):
        ingest.gather_f32(dst.ctypes.data, 10, np.asarray([arrays[3].ctypes.data], dtype=np.uint64), np.asarray([8191]), np.asarray([4]))
    with pytest.raises(_native.SerhipError, match="outside dst"):
        ingest.gather_f32(dst.ctypes.data, dst.size, np.asarray([arrays[3].ctypes.data], dtype=np.uint64), np.asarray([8191]), np.asarray([-1]))
    with pytest.raises(_native.SerhipError, match="NULL"):
        ingest.gather_f32(dst.ctypes.data, dst.size, np.asarray([0], dtype=np.uint64), np.asarray([4]), np.asarray([0]))
    assert ingest.gather_f32(dst.ctypes.data, dst.size, np.zeros(0, dtype=np.uint64), np.zeros(0), np.zeros(0)) is True


def test_take_rows_is_fancy_indexing(native_lib, monkeypatch):
    """``ingest.take_rows`` (the two halves of a train / test split copied by the memcpy team) returns ``matrix[index]``:
    permutations, repeats, negative indices, empty selections; anything it is not built for goes to numpy itself."""
    from ser_amd import ingest

    rng = np.random.default_rng(43)
    m = rng.standard_normal((5000, 187))
    m[17, 3] = np.nan           # a failed row travels like any other
    monkeypatch.setattr(ingest, "TAKE_ROWS_TEAM_BYTES", 1)
    for index in (rng.permutation(5000)[:3750], np.asarray([0, 0, 4999, -1, -5000, 17]), np.zeros(0, dtype=np.int64),
                  rng.integers(0, 5000, 9000).astype(np.int32)):
        got = ingest.take_rows(m, index, threads=3)
        assert got.flags.c_contiguous and got.flags.owndata and np.array_equal(got, m[index], equal_nan=True)
    with pytest.raises(IndexError):
        ingest.take_rows(m, np.asarray([5000]))
    with pytest.raises(IndexError):
        ingest.take_rows(m, np.asarray([-5001]))
    assert np.array_equal(ingest.take_rows(m.astype(np.float32), np.asarray([1, 2])), m.astype(np.float32)[[1, 2]])   # numpy's path
    assert np.array_equal(ingest.take_rows(m[:, ::2], np.asarray([1, 2])), m[:, ::2][[1, 2]])
    mask = np.zeros(5000, dtype=bool)
    mask[[4, 9]] = True
    assert np.array_equal(ingest.take_rows(m, mask), m[mask])


def test_bisect_raises_device_and_config_errors_at_once():
    """only per-clip / size-dependent failures are bisected."""
    from ser_amd._native import SerhipError
    from ser_amd.data_loader import _bisecting

    calls = []

    def run_factory(err):
        def run(sub):
            calls.append(list(sub))
            raise err
        return run

    for err in (SerhipError(-2, "hipErrorIllegalAddress"), NotImplementedError("tonnetz"), RuntimeError("HIP error: an illegal memory access")):
        calls.clear()
        with pytest.raises(type(err)):
            _bisecting(list(range(8)), run_factory(err), lambda i, e: None)
        assert calls == [list(range(8))]
    reported = []
    calls.clear()

    def run(sub):
        calls.append(list(sub))
        if 5 in sub:
            raise SerhipError(-1, "segment 5 is empty")
    _bisecting(list(range(8)), run, lambda i, e: reported.append(i))
    assert reported == [5] and len(calls) == 7
    calls.clear()
    reported.clear()

    def run_oom(sub):
        calls.append(list(sub))
        if len(sub) > 2:
            raise RuntimeError("HIP out of memory. Tried to allocate 1 GiB")
    _bisecting(list(range(8)), run_oom, lambda i, e: reported.append(i))
    assert reported == [] and [len(c) for c in calls] == [8, 4, 2, 2, 4, 2, 2]


# ------------------------------------------------------------------------------------------------------- GPU
def _dataset(tmp_path, rng, n=40):
    """n readable clips of three kinds (PCM16 mono 16 kHz, PCM16 stereo 16 kHz, float32 mono 22.05 kHz) and lengths
    0.4-1.6 s, plus the unreadable ones of _corpus in between"""
    jobs = []
    for i in range(n):
        kind = i % 3
        n_samples = int(rng.integers(6400, 25600))
        if kind == 0:
            data = _wav_bytes(1, 16, 1, 16000, (rng.standard_normal(n_samples) * 6000).astype("<i2").tobytes())
        elif kind == 1:
            data = _wav_bytes(1, 16, 2, 16000, (rng.standard_normal((n_samples, 2)) * 6000).astype("<i2").tobytes())
        else:
            data = _wav_bytes(3, 32, 1, 22050, rng.standard_normal(n_samples).astype("<f4").tobytes())
        p = tmp_path / f"clip{i:03d}.wav"
        p.write_bytes(data)
        jobs.append((str(p), None, None))
    return jobs


@pytest.mark.gpu
def test_pipelined_loader_equals_sequential_and_isolates_failures(tmp_path, monkeypatch):
    import torch

    from ser_amd import AudioReadConfig, FeatureFlags
    from ser_amd.audio_io import AudioDecodeError, AudioIntegrityError
    from ser_amd.data_loader import featurize_audio_jobs_device

    rng = np.random.default_rng(17)
    good = _dataset(tmp_path, rng)
    bad_dir = tmp_path / "bad"
    bad_dir.mkdir()
    bad = _corpus(bad_dir, rng)
    # bad files spread over the slabs: after every fourth good one
    jobs, k = [], 0
    for i, job in enumerate(good):
        jobs.append(job)
        if i % 4 == 3 and k < len(bad):
            jobs.append(bad[k])
            k += 1
    jobs += bad[k:]
    flags = FeatureFlags(tonnetz=False)
    cfg = AudioReadConfig(max_retries=1, retry_delay_seconds=0.0)
    seq, seq_err = featurize_audio_jobs_device(jobs, feature_flags=flags, audio_read_config=cfg, workers=4, pipelined=False)
    for slab_bytes in (1 << 20, 64 << 20):     # ~10 slabs (the floor of 1 MiB applies), then a single one
        monkeypatch.setenv("SERHIP_SLAB_BYTES", str(slab_bytes))
        stats = {}
        pip, pip_err = featurize_audio_jobs_device(jobs, feature_flags=flags, audio_read_config=cfg, workers=4, pipelined=True,
                                                   stats=stats)
        torch.cuda.synchronize()
        assert stats["slabs"] >= (2 if slab_bytes == 1 << 20 else 1)
        a, b = seq.cpu().numpy(), pip.cpu().numpy()
        assert np.array_equal(np.isnan(a), np.isnan(b))
        np.testing.assert_array_equal(np.nan_to_num(a), np.nan_to_num(b))          # bit-equal rows
        for (path, _s, _d), e_seq, e_pip, row in zip(jobs, seq_err, pip_err, b):
            assert type(e_seq) is type(e_pip), (path, e_seq, e_pip)
            assert (e_pip is None) == bool(np.isfinite(row).all()), path
    names = {os.path.basename(p): e for (p, s, d), e in zip(jobs, pip_err) if (s, d) == (None, None)}
    assert isinstance(names["missing.wav"], FileNotFoundError) and isinstance(names["lfs.wav"], AudioIntegrityError)
    assert isinstance(names["junk.wav"], AudioDecodeError) and isinstance(names["empty.wav"], OSError)
    assert names["pcm24.wav"] is None and names["truncated.wav"] is None     # host-decoded / cut at the end of the file
    assert all(e is None for n, e in names.items() if n.startswith("clip"))
    # the 8 kHz stereo file fails for the reference's reason (spectral_contrast needs sr > 12 800), alone
    assert "stereo16.wav" in names and names["stereo16.wav"] is not None


@pytest.mark.gpu
def test_pipelined_loader_overlaps_and_reuses_its_buffers(tmp_path, monkeypatch):
    """Two calls in a row on the same engine (the staging buffers are kept), the second on a smaller corpus."""
    import torch

    from ser_amd import AudioReadConfig, FeatureFlags
    from ser_amd.data_loader import featurize_audio_jobs_device

    rng = np.random.default_rng(23)
    jobs = _dataset(tmp_path, rng, n=30)
    flags = FeatureFlags(tonnetz=False)
    cfg = AudioReadConfig(max_retries=1, retry_delay_seconds=0.0)
    monkeypatch.setenv("SERHIP_SLAB_BYTES", str(1 << 20))
    want, _ = featurize_audio_jobs_device(jobs, feature_flags=flags, audio_read_config=cfg, pipelined=False)
    for subset in (jobs, jobs[:7], jobs):
        got, errs = featurize_audio_jobs_device(subset, feature_flags=flags, audio_read_config=cfg, pipelined=True)
        torch.cuda.synchronize()
        assert all(e is None for e in errs)
        np.testing.assert_array_equal(got.cpu().numpy(), want[: len(subset)].cpu().numpy())


@pytest.mark.gpu
def test_pipelined_loader_sends_rows_to_the_host_slab_by_slab(tmp_path, monkeypatch):
    """``to_host=True``: the float64 numpy matrix assembled from per-slab copies on the way (``ingest._RowsToHost``)
    is the widened device matrix -- failed rows NaN, host-decoded rows in place, a smaller second call through the
    kept pinned landing area, a result over the limit through the end-of-call copy."""
    from ser_amd import AudioReadConfig, FeatureFlags, ingest
    from ser_amd.data_loader import featurize_audio_jobs, featurize_audio_jobs_device

    rng = np.random.default_rng(29)
    jobs = _dataset(tmp_path, rng, n=30)
    bad_dir = tmp_path / "bad"
    bad_dir.mkdir()
    bad = _corpus(bad_dir, rng)
    jobs = jobs[:11] + bad[:4] + jobs[11:] + bad[4:]
    flags = FeatureFlags(tonnetz=False)
    cfg = AudioReadConfig(max_retries=1, retry_delay_seconds=0.0)
    monkeypatch.setenv("SERHIP_SLAB_BYTES", str(1 << 20))
    dev, dev_err = featurize_audio_jobs_device(jobs, feature_flags=flags, audio_read_config=cfg, workers=4, pipelined=False)
    want = dev.double().cpu().numpy()
    assert np.isnan(want).any() and np.isfinite(want).any()
    for subset in (len(jobs), 9, len(jobs)):
        stats = {}
        got, errs = featurize_audio_jobs_device(jobs[:subset], feature_flags=flags, audio_read_config=cfg, workers=4,
                                                pipelined=True, stats=stats, to_host=True)
        assert isinstance(got, np.ndarray) and got.dtype == np.float64 and got.flags.c_contiguous and got.flags.writeable
        assert stats["slabs"] >= (2 if subset > 9 else 1)
        assert np.array_equal(got, want[:subset], equal_nan=True)
        assert [type(e) for e in errs] == [type(e) for e in dev_err[:subset]]
    # the sequential form and the public wrapper give the same array
    seq, _ = featurize_audio_jobs_device(jobs, feature_flags=flags, audio_read_config=cfg, workers=4, pipelined=False, to_host=True)
    assert np.array_equal(seq, want, equal_nan=True)
    pub, _ = featurize_audio_jobs(jobs, feature_flags=flags, audio_read_config=cfg, workers=4)
    assert np.array_equal(pub, want, equal_nan=True)
    # over the limit: no landing area is made for it, the matrix comes back after the call
    monkeypatch.setattr(ingest, "ROWS_TO_HOST_LIMIT", 1024)
    big, _ = featurize_audio_jobs_device(jobs, feature_flags=flags, audio_read_config=cfg, workers=4, pipelined=True, to_host=True)
    assert np.array_equal(big, want, equal_nan=True)
    # nothing to do
    none, errs = featurize_audio_jobs_device([], feature_flags=flags, audio_read_config=cfg, to_host=True)
    assert none.shape == (0, 187) and none.dtype == np.float64 and errs == []


@pytest.mark.gpu
def test_pipelined_loader_with_nothing_readable(tmp_path):
    """Every file missing: per-file FileNotFoundError, NaN rows, no exception from the slab machinery (a slab of zero bytes)."""
    import torch

    from ser_amd import AudioReadConfig, FeatureFlags
    from ser_amd.data_loader import featurize_audio_jobs_device

    jobs = [(str(tmp_path / f"nope{i}.wav"), None, None) for i in range(5)]
    m, errs = featurize_audio_jobs_device(jobs, feature_flags=FeatureFlags(tonnetz=False),
                                          audio_read_config=AudioReadConfig(max_retries=1, retry_delay_seconds=0.0), pipelined=True)
    torch.cuda.synchronize()
    assert all(isinstance(e, FileNotFoundError) for e in errs) and bool(torch.isnan(m).all()) and tuple(m.shape) == (5, 187)


@pytest.mark.gpu
def test_host_buffer_streams_in_chunks_with_the_whole_buffer_result():
    import torch

    from ser_amd import FeatureFlags
    from ser_amd.engine import get_engine
    from ser_amd.ingest import features_from_host
    from ser_amd.synth import harmonic_clips

    eng = get_engine()
    flags = FeatureFlags(tonnetz=False)
    clips = harmonic_clips(np.random.default_rng(2), 24, 48000, 16000)
    audio = np.ascontiguousarray(clips.reshape(-1))
    # ragged, overlapping, non-decreasing segments: windows of 3 s at a 1 s stride plus whole clips
    off = np.sort(np.concatenate([np.arange(0, audio.size - 48000, 16000), np.arange(24) * 48000])).astype(np.int64)
    length = np.minimum(48000, audio.size - off)
    want = eng.features(torch.from_numpy(audio).to(eng.device), off, length, 16000, flags)
    for chunk_bytes in (1 << 20, 3 << 19, 64 << 20):
        stats = {}
        got = features_from_host(eng, audio, off, length, 16000, flags, chunk_bytes=chunk_bytes, stats=stats)
        torch.cuda.synchronize()
        assert stats["chunks"] >= (3 if chunk_bytes < (4 << 20) else 1)
        np.testing.assert_array_equal(got.cpu().numpy(), want.cpu().numpy())
    with pytest.raises(ValueError, match="outside the audio buffer"):
        features_from_host(eng, audio, np.array([audio.size - 10]), np.array([48000]), 16000, flags)


@pytest.mark.gpu
def test_host_clips_are_gathered_and_checked_by_the_staging_threads(monkeypatch):
    """Clips held as separate numpy arrays (``HandcraftedBackend.extract_vectors`` on a large batch) and one long
    recording (``encode_sequence``): no concatenation and no finite test on the host, same bits as the plain path, the
    reference's ValueError for a NaN or an infinity wherever it sits."""
    import torch

    from ser_amd import FeatureFlags, HandcraftedBackend
    from ser_amd.engine import FeatureEngine, get_engine
    from ser_amd.ingest import features_from_host, features_from_host_arrays
    from ser_amd.synth import harmonic_clips

    eng = get_engine()
    flags = FeatureFlags(tonnetz=False)
    rng = np.random.default_rng(37)
    base = harmonic_clips(rng, 40, 48000, 16000)
    lens = rng.integers(3000, 48000, 40)
    clips = [np.ascontiguousarray(base[i, : lens[i]]) for i in range(40)]
    clips[3] = clips[3].astype(np.float64)                  # converted on the way
    clips[5] = np.repeat(clips[5], 2)[::2]                  # a strided view
    clips[8] = clips[8][:700]                               # the generic (n_fft < 2048) path
    packed = np.concatenate([np.asarray(c, dtype=np.float32) for c in clips])
    sizes = np.asarray([c.size for c in clips], dtype=np.int64)
    offs = np.cumsum(sizes) - sizes
    want = eng.features(torch.from_numpy(packed).to(eng.device), offs, sizes, 16000, flags).cpu().numpy()
    for chunk_bytes in (1 << 20, 64 << 20):
        stats = {}
        got = features_from_host_arrays(eng, clips, 16000, flags, chunk_bytes=chunk_bytes, stats=stats)
        assert stats["chunks"] >= (4 if chunk_bytes == 1 << 20 else 1)
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    for which, at, value in ((0, 0, np.nan), (17, 100, np.inf), (39, -1, -np.inf), (3, 5, 1e300)):   # 1e300 is inf as float32
        keep = clips[which][at]
        clips[which][at] = value
        with np.errstate(over="ignore"), pytest.raises(ValueError, match="not finite everywhere"):
            features_from_host_arrays(eng, clips, 16000, flags, chunk_bytes=1 << 20)
        clips[which][at] = keep
    np.testing.assert_array_equal(features_from_host_arrays(eng, clips, 16000, flags).cpu().numpy(), want)   # and usable again
    with pytest.raises(ValueError, match="no samples"):
        features_from_host_arrays(eng, clips[:2] + [np.zeros(0, dtype=np.float32)], 16000, flags)

    # the backend: a batch over the threshold takes this path, and returns what the concatenating path returns
    backend = HandcraftedBackend(feature_flags=flags)
    monkeypatch.setenv("SERHIP_INGEST", "sequential")
    plain = backend.extract_vectors(clips, 16000)
    monkeypatch.delenv("SERHIP_INGEST")
    monkeypatch.setattr(FeatureEngine, "STREAM_FROM_HOST_BYTES", 1 << 16)
    staged = backend.extract_vectors(clips, 16000)
    assert staged.dtype == np.float64 and np.array_equal(staged, plain) and np.array_equal(staged, want.astype(np.float64))
    clips[11][7] = np.nan
    with pytest.raises(ValueError, match="not finite everywhere"):
        backend.extract_vectors(clips, 16000)
    clips[11][7] = 0.0
    with pytest.raises(ValueError, match="mono"):
        backend.extract_vectors(clips[:4] + [np.zeros((2, 2000), dtype=np.float32)], 16000)

    # one long recording: windows by descriptor, the finite test rides on the staging copy
    recording = np.ascontiguousarray(base[:12].reshape(-1))
    seq_plain_env = {"SERHIP_INGEST": "sequential"}
    monkeypatch.setenv("SERHIP_INGEST", "sequential")
    enc_plain = backend.encode_sequence(recording, 16000)
    monkeypatch.delenv("SERHIP_INGEST")
    enc = backend.encode_sequence(recording, 16000)
    assert np.array_equal(enc.embeddings, enc_plain.embeddings) and np.array_equal(enc.frame_start_seconds, enc_plain.frame_start_seconds)
    for at in (0, recording.size // 2, recording.size - 1):
        keep = recording[at]
        recording[at] = np.inf
        with pytest.raises(ValueError, match="not finite everywhere"):
            backend.encode_sequence(recording, 16000)
        recording[at] = keep
    # a stretch no window covers (segments that leave a gap and a tail) is tested too
    off2 = np.asarray([0, 48000, 200000], dtype=np.int64)
    len2 = np.asarray([48000, 48000, 48000], dtype=np.int64)
    ok = features_from_host(eng, recording, off2, len2, 16000, flags, check_finite=True, chunk_bytes=1 << 20)
    np.testing.assert_array_equal(ok.cpu().numpy(), eng.features(torch.from_numpy(recording).to(eng.device), off2, len2, 16000, flags).cpu().numpy())
    for at in (150000, recording.size - 5):
        keep = recording[at]
        recording[at] = np.nan
        with pytest.raises(ValueError, match="not finite everywhere"):
            features_from_host(eng, recording, off2, len2, 16000, flags, check_finite=True, chunk_bytes=1 << 20)
        features_from_host(eng, recording, off2, len2, 16000, flags, chunk_bytes=1 << 20)   # not asked: not tested
        recording[at] = keep
    del seq_plain_env


@pytest.mark.gpu
def test_staged_matrix_to_host_equals_the_plain_copy(monkeypatch):
    """``FeatureEngine.to_host_float64`` (pinned chunks + memcpy team) returns what ``.double().cpu()`` returns:
    every chunk boundary, a last partial chunk, float32 and float64 inputs, NaN rows (failed files) kept."""
    import torch

    from ser_amd.engine import FeatureEngine, get_engine

    eng = get_engine()
    gen = torch.Generator(device=eng.device).manual_seed(5)
    monkeypatch.setattr(FeatureEngine, "STAGED_TO_HOST_BYTES", 1 << 16)
    for rows, dtype in ((46080, torch.float32), (7001, torch.float64), (12, torch.float32), (0, torch.float32)):
        m = torch.randn((rows, 187), generator=gen, device=eng.device, dtype=dtype)
        if rows > 5:
            m[3] = float("nan")
        got = eng.to_host_float64(m)
        want = m.double().cpu().numpy()
        assert got.dtype == np.float64 and got.shape == want.shape and got.flags.writeable and got.flags.c_contiguous
        assert np.array_equal(got, want, equal_nan=True)
    # small chunks: many trips round the pinned ring
    monkeypatch.setattr(FeatureEngine, "STAGED_CHUNK_BYTES", (1 << 20) + 4096)
    m = torch.randn((30011, 187), generator=gen, device=eng.device)
    assert np.array_equal(eng.to_host_float64(m), m.double().cpu().numpy())
    # and the other direction (FeatureEngine.to_device_staged: what the classifier's batch predict uploads with)
    rng = np.random.default_rng(6)
    for shape, dtype in (((30011, 187), np.float64), ((7001, 187), np.float32), ((3, 5), np.float64), ((0, 187), np.float64),
                         ((1 << 20) + 3, np.int16)):
        a = (rng.standard_normal(shape) * 100).astype(dtype)
        dev = eng.to_device_staged(a)
        assert dev.device == eng.device and tuple(dev.shape) == a.shape and np.array_equal(dev.cpu().numpy(), a)
    strided = rng.standard_normal((9000, 374))[:, ::2]
    assert np.array_equal(eng.to_device_staged(strided).cpu().numpy(), strided)


@pytest.mark.gpu
def test_loader_takes_flac_aiff_and_au_beside_wave(tmp_path, monkeypatch):
    """The same PCM as WAV, FLAC, AIFF and AU (and A-law WAV: a lossy cousin, only finite-checked) through both
    loaders: rows of the lossless containers are bit-equal to the WAV row -- they reach the kernels through the host
    decoders (``audio_io.decode_bytes``), several per slab on the decode team."""
    import audioop
    import struct
    import wave

    import torch

    from flac_writer import encode_stream
    from ser_amd import AudioReadConfig, FeatureFlags
    from ser_amd.data_loader import featurize_audio_jobs_device

    rng = np.random.default_rng(23)
    jobs, kinds = [], []
    for k in range(6):
        n = 24000 + 512 * k
        t = np.arange(n)
        pcm = np.rint(9000 * np.sin(2 * np.pi * (200 + 30 * k) * t / 16000) + rng.integers(-300, 300, n)).astype(np.int64)[:, None]
        with wave.open(str(tmp_path / f"c{k}.wav"), "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes(pcm.astype("<i2").tobytes())
        frames = [{"block": 4096, "subframes": [{"type": "fixed", "order": 2, "partition_order": 3}]}] * (n // 4096)
        if n % 4096:
            frames = frames + [{"block": n % 4096, "subframes": [{"type": "fixed", "order": 1}]}]
        (tmp_path / f"c{k}.flac").write_bytes(encode_stream(pcm, bits=16, rate=16000, frames=frames))
        comm = struct.pack(">hIh", 1, n, 16) + struct.pack(">HQ", 16383 + 13, 16000 << 50)
        ssnd = struct.pack(">II", 0, 0) + pcm.astype(">i2").tobytes()
        body = b"AIFF" + b"COMM" + struct.pack(">I", len(comm)) + comm + b"SSND" + struct.pack(">I", len(ssnd)) + ssnd
        (tmp_path / f"c{k}.aiff").write_bytes(b"FORM" + struct.pack(">I", len(body)) + body)
        (tmp_path / f"c{k}.au").write_bytes(b".snd" + struct.pack(">IIIII", 24, 2 * n, 3, 16000, 1) + pcm.astype(">i2").tobytes())
        law = audioop.lin2alaw(pcm.astype("<i2").tobytes(), 2)
        fmt = struct.pack("<HHIIHH", 6, 1, 16000, 16000, 1, 8)
        chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(law)) + law
        (tmp_path / f"c{k}_alaw.wav").write_bytes(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)
        for kind in ("wav", "flac", "aiff", "au", "alaw"):
            name = f"c{k}_alaw.wav" if kind == "alaw" else f"c{k}.{kind}"
            jobs.append((str(tmp_path / name), None, None))
            kinds.append((k, kind))
    flags = FeatureFlags(tonnetz=False)
    cfg = AudioReadConfig(max_retries=1, retry_delay_seconds=0.0)
    monkeypatch.setenv("SERHIP_SLAB_BYTES", str(1 << 20))
    for pipelined in (False, True):
        m, errs = featurize_audio_jobs_device(jobs, feature_flags=flags, audio_read_config=cfg, workers=4, pipelined=pipelined)
        torch.cuda.synchronize()
        rows = m.cpu().numpy()
        assert not any(errs) and np.isfinite(rows).all()
        by_clip = {}
        for (k, kind), row in zip(kinds, rows):
            by_clip.setdefault(k, {})[kind] = row
        for k, group in by_clip.items():
            for kind in ("flac", "aiff", "au"):
                np.testing.assert_array_equal(group[kind], group["wav"], err_msg=f"clip {k} {kind} pipelined={pipelined}")
            assert not np.array_equal(group["alaw"], group["wav"])      # companded: close, not equal
            assert np.abs(group["alaw"][:40] - group["wav"][:40]).max() < 25.0


def test_interpreter_settings_are_restored_by_the_last_concurrent_call():
    """``interpreter_tuned_for_the_pipeline`` (GC held off, short thread switch interval) nests and overlaps: whatever
    order concurrent calls leave in, the process ends with the settings it started with."""
    import gc
    import sys
    import threading

    from ser_amd.ingest import interpreter_tuned_for_the_pipeline

    before = (gc.isenabled(), sys.getswitchinterval())
    assert before[0]
    inside, release_a, release_b = threading.Barrier(3), threading.Event(), threading.Event()

    def call(release):
        with interpreter_tuned_for_the_pipeline():
            assert not gc.isenabled() and sys.getswitchinterval() < 1e-3
            inside.wait()
            release.wait()

    a = threading.Thread(target=call, args=(release_a,))
    b = threading.Thread(target=call, args=(release_b,))
    a.start(); b.start()
    inside.wait()
    release_a.set(); a.join()
    assert not gc.isenabled() and sys.getswitchinterval() < 1e-3      # b is still inside
    release_b.set(); b.join()
    assert (gc.isenabled(), sys.getswitchinterval()) == before
    with interpreter_tuned_for_the_pipeline():
        with pytest.raises(RuntimeError):
            with interpreter_tuned_for_the_pipeline():
                raise RuntimeError("boom")
        assert not gc.isenabled()
    assert (gc.isenabled(), sys.getswitchinterval()) == before
    gc.disable()                                                       # a caller that runs with the collector off keeps it off
    try:
        with interpreter_tuned_for_the_pipeline():
            pass
        assert not gc.isenabled()
    finally:
        gc.enable()


def test_segments_of_one_recording_share_one_read(tmp_path, native_lib):
    """Consecutive jobs that name the same file and the same region (``serhip_read_wav_batch``): the file is read
    once and every job selects its own frames -- the selections equal the Python reader's, for PCM16 stereo and
    for a float32 payload that sits on a 2-byte boundary (the whole payload is slid to a sample boundary once)."""
    from ser_amd import ingest
    from ser_amd.audio_io import read_raw_clip

    rng = np.random.default_rng(41)
    a = tmp_path / "long16.wav"
    a.write_bytes(_wav_bytes(1, 16, 2, 16000, rng.integers(-30000, 30000, (16000 * 4, 2), dtype=np.int16).astype("<i2").tobytes()))
    b = tmp_path / "long32.wav"
    b.write_bytes(_wav_bytes(3, 32, 1, 22050, rng.standard_normal(22050 * 3).astype("<f4").tobytes(), junk=b"sixsix"))
    c = tmp_path / "other.wav"
    c.write_bytes(_wav_bytes(1, 16, 1, 16000, rng.integers(-30000, 30000, 8000, dtype=np.int16).astype("<i2").tobytes()))
    segs = [(0.0, 1.0), (0.5, 0.25), (1.0, None), (3.9, 5.0), (None, 2.0), (7.0, 1.0), (2.0, 1e-9)]   # the last two select nothing
    jobs = [(str(a), s, d) for s, d in segs] + [(str(c), None, None)] + [(str(b), s, d) for s, d in segs] + [(str(a), 0.25, 0.5)]
    paths = [j[0] for j in jobs]
    sizes = ingest.stat_files(paths, 4)
    follows = np.zeros(len(jobs), dtype=bool)
    follows[1:] = [x == y for x, y in zip(paths[1:], paths)]
    assert follows.sum() == 2 * (len(segs) - 1)
    # the planner never cuts a run, whatever the budget
    for budget in (1 << 20, 1 << 12, 1 << 30):
        slabs = ingest.plan_slabs(sizes, budget, follows)
        assert slabs[0][0] == 0 and slabs[-1][1] == len(jobs) and all(x[1] == y[0] for x, y in zip(slabs, slabs[1:]))
        assert not any(follows[lo] for lo, _ in slabs)
    own = (sizes + 15) // 16 * 16
    room = np.where(follows, 0, own)
    head = np.maximum.accumulate(np.where(follows, 0, np.arange(len(jobs))))
    starts = np.concatenate(([0], np.cumsum(room)[:-1]))
    buf = np.zeros(int(room.sum()) + 64, dtype=np.uint8)
    clips = ingest.read_wav_batch(paths, buf.ctypes.data, buf.size, starts[head], own[head],
                                  start_seconds=[j[1] for j in jobs], duration_seconds=[j[2] for j in jobs], threads=4)
    for (path, s, d), clip in zip(jobs, clips):
        want = read_raw_clip(path, start_seconds=s, duration_seconds=d)
        if want.frames == 0:
            assert clip["status"] == 6, (path, s, d)          # SERHIP_WAV_EMPTY
            continue
        assert clip["status"] == 0 and clip["frames"] == want.frames and clip["channels"] == want.channels
        width = (2 if want.sample_format == "s16" else 4) * want.channels
        assert clip["raw_offset"] % (width // want.channels) == 0
        got = buf[clip["raw_offset"] : clip["raw_offset"] + want.frames * width].tobytes()
        assert got == bytes(want.payload), (path, s, d)
    # regions of different files must not overlap
    with pytest.raises(Exception, match="overlap"):
        ingest.read_wav_batch([str(a), str(c)], buf.ctypes.data, buf.size, np.asarray([0, 16]), own[[0, 7]], threads=2)


def test_segment_reads_decode_a_file_once(tmp_path, monkeypatch):
    """The Python reader keeps the last few files it decoded for segment reads (``audio_io._decoded_files``): N
    utterances of one recording cost one decode, also when a team of threads asks at once; a changed file is decoded
    again; whole-file reads do not go through the cache."""
    import threading
    import time
    import wave

    from ser_amd import audio_io

    rng = np.random.default_rng(43)
    p = tmp_path / "rec24.wav"                               # 24-bit: the host decoder's business
    with wave.open(str(p), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(3)
        w.setframerate(8000)
        w.writeframes(rng.integers(0, 255, 3 * 8000 * 6, dtype=np.uint8).tobytes())
    calls = []
    real = audio_io.decode_bytes

    def counting(data):
        calls.append(len(data))
        time.sleep(0.05)
        return real(data)

    monkeypatch.setattr(audio_io, "decode_bytes", counting)
    audio_io._decoded_files.clear()
    whole, rate = audio_io.read_audio_file(str(p))
    assert len(calls) == 1 and rate == 8000
    out = [None] * 12

    def ask(k):
        out[k] = audio_io.read_audio_file(str(p), start_seconds=0.25 * k, duration_seconds=0.5)[0]

    threads = [threading.Thread(target=ask, args=(k,)) for k in range(12)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert len(calls) == 2                                   # one more decode for all twelve segments
    raw, _ = real(p.read_bytes())
    for k in range(12):
        seg = raw[int(0.25 * k * 8000) : int(0.25 * k * 8000) + 4000]
        np.testing.assert_array_equal(out[k], audio_io.prepare_audio_buffer(seg))
    audio_io.read_audio_file(str(p), start_seconds=1.0, duration_seconds=1.0)
    assert len(calls) == 2
    time.sleep(0.01)
    with wave.open(str(p), "wb") as w:                       # the file changes: size / mtime differ
        w.setnchannels(1)
        w.setsampwidth(3)
        w.setframerate(8000)
        w.writeframes(rng.integers(0, 255, 3 * 8000 * 5, dtype=np.uint8).tobytes())
    audio_io.read_audio_file(str(p), start_seconds=1.0, duration_seconds=1.0)
    assert len(calls) == 3
    audio_io._decoded_files.clear()


@pytest.mark.gpu
def test_manifest_segments_of_long_recordings(tmp_path, monkeypatch):
    """Utterances cut from long recordings by ``start_seconds`` / ``duration_seconds`` (the reference's manifests,
    data/manifest.py:175-230; read with ``librosa.load(offset=, duration=)``, audio_utils.py:104-109): the pipelined
    loader reads each recording once per slab and features every utterance from descriptors into it; rows equal the
    sequential loader's, for a PCM16 WAVE (device path) and a FLAC (host-decode lane with its decoded-file cache)."""
    import wave

    import torch

    from flac_writer import encode_stream
    from ser_amd import AudioReadConfig, FeatureFlags, audio_io
    from ser_amd.data_loader import featurize_audio_jobs_device

    rng = np.random.default_rng(47)
    n = 16000 * 40
    t = np.arange(n)[:, None]
    pcm = np.rint(6000 * np.sin(2 * np.pi * (180 + 3e-4 * t) * t / 16000 * np.asarray([1.0, 1.5])) + rng.integers(-400, 400, (n, 2))).astype(np.int64)
    with wave.open(str(tmp_path / "long.wav"), "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(pcm.astype("<i2").tobytes())
    m = 16000 * 12
    mono = pcm[:m, :1]
    frames = [{"block": 4096, "subframes": [{"type": "fixed", "order": 2, "partition_order": 4}]}] * (m // 4096)
    frames = frames + [{"block": m % 4096, "subframes": [{"type": "fixed", "order": 1}]}] if m % 4096 else frames
    (tmp_path / "long.flac").write_bytes(encode_stream(mono, bits=16, rate=16000, frames=frames))
    with wave.open(str(tmp_path / "short.wav"), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(mono[:20000].astype("<i2").tobytes())
    jobs = [(str(tmp_path / "short.wav"), None, None)]
    jobs += [(str(tmp_path / "long.wav"), 1.4 * k, 1.5) for k in range(27)]
    jobs += [(str(tmp_path / "long.wav"), 39.5, 3.0), (str(tmp_path / "long.wav"), 41.0, 1.0)]    # cut at the end; past the end
    jobs += [(str(tmp_path / "short.wav"), 0.25, None)]
    jobs += [(str(tmp_path / "long.flac"), 1.1 * k, 1.25) for k in range(9)]
    jobs += [(str(tmp_path / "long.wav"), 0.0, 2.0)]
    flags = FeatureFlags(tonnetz=False)
    cfg = AudioReadConfig(max_retries=1, retry_delay_seconds=0.0)
    audio_io._decoded_files.clear()
    seq, seq_err = featurize_audio_jobs_device(jobs, feature_flags=flags, audio_read_config=cfg, workers=4, pipelined=False)
    for slab_bytes in (1 << 20, 64 << 20):
        monkeypatch.setenv("SERHIP_SLAB_BYTES", str(slab_bytes))
        stats = {}
        pip, pip_err = featurize_audio_jobs_device(jobs, feature_flags=flags, audio_read_config=cfg, workers=4, pipelined=True, stats=stats)
        torch.cuda.synchronize()
        # the 2.56 MB recording is counted once per run of its jobs, not once per utterance
        assert stats["file_bytes"] < 3 * (tmp_path / "long.wav").stat().st_size
        a, b = seq.cpu().numpy(), pip.cpu().numpy()
        assert np.array_equal(np.isnan(a), np.isnan(b))
        np.testing.assert_array_equal(np.nan_to_num(a), np.nan_to_num(b))
        for e_seq, e_pip in zip(seq_err, pip_err):
            assert type(e_seq) is type(e_pip)
    failed = [i for i, e in enumerate(pip_err) if e is not None]
    assert failed == [29]                                   # the segment that starts past the end: "contains no samples"
    assert np.isfinite(b[[i for i in range(len(jobs)) if i != 29]]).all()
    # segment k of the WAVE equals the same samples handed over as a host clip (prepared like a file read)
    raw = (pcm[int(1.4 * 3 * 16000) : int(1.4 * 3 * 16000) + 24000].astype(np.float32) / np.float32(32768.0))
    clip = audio_io.prepare_audio_buffer(audio_io.mix_down_like_librosa_load(raw))
    from ser_amd.engine import get_engine

    want = get_engine().features(clip, [0], [len(clip)], 16000, flags).cpu().numpy()[0]
    np.testing.assert_array_equal(b[1 + 3], want)
    audio_io._decoded_files.clear()


def test_long_recording_is_read_in_pieces_by_several_threads(tmp_path, native_lib):
    """A file of 32 MiB or more in a batch with fewer files than threads is read with ``pread`` in 8 MiB pieces by
    the spare threads: same bytes as one sequential read, alone in the batch and beside a second file."""
    from ser_amd import ingest

    rng = np.random.default_rng(53)
    payload = rng.integers(-32768, 32767, 9 * (1 << 20) * 2 + 12345, dtype=np.int16).astype("<i2").tobytes()     # ~36 MiB
    big = tmp_path / "big.wav"
    big.write_bytes(_wav_bytes(1, 16, 1, 16000, payload))
    small = tmp_path / "small.wav"
    small.write_bytes(_wav_bytes(1, 16, 1, 16000, payload[:50000]))
    for paths in ([str(big)], [str(small), str(big)], [str(big), str(big)]):
        sizes = ingest.stat_files(paths, 4)
        own = (sizes + 15) // 16 * 16
        starts = np.concatenate(([0], np.cumsum(own)[:-1]))
        if paths[0] == paths[-1] and len(paths) == 2:      # two utterances of the big file: one shared region
            starts, own = np.asarray([0, 0]), np.asarray([own[0], own[0]])
        buf = np.zeros(int(own.sum()) + 64, dtype=np.uint8)
        clips = ingest.read_wav_batch(paths, buf.ctypes.data, buf.size, starts, own, threads=8,
                                      start_seconds=[None, 100.0][: len(paths)] if len(paths) == 2 else None,
                                      duration_seconds=[None, 2.0][: len(paths)] if len(paths) == 2 else None)
        assert not clips["status"].any()
        for path, clip in zip(paths, clips):
            want = payload if path == str(big) else payload[:50000]
            if clip["frames"] == 32000:                     # the (100 s, 2 s) utterance
                want = want[2 * 1600000 : 2 * 1632000]
            got = buf[clip["raw_offset"] : clip["raw_offset"] + 2 * clip["frames"]].tobytes()
            assert len(got) == len(want) and got == want, path


@pytest.mark.gpu
def test_long_file_inference_prepares_on_the_device(tmp_path, monkeypatch):
    """``extract_feature_frames`` on a recording of 4 MiB or more reads it with the reader team and prepares it on the
    GPU (``ingest.prepared_file_on_device``); frames, timestamps and features equal the host-prepared path's bit for
    bit (stereo PCM16 and misaligned float32), and a file the native reader does not take falls back unnoticed."""
    import wave

    from ser_amd import FeatureFlags, extract_feature_frames, ingest
    from ser_amd.feature_extractor import FeatureSettings

    rng = np.random.default_rng(59)
    n = 16000 * 80
    t = np.arange(n)[:, None]
    pcm = np.rint(5000 * np.sin(2 * np.pi * (140 + 2e-4 * t) * t / 16000 * np.asarray([1.0, 1.25])) + rng.integers(-300, 300, (n, 2))).astype("<i2")
    with wave.open(str(tmp_path / "talk.wav"), "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(pcm.tobytes())
    f32 = (pcm[:, 0].astype(np.float32) / 20000.0)
    f32[1000] = np.nan
    (tmp_path / "talk32.wav").write_bytes(_wav_bytes(3, 32, 1, 16000, f32.astype("<f4").tobytes(), junk=b"sixsix"))
    with wave.open(str(tmp_path / "talk24.wav"), "wb") as w:       # 24-bit: host decode, whatever its size
        w.setnchannels(1)
        w.setsampwidth(3)
        w.setframerate(16000)
        w.writeframes(rng.integers(0, 255, 3 * 16000 * 100, dtype=np.uint8).tobytes())
    settings = FeatureSettings(feature_flags=FeatureFlags(tonnetz=False))
    calls = []
    real = ingest.prepared_file_on_device
    monkeypatch.setattr(ingest, "prepared_file_on_device", lambda *a, **k: (calls.append(a[1]), real(*a, **k))[1])
    for name, expect_device in (("talk.wav", True), ("talk32.wav", True), ("talk24.wav", False)):
        calls.clear()
        monkeypatch.delenv("SERHIP_INGEST", raising=False)
        fast = extract_feature_frames(str(tmp_path / name), settings=settings)
        assert len(calls) == 1
        monkeypatch.setenv("SERHIP_INGEST", "sequential")
        slow = extract_feature_frames(str(tmp_path / name), settings=settings)
        assert len(calls) == 1                                     # the host path did not ask
        assert len(fast) == len(slow) >= 80
        for a, b in zip(fast, slow):
            assert (a.start_seconds, a.end_seconds) == (b.start_seconds, b.end_seconds)
            np.testing.assert_array_equal(a.features, b.features)
        assert (real(__import__("ser_amd.engine", fromlist=["get_engine"]).get_engine(), str(tmp_path / name)) is not None) == expect_device
        from ser_amd import extract_feature

        monkeypatch.delenv("SERHIP_INGEST", raising=False)
        whole_fast = extract_feature(str(tmp_path / name), settings=settings)
        monkeypatch.setenv("SERHIP_INGEST", "sequential")
        whole_slow = extract_feature(str(tmp_path / name), settings=settings)
        assert whole_fast.dtype == np.float64 and whole_fast.shape == (187,)
        np.testing.assert_array_equal(whole_fast, whole_slow)
