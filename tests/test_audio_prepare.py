"""Device-side audio preparation: raw WAVE payloads -> mono, scrubbed,
peak-normalised float32 on the GPU, bit-identical to the numpy code of
ser/_internal/utils/audio_utils.py:28-60 (restated in oracle/ser_fast_path.py)."""

import struct

import numpy as np
import pytest


def _wav_bytes(fmt_code: int, bits: int, channels: int, rate: int, payload: bytes) -> bytes:
    block = channels * bits // 8
    fmt = struct.pack("<HHIIHH", fmt_code, channels, rate, rate * block, block, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 4) + b"abcd"
    body += b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_read_raw_clip_headers_and_segments(tmp_path):
    from ser_amd.audio_io import AudioDecodeError, AudioIntegrityError, read_audio_file, read_raw_clip

    rng = np.random.default_rng(3)
    pcm = rng.integers(-20000, 20000, size=(1000, 2), dtype=np.int16)
    p = tmp_path / "a.wav"
    p.write_bytes(_wav_bytes(1, 16, 2, 8000, pcm.astype("<i2").tobytes()))
    clip = read_raw_clip(str(p))
    assert (clip.sample_format, clip.channels, clip.frames, clip.sample_rate) == ("s16", 2, 1000, 8000)
    np.testing.assert_array_equal(np.frombuffer(clip.payload, dtype="<i2").reshape(-1, 2), pcm)
    seg = read_raw_clip(str(p), start_seconds=0.05, duration_seconds=0.01)   # frames 400..480
    assert seg.frames == 80
    np.testing.assert_array_equal(np.frombuffer(seg.payload, dtype="<i2").reshape(-1, 2), pcm[400:480])
    host, rate = read_audio_file(str(p), start_seconds=0.05, duration_seconds=0.01)
    assert rate == 8000 and host.shape == (80,)
    # encodings the device path does not take -> None (callers fall back to the host decoder)
    q = tmp_path / "b.wav"
    q.write_bytes(_wav_bytes(1, 24, 1, 8000, bytes(300)))
    assert read_raw_clip(str(q)) is None
    f = tmp_path / "c.wav"
    x = rng.standard_normal(64).astype("<f4")
    f.write_bytes(_wav_bytes(3, 32, 1, 22050, x.tobytes()))
    fc = read_raw_clip(str(f))
    assert fc.sample_format == "f32" and fc.frames == 64 and fc.sample_rate == 22050
    with pytest.raises(FileNotFoundError):
        read_raw_clip(str(tmp_path / "missing.wav"))
    lfs = tmp_path / "l.wav"
    lfs.write_bytes(b"version https://git-lfs.github.com/spec/v1\noid sha256:0\n")
    with pytest.raises(AudioIntegrityError):
        read_raw_clip(str(lfs))
    bad = tmp_path / "x.wav"
    bad.write_bytes(b"not audio at all")
    with pytest.raises(AudioDecodeError):
        read_raw_clip(str(bad))
    with pytest.raises(ValueError):
        read_raw_clip(str(p), start_seconds=-1.0)


def test_host_prepare_orders_match_the_reference_functions():
    """Two orders exist in the reference and both are kept: ``prepare_audio_buffer`` IS ``_prepare_audio_buffer``
    (scrub, then mono: audio_utils.py:55-57), while a file read mixes down first like ``librosa.load(mono=True)``
    and scrubs afterwards (audio_utils.py:104-110).  1..9 channels (numpy's pairwise order from 8 on), non-finite
    samples in one channel, the reference's own known answer (tests/suites/unit/utils/test_audio_utils.py:14-27).
    No reference fixture holds a multichannel float file with NaN: the difference between the orders is
    parity unpinned; the known answers below are hand-derived from the two code paths."""
    from oracle import ser_fast_path as ofp
    from ser_amd.audio_io import mix_down_like_librosa_load, prepare_audio_buffer

    rng = np.random.default_rng(3)
    for ch in range(1, 10):
        x = rng.standard_normal((257, ch)).astype(np.float32)
        if ch > 1:
            x[5, 1] = np.nan
            x[9, 0] = np.inf
        arg = x if ch > 1 else x[:, 0]
        np.testing.assert_array_equal(prepare_audio_buffer(arg), ofp.prepare_audio_buffer(arg))
        np.testing.assert_array_equal(prepare_audio_buffer(mix_down_like_librosa_load(arg)), ofp.read_audio_prepared(arg))
    two = np.array([[0.5, np.nan], [0.25, 0.75], [np.inf, -np.inf]], dtype=np.float32)
    # file order: frames 0 and 2 are non-finite after the mean -> 0; frame 1 = 0.5 -> peak
    np.testing.assert_array_equal(prepare_audio_buffer(mix_down_like_librosa_load(two)),
                                  np.array([0.0, 1.0, 0.0], dtype=np.float32))
    # the reference function's order: scrub first -> [[0.5, 0], [0.25, 0.75], [0, 0]] -> means [0.25, 0.5, 0] -> / 0.5
    np.testing.assert_array_equal(prepare_audio_buffer(two), np.array([0.5, 1.0, 0.0], dtype=np.float32))
    np.testing.assert_array_equal(prepare_audio_buffer(np.array([[0.2, 0.0], [0.4, 0.0]], np.float32)),
                                  np.array([0.5, 1.0], np.float32))


def test_file_read_uses_the_librosa_load_order(tmp_path):
    """A stereo float32 WAVE with a NaN in one channel, read through ``read_audio_file``: the frame is zeroed
    (mix-down first), not averaged over the finite channel."""
    import struct

    from ser_amd.audio_io import read_audio_file

    frames = np.array([[0.5, np.nan], [0.25, 0.75]], dtype="<f4")
    body = frames.tobytes()
    fmt = struct.pack("<HHIIHH", 3, 2, 8000, 8000 * 8, 8, 32)
    data = b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(body)) + b"WAVE" + b"fmt " + \
        struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(body)) + body
    p = tmp_path / "nan.wav"
    p.write_bytes(data)
    audio, rate = read_audio_file(str(p))
    assert rate == 8000
    np.testing.assert_array_equal(audio, np.array([0.0, 1.0], dtype=np.float32))


@pytest.mark.gpu
def test_prepare_audio_bit_exact_vs_numpy():
    """PCM16 and float32, 1..9 channels, NaN/inf, an all-zero clip: identical bits to the oracle."""
    from oracle import ser_fast_path as ofp
    from ser_amd.engine import get_engine

    eng = get_engine()
    rng = np.random.default_rng(11)
    # --- PCM16
    clips = [rng.integers(-32768, 32767, size=(n, c), dtype=np.int16) for n, c in
             ((48000, 1), (1, 1), (1777, 2), (999, 3), (4096, 6), (513, 8), (700, 9))]
    clips.append(np.zeros((300, 2), dtype=np.int16))
    offsets, cursor, chunks = [], 0, []
    for c in clips:
        offsets.append(cursor)
        b = c.astype("<i2").tobytes()
        pad = (-len(b)) % 16
        chunks.append(b + bytes(pad))
        cursor += len(b) + pad
    audio, out_off, frames = eng.prepare_audio(b"".join(chunks), "s16", offsets, [c.shape[0] for c in clips],
                                               [c.shape[1] for c in clips])
    got = audio.cpu().numpy()
    for c, o, n in zip(clips, out_off, frames):
        raw = c.astype(np.float32) / np.float32(32768.0)
        want = ofp.read_audio_prepared(raw if c.shape[1] > 1 else raw[:, 0])
        np.testing.assert_array_equal(got[o : o + n], want)
    # --- float32 with non-finite samples: the reference mixes down first (librosa.load) and scrubs the mono
    # signal, so a NaN / inf in ONE channel zeroes the frame (f[1] frame 5, f[3] frames 7 and 9)
    f = [rng.standard_normal((n, c)).astype(np.float32) * np.float32(3.0) for n, c in ((5000, 1), (2049, 2), (100, 5))]
    f[0][[3, 77, 4000], 0] = [np.nan, np.inf, -np.inf]
    f[1][5, 1] = np.nan
    f[2][50, :] = np.inf
    f.append(rng.standard_normal((64, 2)).astype(np.float32))
    f[3][7] = [np.inf, 0.25]
    f[3][9] = [np.inf, -np.inf]
    raw = np.concatenate([x.reshape(-1) for x in f])
    offs = np.cumsum([0] + [x.size * 4 for x in f[:-1]])
    audio, out_off, frames = eng.prepare_audio(raw, "f32", offs, [x.shape[0] for x in f], [x.shape[1] for x in f])
    got = audio.cpu().numpy()
    for x, o, n in zip(f, out_off, frames):
        want = ofp.read_audio_prepared(x if x.shape[1] > 1 else x[:, 0])
        np.testing.assert_array_equal(got[o : o + n], want)
        assert np.abs(got[o : o + n]).max() == 1.0
    assert got[out_off[1] + 5] == 0.0 and got[out_off[3] + 7] == 0.0 and got[out_off[3] + 9] == 0.0
    # --- both kernels' routes: clips of <= 2^18 frames take the fused one-workgroup-per-clip kernel (vector loads when
    # the clip starts on a 4-byte boundary, scalar ones at offsets = 2 mod 4); a batch holding a longer clip takes the
    # convert + normalise pair for every clip of the call
    for long_frames in (0, 300_000):
        pcm = [rng.integers(-32768, 32767, size=(n, c), dtype=np.int16) for n, c in ((4001, 1), (3003, 2), (1502, 1), (77, 2))]
        if long_frames:
            pcm.append(rng.integers(-32768, 32767, size=(long_frames, 1), dtype=np.int16))
        blob, offs = b"", []
        for k, c in enumerate(pcm):
            blob += bytes((-len(blob)) % 4) + (b"\0\0" if k % 2 else b"")    # every other clip at an offset = 2 mod 4
            offs.append(len(blob))
            blob += c.astype("<i2").tobytes()
        audio, out_off, frames = eng.prepare_audio(blob, "s16", offs, [c.shape[0] for c in pcm], [c.shape[1] for c in pcm])
        got = audio.cpu().numpy()
        for c, o, n in zip(pcm, out_off, frames):
            raw = c.astype(np.float32) / np.float32(32768.0)
            np.testing.assert_array_equal(got[o : o + n], ofp.read_audio_prepared(raw if c.shape[1] > 1 else raw[:, 0]))
    # --- argument errors come back as SerhipError, nothing is launched
    from ser_amd._native import SerhipError

    with pytest.raises(SerhipError):
        eng.prepare_audio(bytes(64), "s16", [0], [100], [1])       # clip larger than the buffer
    with pytest.raises(SerhipError):
        eng.prepare_audio(bytes(64), "s16", [1], [4], [1])         # odd byte offset
    with pytest.raises(ValueError):
        eng.prepare_audio(bytes(64), "u8", [0], [4], [1])


@pytest.mark.gpu
def test_loader_device_prepare_equals_host_prepare(tmp_path):
    """featurize_audio_jobs: the raw-bytes path (GPU conversion + normalisation) and the host-decoded
    path give the same matrix; mixed encodings and a broken file in one call."""
    import wave

    from ser_amd import FeatureFlags
    from ser_amd.data_loader import featurize_audio_jobs
    from ser_amd.synth import ravdess_sine_clip
    from ser_amd.types import AudioReadConfig

    rng = np.random.default_rng(5)
    paths = []
    for k in range(6):
        p = tmp_path / f"clip{k}.wav"
        pcm = ravdess_sine_clip(1 + k % 2, 1 + k)
        if k == 3:   # stereo
            pcm = np.stack([pcm, pcm[::-1]], axis=1)
        with wave.open(str(p), "wb") as w:
            w.setnchannels(1 if pcm.ndim == 1 else 2)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes(np.asarray(pcm, dtype="<i2").tobytes())
        paths.append(str(p))
    f32 = tmp_path / "float.wav"
    x = (0.3 * rng.standard_normal(20000)).astype("<f4")
    f32.write_bytes(_wav_bytes(3, 32, 1, 16000, x.tobytes()))
    paths.append(str(f32))
    p24 = tmp_path / "pcm24.wav"   # host-decoded encoding
    v = (rng.integers(-(1 << 22), 1 << 22, size=9000)).astype(np.int32)
    b = np.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], axis=1).astype(np.uint8).tobytes()
    p24.write_bytes(_wav_bytes(1, 24, 1, 16000, b))
    paths.append(str(p24))
    broken = tmp_path / "broken.wav"
    broken.write_bytes(b"RIFFxxxxWAVEjunk")
    paths.append(str(broken))
    flags = FeatureFlags(tonnetz=False)
    cfg = AudioReadConfig(max_retries=1, retry_delay_seconds=0.0)
    jobs = [(p, None, None) for p in paths] + [(paths[0], 0.25, 0.5)]
    dev, dev_err = featurize_audio_jobs(jobs, feature_flags=flags, audio_read_config=cfg, workers=2)
    host, host_err = featurize_audio_jobs(jobs, feature_flags=flags, audio_read_config=cfg, workers=2,
                                          device_prepare=False)
    assert [e is None for e in dev_err] == [e is None for e in host_err] == [True] * 8 + [False, True]
    ok = [i for i, e in enumerate(dev_err) if e is None]
    np.testing.assert_array_equal(dev[ok], host[ok])   # same samples bit for bit -> same features
    assert np.isnan(dev[8]).all()
