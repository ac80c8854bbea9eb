"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Tolerances are BASELINE.md's parity gates (tests/parity_utils.py): log-mel/MFCC rtol 1e-4
+ atol 2e-3; mel power and chroma rtol 1e-4 + atol 1e-6*max; contrast exactly 0; tuning
estimate exactly equal on these seeded inputs.
"""

import numpy as np
import pytest

from parity_utils import CHROMA, MEL, MFCC, assert_features_close

pytestmark = pytest.mark.gpu

SR = 16000


@pytest.fixture(scope="module")
def engine():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ser_amd.engine import get_engine

    return get_engine()


@pytest.fixture(scope="module")
def flags():
    from ser_amd import FeatureFlags

    return FeatureFlags(tonnetz=False)


def oracle_rows(audio, offs, lens, sr=SR, oflags=None):
    from oracle import ser_fast_path as ofp

    rows, tunings = [], []
    for o, n in zip(offs, lens):
        v, dbg = ofp.extract_feature_from_signal(audio[o : o + n], sr, flags=oflags, return_debug=True)
        rows.append(v)
        tunings.append(dbg.get("tuning", 0.0))
    return np.asarray(rows), np.asarray(tunings)


def gpu_rows(engine, flags, audio, offs, lens, sr=SR):
    out, tun = engine.features(audio, offs, lens, sr, flags, return_tuning=True)
    return out.cpu().numpy().astype(np.float64), tun.cpu().numpy().astype(np.float64)


def test_loaded_native_library(engine):
    from ser_amd import _native

    assert _native.LIB_PATH.exists()
    assert _native.load_library().serhip_version() >= 100


def test_driver_entry_points_in_a_fresh_process():
    """``build()`` then ``smoke()`` exactly as the driver calls them: a new interpreter in which the
    shared library is touched before anything has imported torch (two HIP runtimes in one process
    was a real failure: the second to initialise sees no device)."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    code = "import __graft_entry__ as g; g.build(); g.smoke()"
    res = subprocess.run([sys.executable, "-c", code], cwd=str(root), capture_output=True, text=True, timeout=170)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "smoke ok" in res.stdout


def test_plain_c_host_over_the_c_abi(engine, flags, tmp_path):
    """``examples/host_c_abi`` -- gcc, the system HIP runtime, no Python, no torch -- reads WAVE files with the
    library's reader team, prepares and featurises them through the C ABI and prints the rows: the same bits the
    Python host gets for the same files, and a file the device path does not take is reported, not guessed."""
    import subprocess
    import wave
    from pathlib import Path

    from ser_amd import AudioReadConfig
    from ser_amd.data_loader import featurize_audio_jobs_device

    exe = Path(__file__).resolve().parent.parent / "examples" / "host_c_abi"
    if not exe.exists():      # a tree that was copied without its built artefacts: build() compiles it (gcc, a second)
        import sys

        sys.path.insert(0, str(exe.parent.parent))
        import __graft_entry__

        __graft_entry__.build()
    assert exe.exists(), "examples/host_c_abi is built by __graft_entry__.build()"
    rng = np.random.default_rng(31)
    paths = []
    for k, (channels, n) in enumerate(((1, 48000), (2, 30001), (1, 1500), (2, 2048))):
        t = np.arange(n)[:, None]
        pcm = np.rint(7000 * np.sin(2 * np.pi * (150 + 40 * k + 25 * np.arange(channels)) * t / 16000)
                      + rng.integers(-200, 200, (n, channels))).astype("<i2")
        with wave.open(str(tmp_path / f"h{k}.wav"), "wb") as w:
            w.setnchannels(channels)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes(pcm.tobytes())
        paths.append(str(tmp_path / f"h{k}.wav"))
    with wave.open(str(tmp_path / "wide.wav"), "wb") as w:      # 24-bit PCM: the host decoder's business
        w.setnchannels(1)
        w.setsampwidth(3)
        w.setframerate(16000)
        w.writeframes(bytes(3 * 4000))
    args = paths[:2] + [str(tmp_path / "wide.wav"), str(tmp_path / "missing.wav")] + paths[2:]
    res = subprocess.run([str(exe), *args], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = res.stdout.strip().split("\n")
    assert len(lines) == len(args)
    assert lines[2].endswith("SKIPPED status=1") and lines[3].endswith("SKIPPED status=2")     # HOST_DECODE, IO
    rows = {}
    for line in lines:
        parts = line.split(" ")
        if parts[1] != "SKIPPED":
            assert int(parts[2]) == 16000 and int(parts[3]) == 187
            rows[parts[0]] = np.asarray([np.float32(v) for v in parts[4:]], dtype=np.float32)
    want, errs = featurize_audio_jobs_device([(p, None, None) for p in paths], feature_flags=flags,
                                             audio_read_config=AudioReadConfig(), workers=2)
    assert not any(errs)
    want = want.cpu().numpy()
    for i, p in enumerate(paths):
        np.testing.assert_array_equal(rows[p], want[i], err_msg=p)


def test_config3_clips_batch(engine, flags):
    """Config 3 shape (3 s clips @16 kHz, harmonics + noise), 24 clips."""
    from ser_amd.synth import harmonic_clips

    clips = harmonic_clips(1234, 24, 48000)
    audio = clips.reshape(-1)
    offs = [i * 48000 for i in range(24)]
    lens = [48000] * 24
    got, tun = gpu_rows(engine, flags, audio, offs, lens)
    want, wtun = oracle_rows(audio, offs, lens)
    np.testing.assert_allclose(tun, wtun, atol=1e-6)
    assert_features_close(got, want)


def test_config1_sample_wav_windows(engine, flags):
    """Config 1: 69 937-sample recording -> 3 s / 1 s windows through HandcraftedBackend."""
    from oracle import ser_fast_path as ofp
    from ser_amd import HandcraftedBackend
    from ser_amd.synth import sample_wav_like

    audio = sample_wav_like()
    enc = HandcraftedBackend(feature_flags=flags).encode_sequence(audio, SR)
    ref, starts, ends = ofp.encode_sequence(audio, SR)
    assert enc.embeddings.dtype == np.float32 and enc.embeddings.shape == (5, 187)
    np.testing.assert_array_equal(enc.frame_start_seconds, starts)
    np.testing.assert_array_equal(enc.frame_end_seconds, ends)
    assert enc.backend_id == "handcrafted"
    assert_features_close(enc.embeddings, ref)


def test_config2_synthetic_ravdess(engine, flags):
    """Config 2: the 16 PCM16 sine clips of the synthetic RAVDESS generator, whole-clip vectors."""
    from oracle import ser_fast_path as ofp
    from ser_amd import HandcraftedBackend
    from ser_amd.audio_io import prepare_audio_buffer
    from ser_amd.synth import ravdess_sine_clip

    clips = [prepare_audio_buffer(ravdess_sine_clip(a, e)) for a in (1, 2) for e in range(1, 9)]
    ref_clip = ofp.prepare_audio_buffer(ofp.synthetic_ravdess_clip(2, 5))
    np.testing.assert_array_equal(clips[8 + 4], ref_clip)
    got = HandcraftedBackend(feature_flags=flags).extract_vectors(clips, SR)
    want = np.asarray([ofp.extract_vector(c, SR) for c in clips])
    assert got.dtype == np.float64 and got.shape == (16, 187)
    assert_features_close(got, want)


def test_ragged_unaligned_and_short_segments(engine, flags):
    """Ragged lengths, odd offsets, and every n_fft regime: N<512 (padded), 512<=N<2048
    (n_fft = N, odd and even, direct DFT kernel), N = 2048, N just above."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal(40000).astype(np.float32)
    x /= np.abs(x).max()
    offs = [0, 1, 4001, 100, 7, 0, 10, 33, 500, 77, 3, 9]
    lens = [2048, 2049, 5937, 20000, 29993, 100, 512, 1023, 1500, 2047, 1, 513]
    got, tun = gpu_rows(engine, flags, x, offs, lens)
    want, wtun = oracle_rows(x, offs, lens)
    np.testing.assert_allclose(tun, wtun, atol=1e-6)
    assert_features_close(got, want)


def test_long_single_segments(engine, flags):
    """Segments of more than 1024 frames take the tuning kernel's other gather paths: (a) many peaks
    (streamed from the scratch, never staged in LDS), (b) few peaks in a long, mostly silent
    recording (staged in LDS by the sequential gather)."""
    from ser_amd.synth import harmonic_clip

    n = 40 * SR                                      # 1251 frames
    a = harmonic_clip(np.random.default_rng(40), n, SR)
    b = np.zeros(n, dtype=np.float32)
    b[5 * SR : 6 * SR] = harmonic_clip(np.random.default_rng(41), SR, SR)
    audio = np.concatenate([a, b])
    got, tun = gpu_rows(engine, flags, audio, [0, n], [n, n])
    want, wtun = oracle_rows(audio, [0, n], [n, n])
    np.testing.assert_allclose(tun, wtun, atol=1e-6)
    assert_features_close(got, want)


def test_dense_peaks_tuning_paths(engine, flags):
    """3 s clips with thousands of piptrack peaks: too many for the tuning kernel's LDS staging, so the
    median runs over the rows in HBM -- with the selected radix bucket compacted into LDS (harmonics in
    noise) or, when even that bucket is too large, digit by digit from HBM (white noise)."""
    from ser_amd.synth import harmonic_clip

    rng = np.random.default_rng(77)
    n = 3 * SR
    clips = []
    for noise_level in (0.3, 1.0, 0.0):
        x = noise_level * rng.standard_normal(n).astype(np.float32)
        if noise_level < 1.0:
            x = x + harmonic_clip(rng, n, SR)
        clips.append((x / np.abs(x).max()).astype(np.float32))
    clips.append(clips[1][:2 * SR + 1])              # white noise, odd length (odd peak count likely differs)
    audio = np.concatenate(clips)
    offs = np.cumsum([0] + [len(c) for c in clips[:-1]]).tolist()
    lens = [len(c) for c in clips]
    got, tun = gpu_rows(engine, flags, audio, offs, lens)
    want, wtun = oracle_rows(audio, offs, lens)
    np.testing.assert_allclose(tun, wtun, atol=1e-6)
    assert_features_close(got, want)


def test_top_db_clipping_paths(engine, flags):
    """power_to_db(top_db=80) clips against the segment's maximum.  The projection kernel stores a block
    of dB rows only if it might be clipped, and the reduction re-reads only tiles whose minimum is below
    the floor: loud-then-faint clips (the switch inside a 16-frame tile, faint part 70 to 130 dB down),
    a faint clip alone (nothing clipped, nothing stored) and digital silence after sound."""
    from ser_amd.synth import harmonic_clip

    rng = np.random.default_rng(91)
    n = 3 * SR
    loud = harmonic_clip(rng, n, SR)
    clips = []
    for gain in (3e-4, 1e-5, 3e-7):
        x = loud.copy()
        cut = 20000 + int(rng.integers(0, 5000))        # not on a tile boundary
        x[cut:] = gain * rng.standard_normal(n - cut).astype(np.float32)
        clips.append(x)
    clips.append((1e-4 * rng.standard_normal(n)).astype(np.float32))
    x = loud.copy()
    x[30000:] = 0.0
    clips.append(x)
    audio = np.concatenate(clips)
    offs = [i * n for i in range(len(clips))]
    lens = [n] * len(clips)
    got, tun = gpu_rows(engine, flags, audio, offs, lens)
    want, wtun = oracle_rows(audio, offs, lens)
    np.testing.assert_allclose(tun, wtun, atol=1e-6)
    assert_features_close(got, want)
    # the same clips one by one and in the reverse order: stale dB rows of other calls must never be read
    back, _ = gpu_rows(engine, flags, audio, offs[::-1], lens)
    np.testing.assert_array_equal(back[::-1], got)
    # without chroma the segment's magnitude bound takes another route (no tuning kernel to reduce it: per-wave
    # running maxima and atomics in the FFT kernel); the clipped MFCC / mel blocks must come out the same
    from ser_amd import FeatureFlags

    no_chroma = FeatureFlags(mfcc=True, chroma=False, mel=True, contrast=False, tonnetz=False)
    part = engine.features(audio, offs, lens, SR, no_chroma).cpu().numpy()
    np.testing.assert_array_equal(part, np.concatenate([got[:, MFCC], got[:, MEL]], axis=1))


def test_more_segments_than_one_launch_holds(engine, flags):
    """70 000 overlapping windows in one call: more segments than a grid's y dimension (65 535), so the
    library cuts chunks; fast path (2100 samples) and generic path (700 samples).  Rows must equal
    the same windows featurised on their own (bit-exact: results do not depend on the batch)."""
    rng = np.random.default_rng(70)
    x = rng.standard_normal(300000).astype(np.float32)
    x /= np.abs(x).max()
    n = 70000
    offs = (np.arange(n, dtype=np.int64) * 4) % (x.size - 2100)
    for length in (2100, 700):
        lens = np.full(n, length, dtype=np.int64)
        got = engine.features(x, offs, lens, SR, flags).cpu().numpy()
        assert got.shape == (n, 187) and np.isfinite(got).all()
        pick = np.asarray([0, 1, 65534, 65535, 65536, 69999, 31337])
        alone = engine.features(x, offs[pick], lens[pick], SR, flags).cpu().numpy()
        np.testing.assert_array_equal(got[pick], alone)
        want, _ = oracle_rows(x, offs[pick[:3]], lens[pick[:3]])
        assert_features_close(got[pick[:3]], want)


def test_uneven_units_dealt_by_frame_count(engine, flags):
    """3 000 segments of four interleaved lengths (period 4 divides every grid size): several work units
    per persistent workgroup, units of unequal frame counts, so the host deals them least-loaded-first
    instead of round robin.  Rows must equal the same segments featurised on their own (bit-exact) and
    the oracle's."""
    rng = np.random.default_rng(71)
    x = rng.standard_normal(400000).astype(np.float32)
    x /= np.abs(x).max()
    n = 3000
    lens = np.asarray([48000, 2048, 24000, 70001], dtype=np.int64)[np.arange(n) % 4]
    offs = (np.arange(n, dtype=np.int64) * 37) % (x.size - 70001)
    got = engine.features(x, offs, lens, SR, flags).cpu().numpy()
    assert got.shape == (n, 187) and np.isfinite(got).all()
    pick = np.asarray([0, 1, 2, 3, 1023, 1024, 1025, 1026, 2996, 2997, 2998, 2999])
    alone = engine.features(x, offs[pick], lens[pick], SR, flags).cpu().numpy()
    np.testing.assert_array_equal(got[pick], alone)
    want, _ = oracle_rows(x, offs[pick[:4]], lens[pick[:4]])
    assert_features_close(got[pick[:4]], want)


def test_silence_and_constant(engine, flags):
    z = np.zeros(9000, np.float32)
    got, tun = gpu_rows(engine, flags, z, [0, 100], [9000, 1000])
    want, _ = oracle_rows(z, [0, 100], [9000, 1000])
    assert tun.tolist() == [0.0, 0.0]
    assert_features_close(got, want)
    assert got[0, 0] == pytest.approx(-1131.3708, rel=1e-6)   # -100 dB * sqrt(128)
    assert not got[:, CHROMA].any() and not got[:, MEL].any()


def test_other_sample_rates(engine, flags):
    from ser_amd.synth import harmonic_clip

    for sr in (22050, 44100, 48000):
        a = harmonic_clip(np.random.default_rng(sr), 3 * sr // 2, sr)
        got, tun = gpu_rows(engine, flags, a, [0], [a.size], sr=sr)
        want, wtun = oracle_rows(a, [0], [a.size], sr=sr)
        np.testing.assert_allclose(tun, wtun, atol=1e-6)
        assert_features_close(got, want)


def test_low_and_high_sample_rates(engine, flags):
    """11 025 / 12 000 Hz: the piptrack range (150-4000 Hz) extends past bin 511, the stripes the FFT
    kernel scans from LDS instead of registers; contrast is off (the reference raises for sr <= 12 800).
    8 000 Hz: the range ends one bin below the Nyquist row.
    96 000 Hz: the low mel filters are narrower than an FFT bin and several 16-filter blocks share groups."""
    from ser_amd import FeatureFlags
    from ser_amd.synth import harmonic_clip
    from oracle import ser_fast_path as ofp

    for sr in (11025, 12000):
        fl = FeatureFlags(mfcc=True, chroma=True, mel=True, contrast=False, tonnetz=False)
        a = harmonic_clip(np.random.default_rng(sr), 2 * sr, sr)
        got, tun = gpu_rows(engine, fl, a, [0, 100], [a.size, 5000], sr=sr)
        want, wtun = oracle_rows(a, [0, 100], [a.size, 5000], sr=sr, oflags=ofp.Flags(contrast=False))
        np.testing.assert_allclose(tun, wtun, atol=1e-6)
        pad = np.zeros((2, 7))   # these plans have no contrast block: append the zeros the checker expects
        assert_features_close(np.hstack([got, pad]), np.hstack([want, pad]))
    # 8 000 Hz: the piptrack range ends at bin 1023, one below the Nyquist row
    fl = FeatureFlags(mfcc=True, chroma=True, mel=True, contrast=False, tonnetz=False)
    a = harmonic_clip(np.random.default_rng(8), 16000, 8000)
    got, tun = gpu_rows(engine, fl, a, [0], [a.size], sr=8000)
    want, wtun = oracle_rows(a, [0], [a.size], sr=8000, oflags=ofp.Flags(contrast=False))
    np.testing.assert_allclose(tun, wtun, atol=1e-6)
    pad = np.zeros((1, 7))
    assert_features_close(np.hstack([got, pad]), np.hstack([want, pad]))
    a = harmonic_clip(np.random.default_rng(96), 96000, 96000)
    got, tun = gpu_rows(engine, flags, a, [0], [a.size], sr=96000)
    want, wtun = oracle_rows(a, [0], [a.size], sr=96000)
    np.testing.assert_allclose(tun, wtun, atol=1e-6)
    assert_features_close(got, want)


def test_flag_subsets(engine, flags):
    """Every block is computed identically whatever the other flags are; order mfcc|chroma|mel|contrast."""
    from ser_amd import FeatureFlags
    from ser_amd.synth import harmonic_clips

    audio = harmonic_clips(77, 2, 30000).reshape(-1)
    offs, lens = [0, 30000], [30000, 30000]
    full = engine.features(audio, offs, lens, SR, flags).cpu().numpy()
    blocks = {"mfcc": MFCC, "chroma": CHROMA, "mel": MEL, "contrast": slice(180, 187)}
    for names in (("mfcc",), ("mel",), ("chroma",), ("mfcc", "mel"), ("chroma", "contrast"), ("contrast",)):
        fl = FeatureFlags(**{k: (k in names) for k in blocks}, tonnetz=False)
        out = engine.features(audio, offs, lens, SR, fl).cpu().numpy()
        want = np.concatenate([full[:, blocks[k]] for k in blocks if k in names], axis=1)
        np.testing.assert_array_equal(out, want)
    empty = engine.features(audio, [0], [30000], SR, FeatureFlags(False, False, False, False, False))
    assert tuple(empty.shape) == (1, 0)
    with pytest.raises(NotImplementedError):
        engine.features(audio, [0], [30000], SR, FeatureFlags())   # tonnetz on


def test_batch_independence_and_exact_scaling(engine, flags):
    """Size-independent properties: a segment's vector does not depend on its batch, and
    scaling the audio by 2 scales mel power by exactly 4, shifts MFCC[0] by 10*log10(4)*sqrt(128)
    and leaves chroma untouched (power-of-two scaling is exact in binary floating point)."""
    from ser_amd.synth import harmonic_clips

    clips = harmonic_clips(99, 6, 48000) * np.float32(0.25)
    audio = clips.reshape(-1)
    offs = [i * 48000 for i in range(6)]
    lens = [48000] * 6
    batch, _ = gpu_rows(engine, flags, audio, offs, lens)
    for i in (0, 3, 5):
        alone, _ = gpu_rows(engine, flags, clips[i], [0], [48000])
        np.testing.assert_array_equal(alone[0], batch[i])
    doubled, _ = gpu_rows(engine, flags, audio * np.float32(2.0), offs, lens)
    np.testing.assert_array_equal(doubled[:, MEL], 4.0 * batch[:, MEL])
    np.testing.assert_array_equal(doubled[:, CHROMA], batch[:, CHROMA])
    np.testing.assert_allclose(doubled[:, 0] - batch[:, 0], 10.0 * np.log10(4.0) * np.sqrt(128.0), rtol=1e-5)
    np.testing.assert_allclose(doubled[:, 1:40], batch[:, 1:40], atol=2e-3)


def test_native_error_reporting(engine, flags):
    from ser_amd._native import SerhipError

    x = np.zeros(5000, np.float32)
    with pytest.raises(SerhipError, match="outside the audio buffer"):
        engine.features(x, [4000], [2000], SR, flags)
    with pytest.raises(SerhipError, match="empty"):
        engine.features(x, [0], [0], SR, flags)


def test_extract_feature_from_signal_contract(engine):
    """Reference call signature, validation order and messages (dsp.py:85-95)."""
    from ser_amd import FeatureFlags, extract_feature_from_signal
    from ser_amd.dsp import ParameterError

    x = np.linspace(-1, 1, 6000, dtype=np.float32)
    v = extract_feature_from_signal(x, SR, feature_flags=FeatureFlags(tonnetz=False))
    assert v.dtype == np.float64 and v.shape == (187,)
    with pytest.raises(NotImplementedError, match="tonnetz"):   # None = the reference's 193-dim default: loud
        extract_feature_from_signal(x, SR)
    with pytest.raises(ValueError, match="Sample rate must be a positive integer."):
        extract_feature_from_signal(x, 0)
    with pytest.raises(ValueError, match=r"Audio must be mono \(1D array\)."):
        extract_feature_from_signal(np.zeros((4, 2), np.float32), SR)
    with pytest.raises(ValueError, match="no samples"):
        extract_feature_from_signal(np.zeros(0, np.float32), SR)
    bad = x.copy()
    bad[17] = np.nan
    with pytest.raises(ValueError, match="not finite everywhere"):
        extract_feature_from_signal(bad, SR)
    with pytest.raises(ParameterError, match="exceeds Nyquist"):
        extract_feature_from_signal(x, 8000, feature_flags=FeatureFlags(tonnetz=False))
    assert extract_feature_from_signal(x, SR, feature_flags=FeatureFlags(False, False, False, False, False)).shape == (0,)


def test_repeated_plan_skips_descriptor_upload_safely(engine, flags):
    """A call with the same segment plan as the one before reuses the descriptors already in the
    workspace; any other use of the workspace in between (another plan, a pooling call) must force the
    upload again.  Every variant has to give the same bits."""
    from ser_amd import EncodedSequence, PoolingWindow, mean_std_pool
    from ser_amd.synth import harmonic_clips

    a1 = harmonic_clips(501, 8, 48000).reshape(-1)
    a2 = harmonic_clips(502, 8, 48000).reshape(-1)
    offs, lens = [i * 48000 for i in range(8)], [48000] * 8
    other_offs, other_lens = [0, 1000], [30000, 2500]
    first, _ = gpu_rows(engine, flags, a1, offs, lens)
    hit, _ = gpu_rows(engine, flags, a2, offs, lens)                 # same plan: no upload
    gpu_rows(engine, flags, a1, other_offs, other_lens)              # different plan
    miss, _ = gpu_rows(engine, flags, a2, offs, lens)                # uploaded again
    np.testing.assert_array_equal(hit, miss)
    enc = EncodedSequence(miss.astype(np.float32), np.arange(8.0), np.arange(8.0) + 1.0, "handcrafted")
    mean_std_pool(enc, [PoolingWindow(0.0, 4.0), PoolingWindow(2.0, 8.0)])   # overwrites the workspace head
    after_pool, _ = gpu_rows(engine, flags, a2, offs, lens)
    np.testing.assert_array_equal(after_pool, miss)
    again, _ = gpu_rows(engine, flags, a1, offs, lens)               # same plan, other audio: hit
    np.testing.assert_array_equal(again, first)


def test_calls_on_alternating_streams_share_the_workspace_safely(engine, flags):
    """Back-to-back calls on two different HIP streams, no host synchronisation in between: every call
    uses the context's one workspace, so the library orders them on the device (include/serhip.h,
    conventions).  Each result must equal the same call made alone."""
    import torch

    from ser_amd.synth import harmonic_clips

    n = 600                                    # ~0.2 ms of kernels per call: calls do overlap if unordered
    audios = [torch.from_numpy(harmonic_clips(600 + k, n, 48000).reshape(-1)).to(engine.device) for k in range(2)]
    offs, lens = [i * 48000 for i in range(n)], [48000] * n
    short_lens = [30000] * n                   # another plan: other scratch offsets in the same workspace
    alone = [engine.features(audios[0], offs, lens, SR, flags).cpu().numpy(),
             engine.features(audios[1], offs, short_lens, SR, flags).cpu().numpy()]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=engine.device) for _ in range(2)]
    outs = []
    for rep in range(6):
        k = rep % 2
        with torch.cuda.stream(streams[k]):
            outs.append(engine.features(audios[k], offs, lens if k == 0 else short_lens, SR, flags))
    torch.cuda.synchronize()
    for rep, out in enumerate(outs):
        np.testing.assert_array_equal(out.cpu().numpy(), alone[rep % 2])


def test_worker_thread_and_spawned_process(engine, flags):
    """The reference runs inference on a ThreadPoolExecutor(max_workers=1) thread when a timeout is set
    (runtime/worker_lifecycle.py:98-118) and inside spawned workers: the engine must work
    from a non-main thread and in a fresh spawned process (lazy per-process context)."""
    import multiprocessing as mp
    from concurrent.futures import ThreadPoolExecutor

    from ser_amd import HandcraftedBackend
    from ser_amd.synth import sample_wav_like

    audio = sample_wav_like(seed=5, n_samples=50000)
    backend = HandcraftedBackend(feature_flags=flags)
    here = backend.extract_vector(audio, SR)
    with ThreadPoolExecutor(max_workers=1) as pool:
        there = pool.submit(backend.extract_vector, audio, SR).result(timeout=100)
    np.testing.assert_array_equal(there, here)
    with mp.get_context("spawn").Pool(1) as workers:
        spawned = workers.apply_async(_extract_in_worker, (audio, SR)).get(timeout=110)
    np.testing.assert_array_equal(spawned, here)


def _extract_in_worker(audio, sr):
    from ser_amd import FeatureFlags, HandcraftedBackend

    return HandcraftedBackend(feature_flags=FeatureFlags(tonnetz=False)).extract_vector(audio, sr)


def test_wave_and_workgroup_tuning_kernels_agree_on_a_large_batch(engine, flags):
    """Batches of 2 048 segments or more estimate the tuning with one wave per segment (``k_tuning_waves``), smaller
    ones with one workgroup per segment (``k_segment_finish<1>``), and segments with more than 1 024 peaks fall back
    to the latter inside a large batch: the same clips through one large call and through calls of 512 give the same
    tuning estimates and the same feature bits.  (The small-batch form is the one the oracle tests pin.)"""
    import torch

    from ser_amd.synth import harmonic_clips_torch

    n, length = 3072, 48000
    audio = harmonic_clips_torch(77, n, length, engine.device)
    gen = torch.Generator(device=engine.device).manual_seed(3)
    noisy = torch.arange(0, n, 7, device=engine.device)                      # dense peaks: beyond one wave's staging
    audio[noisy] += 0.3 * torch.randn((len(noisy), length), generator=gen, device=engine.device)
    audio[5] = 0.0                                                            # no peaks at all
    audio[torch.arange(3, n, 11, device=engine.device), : length // 3] *= 1e-5   # quiet stretches: tiles that top_db clips
    audio = audio.reshape(-1).contiguous()
    rng = np.random.default_rng(9)
    lens = rng.integers(2048, length + 1, n)
    lens[:64] = length
    offs = np.arange(n, dtype=np.int64) * length
    big, big_t = engine.features(audio, offs, lens, 16000, flags, return_tuning=True)
    parts, parts_t = [], []
    for lo in range(0, n, 512):
        f, t = engine.features(audio, offs[lo:lo + 512], lens[lo:lo + 512], 16000, flags, return_tuning=True)
        parts.append(f)
        parts_t.append(t)
    small, small_t = torch.cat(parts), torch.cat(parts_t)
    assert torch.equal(big_t, small_t)
    assert torch.equal(big, small)
    assert float(big_t[5]) == 0.0 and len(torch.unique(big_t)) > 20 and bool(torch.isfinite(big).all())


def test_pooling_kernels_bit_exact(engine):
    """mean_std_pool / HandcraftedBackend.pool on the GPU equal numpy bit for bit."""
    from oracle import ser_fast_path as ofp
    from ser_amd import EncodedSequence, HandcraftedBackend, PoolingWindow, mean_std_pool, temporal_pooling_windows

    rng = np.random.default_rng(3)
    emb = (rng.standard_normal((57, 187)) * 30).astype(np.float32)
    starts = np.arange(57, dtype=np.float64)
    ends = np.minimum(starts + 3.0, 59.0)
    enc = EncodedSequence(emb, starts, ends, "handcrafted")
    wins = temporal_pooling_windows(enc, window_size_seconds=10.0, window_stride_seconds=5.0)
    want_w = ofp.temporal_pooling_windows(starts, ends, window_size_seconds=10.0, window_stride_seconds=5.0)
    assert [(w.start_seconds, w.end_seconds) for w in wins] == want_w
    got = mean_std_pool(enc, wins)
    want = ofp.mean_std_pool(emb, starts, ends, want_w)
    assert got.dtype == np.float64 and got.shape == (len(wins), 374)
    np.testing.assert_array_equal(got, want)
    got_mean = HandcraftedBackend().pool(enc, wins)
    np.testing.assert_array_equal(got_mean, ofp.mean_pool(emb, starts, ends, want_w))
    # reference known answers (tests/suites/unit/repr/test_pooling.py:12-45, test_handcrafted_backend.py:66-91)
    small = EncodedSequence(np.asarray([[1, 2], [3, 4], [5, 6]], np.float32), np.asarray([0.0, 1.0, 2.0]),
                            np.asarray([1.0, 2.0, 3.0]), "hf_xlsr")
    np.testing.assert_allclose(
        mean_std_pool(small, [PoolingWindow(0.0, 2.0), PoolingWindow(1.0, 3.0)]),
        [[2.0, 3.0, 1.0, 1.0], [4.0, 5.0, 1.0, 1.0]])
    assert mean_std_pool(small, []).shape == (0, 4)
    with pytest.raises(ValueError, match="outside encoded sequence range"):
        mean_std_pool(small, [PoolingWindow(0.0, 4.0)])


def test_config4_long_form_reduced_and_properties(engine, flags):
    """Config 4 (timeline sliding windows + per-segment pooling): a 150 s recording checked
    window-by-window against the oracle on a sample, every window against a standalone
    GPU call (no copies: windows are descriptors into one buffer), then mean/std pooling."""
    from oracle import ser_fast_path as ofp
    from ser_amd import HandcraftedBackend, mean_std_pool, temporal_pooling_windows
    from ser_amd.synth import harmonic_clip

    rng = np.random.default_rng(4321)
    audio = np.concatenate([harmonic_clip(rng, 5 * SR, SR) for _ in range(30)])
    audio = (audio / np.abs(audio).max()).astype(np.float32)
    backend = HandcraftedBackend(feature_flags=flags)
    enc = backend.encode_sequence(audio, SR)
    assert enc.embeddings.shape == (150, 187)
    bounds = ofp.window_bounds(audio.size, SR, 3, 1)
    for i in (0, 1, 73, 147, 148, 149):
        s, e = bounds[i]
        want = ofp.extract_feature_from_signal(audio[s:e], SR)
        assert_features_close(enc.embeddings[i], want)
        alone, _ = gpu_rows(engine, flags, audio[s:e], [0], [e - s])
        np.testing.assert_array_equal(alone[0].astype(np.float32), enc.embeddings[i])
    wins = temporal_pooling_windows(enc, window_size_seconds=10.0, window_stride_seconds=5.0)
    pooled = mean_std_pool(enc, wins)
    want_w = ofp.temporal_pooling_windows(enc.frame_start_seconds, enc.frame_end_seconds,
                                          window_size_seconds=10.0, window_stride_seconds=5.0)
    np.testing.assert_array_equal(
        pooled, ofp.mean_std_pool(enc.embeddings, enc.frame_start_seconds, enc.frame_end_seconds, want_w))


def test_config4_full_hour_properties(engine, flags):
    """Config 4 at full size (1 h @16 kHz, 3600 windows): size-independent properties only."""
    import torch

    from ser_amd.synth import harmonic_clips_torch

    dev = engine.device
    audio = harmonic_clips_torch(4321, 720, 5 * SR, dev).reshape(-1)       # 57.6 M samples
    n = int(audio.numel())
    starts = np.arange(0, n, SR, dtype=np.int64)
    lens = np.minimum(starts + 3 * SR, n) - starts
    out = engine.features(audio, starts, lens, SR, flags)
    assert tuple(out.shape) == (3600, 187) and bool(torch.isfinite(out).all())
    # any window recomputed alone from a copy of its samples is bit-identical
    for i in (0, 1799, 3598, 3599):
        seg = audio[starts[i] : starts[i] + lens[i]].clone()
        alone = engine.features(seg, [0], [int(lens[i])], SR, flags)
        assert torch.equal(alone[0], out[i])
    host = out.cpu().numpy()
    assert not host[:, 180:].any()
    assert np.all(np.abs(host[:, CHROMA]).max(axis=1) <= 1.0 + 1e-6)
    assert np.all(host[:, MEL] >= 0.0)


def test_config3_full_shard_properties(engine, flags):
    """Config 3 at one GPU's full shard (12 500 clips): duplicated clips give duplicated rows,
    every row is finite, chunking does not change results."""
    import torch

    from ser_amd.synth import harmonic_clips_torch

    base = harmonic_clips_torch(1234, 500, 48000, engine.device)
    audio = base.repeat(25, 1).reshape(-1)                                   # 12 500 clips
    offs = np.arange(12500, dtype=np.int64) * 48000
    out = engine.features(audio, offs, np.full(12500, 48000), SR, flags)
    assert tuple(out.shape) == (12500, 187) and bool(torch.isfinite(out).all())
    first = out[:500]
    for rep in (1, 7, 24):
        assert torch.equal(out[rep * 500 : (rep + 1) * 500], first)
    small = engine.features(base.reshape(-1), offs[:500], np.full(500, 48000), SR, flags)
    assert torch.equal(small, first)


def test_classifier_label_parity(engine, flags):
    """Same sklearn artifact, GPU features vs oracle features: identical labels, proba within 1e-4
    (reference classifier: StandardScaler + MLPClassifier, training_support.py:87-106)."""
    from sklearn.neural_network import MLPClassifier
    from sklearn.pipeline import Pipeline
    from sklearn.preprocessing import StandardScaler

    from oracle import ser_fast_path as ofp
    from ser_amd import HandcraftedBackend
    from ser_amd.audio_io import prepare_audio_buffer
    from ser_amd.synth import ravdess_sine_clip

    clips, labels = [], []
    for actor in (1, 2, 3):
        for emo in range(1, 9):
            clips.append(prepare_audio_buffer(ravdess_sine_clip(actor, emo)))
            labels.append(f"emo{emo}")
    x_ref = np.asarray([ofp.extract_vector(c, SR) for c in clips])
    x_gpu = HandcraftedBackend(feature_flags=flags).extract_vectors(clips, SR)
    model = Pipeline([
        ("scaler", StandardScaler()),
        ("classifier", MLPClassifier(alpha=0.01, batch_size=256, epsilon=1e-08, hidden_layer_sizes=(300,),
                                     learning_rate="adaptive", max_iter=500, random_state=42)),
    ])
    model.fit(x_ref, labels)
    assert list(model.predict(x_gpu)) == list(model.predict(x_ref))
    np.testing.assert_allclose(model.predict_proba(x_gpu), model.predict_proba(x_ref), atol=1e-4)


def test_config5_whisper_logmel(engine):
    """Config 5: whisper-style 80-bin log-mel vs the oracle (atol 1e-5) and vs HF's extractor."""
    from oracle.whisper_logmel import log_mel_spectrogram as oracle_logmel
    from ser_amd import log_mel_spectrogram
    from ser_amd.synth import harmonic_clips

    clips = harmonic_clips(99, 3, 16000 * 5)
    got = log_mel_spectrogram(clips, 80)
    assert got.shape == (3, 80, 500) and got.dtype == np.float32
    for i in range(3):
        np.testing.assert_allclose(got[i], oracle_logmel(clips[i], 80), atol=1e-5)
    one = log_mel_spectrogram(clips[1], 128)
    np.testing.assert_allclose(one, oracle_logmel(clips[1], 128), atol=1e-5)
    odd = log_mel_spectrogram(clips[2][:12345], 80)
    np.testing.assert_allclose(odd, oracle_logmel(clips[2][:12345], 80), atol=1e-5)
    from transformers.models.whisper.feature_extraction_whisper import WhisperFeatureExtractor

    hf = WhisperFeatureExtractor()._np_extract_fbank_features(clips[:1].astype(np.float64), "cpu")
    np.testing.assert_allclose(got[0], hf[0][:, :500], atol=1e-5)


def test_config5_full_length_clips_vs_hf_extractor(engine):
    """Config 5 at its real clip length: 30 s clips -> [B, 80, 3000], every value against HF's
    WhisperFeatureExtractor (numpy path; the pinned stand-in for whisper.log_mel_spectrogram), atol 1e-5,
    and against the oracle for the 128-bin variant."""
    from transformers.models.whisper.feature_extraction_whisper import WhisperFeatureExtractor

    from oracle.whisper_logmel import log_mel_spectrogram as oracle_logmel
    from ser_amd import log_mel_spectrogram
    from ser_amd.synth import harmonic_clips

    clips = harmonic_clips(99, 6, 16000 * 30)
    clips[4, 200000:] = 0.0                      # long silence: the per-clip max - 8 clamp takes over
    clips[5] *= np.float32(1e-3)                 # quiet clip
    got = log_mel_spectrogram(clips, 80)
    assert got.shape == (6, 80, 3000) and got.dtype == np.float32
    hf = WhisperFeatureExtractor()._np_extract_fbank_features(clips.astype(np.float64), "cpu")
    np.testing.assert_allclose(got, np.asarray(hf)[:, :, :3000], atol=1e-5)
    wide = log_mel_spectrogram(clips[:2], 128)
    for i in range(2):
        np.testing.assert_allclose(wide[i], oracle_logmel(clips[i], 128), atol=1e-5)


def test_against_committed_golden_vectors(engine, flags):
    """GPU path vs tests/golden/fast_path_golden.npz (no oracle run needed on the GPU box)."""
    from pathlib import Path

    from ser_amd import HandcraftedBackend, log_mel_spectrogram
    from ser_amd.audio_io import prepare_audio_buffer
    from ser_amd.synth import harmonic_clips, ravdess_sine_clip, sample_wav_like

    g = np.load(Path(__file__).parent / "golden" / "fast_path_golden.npz")
    clips = harmonic_clips(1234, 6, 48000)
    got, tun = gpu_rows(engine, flags, clips.reshape(-1), [i * 48000 for i in range(6)], [48000] * 6)
    np.testing.assert_allclose(tun, g["clips_tuning"], atol=1e-6)
    assert_features_close(got, g["clips_features"])
    backend = HandcraftedBackend(feature_flags=flags)
    enc = backend.encode_sequence(sample_wav_like(), SR)
    assert_features_close(enc.embeddings, g["sample_embeddings"])
    np.testing.assert_array_equal(enc.frame_start_seconds, g["sample_starts"])
    np.testing.assert_array_equal(enc.frame_end_seconds, g["sample_ends"])
    rav = [prepare_audio_buffer(ravdess_sine_clip(a, e)) for a in (1, 2) for e in range(1, 9)]
    assert_features_close(backend.extract_vectors(rav, SR), g["ravdess_features"])
    rng = np.random.default_rng(5)
    x = rng.standard_normal(40000).astype(np.float32)
    x /= np.abs(x).max()
    got, tun = gpu_rows(engine, flags, x, g["ragged_offsets"], g["ragged_lengths"])
    np.testing.assert_allclose(tun, g["ragged_tuning"], atol=1e-6)
    assert_features_close(got, g["ragged_features"])
    np.testing.assert_allclose(log_mel_spectrogram(harmonic_clips(99, 2, 32000), 80), g["whisper_logmel80"], atol=1e-5)


@pytest.mark.gpu
def test_native_rccl_row_gather_world1():
    """serhip_comm_* / serhip_allgather_rows through the C ABI on a one-rank communicator (the only size a
    one-GPU box can form: RCCL refuses two ranks on one device).  The gathered matrix is the shard, in place in
    the output buffer, on torch's current stream; argument errors come back as messages, not crashes."""
    import torch

    from ser_amd._native import SerhipError
    from ser_amd.distributed import COMM_ID_BYTES, NativeRowGather, new_comm_id

    comm_id = new_comm_id()
    assert len(comm_id) == COMM_ID_BYTES and any(comm_id)
    gather = NativeRowGather(comm_id, rank=0, world=1)
    try:
        local = torch.arange(5 * 187, dtype=torch.float32, device="cuda").reshape(5, 187) * 0.5
        full = gather.gather(local, [5])
        torch.cuda.synchronize()
        assert full.data_ptr() != local.data_ptr()
        assert torch.equal(full, local)
        empty = gather.gather(local[:0], [0])
        assert tuple(empty.shape) == (0, 187)
        with pytest.raises(ValueError):
            gather.gather(local, [4])
        with pytest.raises(ValueError):
            gather.gather(local, [5, 5])
        import ctypes

        from ser_amd._native import check, load_library
        rows = (ctypes.c_int64 * 1)(-1)
        with pytest.raises(SerhipError, match="negative row count"):
            check(load_library().serhip_allgather_rows(gather._handle, None, rows, 187, ctypes.c_void_p(full.data_ptr()), None))
    finally:
        gather.close()
    with pytest.raises(ValueError):
        NativeRowGather(b"short", 0, 1)
