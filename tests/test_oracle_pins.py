"""Pins for the CPU oracle (oracle/): independent in-container implementations of the same
published formulas, the committed golden vectors, and analytic known answers.

The reference's own tests hold no numeric vectors for the librosa arithmetic (they stub it,
tests/suites/unit/utils/test_dsp.py:21-48), so piptrack / estimate_tuning stay "parity
unpinned" (oracle/__init__.py); everything else on the path is cross-checked here.
"""

from pathlib import Path

import numpy as np
import pytest
import scipy.fft

from oracle import librosa_restated as lr
from oracle import ser_fast_path as ofp
from oracle.whisper_logmel import log_mel_spectrogram

GOLDEN = Path(__file__).parent / "golden" / "fast_path_golden.npz"
SR = 16000


def test_mel_filterbank_matches_hf_slaney():
    from transformers.audio_utils import mel_filter_bank

    for sr, n_fft, n_mels in ((16000, 2048, 128), (16000, 400, 80), (48000, 2048, 128), (22050, 1500, 128)):
        hf = mel_filter_bank(1 + n_fft // 2, n_mels, 0.0, sr / 2, sr, norm="slaney", mel_scale="slaney").T
        ours = lr.mel_filter(sr=sr, n_fft=n_fft, n_mels=n_mels)
        assert ours.dtype == np.float32 and ours.shape == (n_mels, 1 + n_fft // 2)
        np.testing.assert_allclose(ours, hf, rtol=0, atol=1e-8)


def test_chroma_filterbank_matches_hf_port_of_librosa():
    from transformers.audio_utils import chroma_filter_bank

    for n_fft in (2048, 1023, 512):
        for tuning in (0.0, -0.5, 0.17, 0.49):
            hf = chroma_filter_bank(n_fft, 12, SR, tuning=tuning, power=2.0)
            ours = lr.chroma_filter(sr=SR, n_fft=n_fft, tuning=tuning)
            assert ours.shape == (12, 1 + n_fft // 2)
            np.testing.assert_allclose(ours, hf, rtol=0, atol=6e-8)


def test_power_to_db_matches_hf():
    from transformers.audio_utils import power_to_db as hf_db

    rng = np.random.default_rng(0)
    S = (rng.random((40, 30)) ** 6).astype(np.float32)
    np.testing.assert_allclose(lr.power_to_db(S), hf_db(S.astype(np.float64), db_range=80.0), atol=2e-5)
    db = lr.power_to_db(S, ref=np.max)
    assert db.max() == 0.0 and db.min() >= -80.0


def test_stft_matches_torch_stft():
    import torch

    rng = np.random.default_rng(1)
    for n, n_fft, hop in ((48000, 2048, None), (5937, 2048, 512), (1023, 1023, None), (1500, 1500, 512)):
        y = rng.standard_normal(n).astype(np.float32)
        ours = lr.stft(y, n_fft=n_fft, hop_length=hop)
        h = hop if hop is not None else n_fft // 4
        window = torch.from_numpy(scipy.signal.get_window("hann", n_fft, fftbins=True))
        ref = torch.stft(torch.from_numpy(y).double(), n_fft, hop_length=h, window=window, center=True,
                         pad_mode="constant", return_complex=True).numpy()
        assert ours.shape == ref.shape == (1 + n_fft // 2, lr.stft_frame_count(n, n_fft, h))
        assert ours.dtype == np.complex64
        np.testing.assert_allclose(ours, ref, rtol=0, atol=2e-5 * np.abs(ref).max())


def test_mfcc_is_ortho_dct_of_log_mel():
    rng = np.random.default_rng(2)
    y = rng.standard_normal(20000).astype(np.float32)
    m = lr.mfcc(y, sr=SR, n_mfcc=40, n_fft=2048)
    logmel = lr.power_to_db(lr.melspectrogram(y, sr=SR, n_fft=2048))
    k = np.arange(40)[:, None]
    n = np.arange(128)[None, :]
    basis = np.sqrt(2.0 / 128) * np.cos(np.pi * (2 * n + 1) * k / 256.0)
    basis[0] *= np.sqrt(0.5)
    np.testing.assert_allclose(m, basis @ logmel.astype(np.float64), rtol=1e-5, atol=1e-3)


def test_log_mel_and_mfcc_chain_matches_hf_spectrogram_end_to_end():
    """The whole mel / MFCC lane of ``dsp.py:106-125`` against ONE call of an independent implementation:
    ``transformers.audio_utils.spectrogram`` (its own framing, zero padding, windowing, rfft, mel projection and dB
    conversion with an 80 dB range against the global maximum) -- not piece by piece as above.  Pins the composition:
    centre padding with zeros, hop 512, the frame count, power (not magnitude) into the mel bank, ``top_db`` against the
    segment-global maximum, and the time means the reference takes."""
    import scipy.fft
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function

    rng = np.random.default_rng(12)
    bank = mel_filter_bank(num_frequency_bins=1025, num_mel_filters=128, min_frequency=0.0, max_frequency=SR / 2,
                           sampling_rate=SR, norm="slaney", mel_scale="slaney")
    window = window_function(2048, "hann", periodic=True)
    for n, loud in ((48000, 1.0), (21937, 0.05), (2048, 1.0)):
        t = np.arange(n) / SR
        y = (loud * (0.6 * np.sin(2 * np.pi * 233.0 * t) + 0.3 * np.sin(2 * np.pi * 1470.0 * t) * np.hanning(n))
             + 1e-3 * rng.standard_normal(n)).astype(np.float32)
        y[: n // 5] *= 1e-4                                    # a quiet stretch: frames that top_db clips
        power = spectrogram(y.astype(np.float64), window, frame_length=2048, hop_length=512, fft_length=2048, power=2.0,
                            center=True, pad_mode="constant", mel_filters=bank, mel_floor=0.0, dtype=np.float64)
        logmel = spectrogram(y.astype(np.float64), window, frame_length=2048, hop_length=512, fft_length=2048, power=2.0,
                             center=True, pad_mode="constant", mel_filters=bank, mel_floor=1e-10, log_mel="dB",
                             reference=1.0, min_value=1e-10, db_range=80.0, dtype=np.float64)
        assert power.shape == logmel.shape == (128, 1 + n // 512)
        ours_power = lr.melspectrogram(y, sr=SR, n_fft=2048)
        np.testing.assert_allclose(ours_power, power, rtol=2e-4, atol=1e-6 * power.max())
        ours_db = lr.power_to_db(ours_power)
        assert n < 4096 or (ours_db == ours_db.max() - 80.0).any()          # the clip is exercised on the long clips
        np.testing.assert_allclose(ours_db, logmel, rtol=0, atol=2e-3)
        want_mfcc = scipy.fft.dct(logmel, type=2, norm="ortho", axis=0)[:40].mean(axis=1)
        np.testing.assert_allclose(lr.mfcc(y, sr=SR, n_mfcc=40, n_fft=2048).mean(axis=1), want_mfcc, rtol=1e-4, atol=2e-3)


def test_pure_tone_tuning_and_chroma_known_answer():
    """A 452.9 Hz tone is A4 + 50 cents: tuning estimate -0.5/+0.5 boundary aside, a 446.4 Hz
    tone (+25 cents) must estimate +0.25 and put the chroma peak on A (index 9, C-based)."""
    t = np.arange(48000) / SR
    f = 440.0 * 2 ** (0.25 / 12)
    y = np.sin(2 * np.pi * f * t).astype(np.float32)
    S = np.abs(lr.stft(y, n_fft=2048))
    tuning = lr.estimate_tuning(S, sr=SR)
    assert tuning == pytest.approx(0.25, abs=0.011)
    chroma = lr.chroma_stft(S, sr=SR, n_fft=2048)
    assert chroma.shape == (12, 94) and int(np.argmax(chroma.mean(axis=1))) == 9
    assert chroma.max() == pytest.approx(1.0)


def test_contrast_is_identically_zero_and_validates_sample_rate():
    rng = np.random.default_rng(3)
    y = rng.standard_normal(9000).astype(np.float32)
    S = np.abs(lr.stft(y, n_fft=2048))
    db = lr.power_to_db(np.square(S), ref=np.max)
    c = lr.spectral_contrast(db, sr=SR, n_fft=2048)
    assert c.shape == (7, S.shape[1]) and not c.any()
    with pytest.raises(lr.ParameterError, match="exceeds Nyquist"):
        lr.spectral_contrast(db, sr=12800, n_fft=2048)


def test_extract_feature_blocks_dims_and_errors():
    """193 = 40+12+128+7+6 with every flag, 187 in scope, block order mfcc|chroma|mel|contrast
    (tests/suites/unit/repr/test_handcrafted_backend.py:13-24, test_feature_extractor.py:64-76)."""
    assert ofp.feature_dim(ofp.Flags(tonnetz=True)) == 193
    assert ofp.feature_dim(ofp.Flags()) == 187
    rng = np.random.default_rng(4)
    y = rng.standard_normal(6000).astype(np.float32)
    y /= np.abs(y).max()
    v = ofp.extract_feature_from_signal(y, SR)
    assert v.dtype == np.float64 and v.shape == (187,)
    np.testing.assert_array_equal(v[:40], ofp.extract_feature_from_signal(y, SR, flags=ofp.Flags(True, False, False, False)))
    np.testing.assert_array_equal(v[52:180], ofp.extract_feature_from_signal(y, SR, flags=ofp.Flags(False, False, True, False)))
    assert not v[180:].any()
    with pytest.raises(ValueError, match="no samples"):
        ofp.extract_feature_from_signal(np.zeros(0, np.float32), SR)
    with pytest.raises(ValueError, match="positive integer"):
        ofp.extract_feature_from_signal(y, 0)
    with pytest.raises(ValueError, match="mono"):
        ofp.extract_feature_from_signal(np.zeros((3, 2), np.float32), SR)


def test_reference_windowing_and_pooling_known_answers():
    """Known answers the reference's own tests hold."""
    assert ofp.window_bounds(10, 2, 2, 1) == [(0, 4), (2, 6), (4, 8), (6, 10), (8, 10)]
    starts = np.asarray([0.0, 1.0, 2.0])
    ends = np.asarray([1.0, 2.0, 3.0])
    emb = np.asarray([[1, 2], [3, 4], [5, 6]], np.float32)
    np.testing.assert_allclose(ofp.mean_std_pool(emb, starts, ends, [(0.0, 2.0), (1.0, 3.0)]),
                               [[2, 3, 1, 1], [4, 5, 1, 1]])
    assert ofp.mean_std_pool(emb, starts, ends, []).shape == (0, 4)
    np.testing.assert_allclose(ofp.mean_pool(emb, starts, ends, [(0.0, 2.0), (1.0, 3.0)]), [[2, 3], [4, 5]])
    with pytest.raises(ValueError, match="outside encoded sequence range"):
        ofp.mean_std_pool(emb, starts, ends, [(0.0, 4.0)])
    s4, e4 = np.asarray([0.0, 1.0, 2.0, 3.0]), np.asarray([1.0, 2.0, 3.0, 4.0])
    assert ofp.temporal_pooling_windows(s4, e4, window_size_seconds=2.0, window_stride_seconds=1.5) == [
        (0.0, 2.0), (1.5, 3.5), (2.0, 4.0)]
    assert ofp.temporal_pooling_windows(np.asarray([0.0]), np.asarray([1.0]), window_size_seconds=3.0,
                                        window_stride_seconds=1.0) == [(0.0, 1.0)]
    np.testing.assert_allclose(ofp.prepare_audio_buffer(np.asarray([[0.2, 0.0], [0.4, 0.0]], np.float32)), [0.5, 1.0])


def test_whisper_oracle_matches_hf_extractor():
    from transformers.models.whisper.feature_extraction_whisper import WhisperFeatureExtractor

    from ser_amd.synth import harmonic_clips

    clips = harmonic_clips(99, 2, 16000 * 2)
    fe = WhisperFeatureExtractor()
    hf = fe._np_extract_fbank_features(clips.astype(np.float64), "cpu")
    for i in range(2):
        np.testing.assert_allclose(log_mel_spectrogram(clips[i], 80), hf[i], atol=1e-6)
    assert log_mel_spectrogram(clips[0], 128).shape == (128, 200)


def test_oracle_reproduces_golden_vectors():
    """The committed vectors (tests/golden/make_golden.py) are what the oracle computes today."""
    from ser_amd.synth import harmonic_clips, ravdess_sine_clip, sample_wav_like

    g = np.load(GOLDEN)
    clips = harmonic_clips(1234, 6, 48000)
    for i in (0, 5):
        v, d = ofp.extract_feature_from_signal(clips[i], SR, return_debug=True)
        np.testing.assert_allclose(v, g["clips_features"][i], rtol=1e-6, atol=1e-6)
        assert d["tuning"] == pytest.approx(g["clips_tuning"][i])
    emb, starts, ends = ofp.encode_sequence(sample_wav_like(), SR)
    np.testing.assert_allclose(emb, g["sample_embeddings"], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(starts, [0.0, 1.0, 2.0, 3.0, 4.0])
    np.testing.assert_allclose(ends, [3.0, 4.0, 69937 / 16000, 69937 / 16000, 69937 / 16000])
    rav = ofp.prepare_audio_buffer(ravdess_sine_clip(2, 3))
    np.testing.assert_allclose(ofp.extract_vector(rav, SR), g["ravdess_features"][8 + 2], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(rav, ofp.prepare_audio_buffer(ofp.synthetic_ravdess_clip(2, 3)))
