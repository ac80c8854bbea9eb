"""Hand-derived known answers for the least-pinned part of the oracle: ``piptrack`` /
``estimate_tuning`` / ``pitch_tuning`` (oracle/librosa_restated.py:174-270), on tiny spectra.

The reference holds no numeric vector for the librosa boundary (its tests stub every librosa call,
/root/reference/tests/suites/unit/utils/test_dsp.py:21-48), so parity there stays UNPINNED; what this file
adds is independence from the oracle's own code: every expected number below is worked out in the
comments from librosa 0.11's published formulas, never by calling the oracle.

Set-up used throughout: n_fft = 32 (17 bins), sr = 1760 Hz, so bin k sits at 55 k Hz: bin 4 = 220 Hz,
bin 8 = 440 Hz (A4: residual exactly 0), bin 5 = 275 Hz and bin 10 = 550 Hz (12 log2(f / 27.5) = 39.863 /
51.863: residual 0.863 -> folded to -0.137), bin 6 = 330 Hz and bin 12 = 660 Hz (43.0196 / 55.0196:
residual +0.0196).  piptrack's band is 150 <= f < min(4000, sr/2) = 880: bins 3..15.
"""

import numpy as np
import pytest

SR, N_FFT, BINS = 1760, 32, 17
EDGES = np.linspace(-0.5, 0.5, 101)     # pitch_tuning's histogram edges (resolution 0.01)


def _column(peaks):
    """A magnitude column with symmetric peaks: {bin: height}; both neighbours get height / 2."""
    col = np.zeros(BINS, dtype=np.float32)
    for k, h in peaks.items():
        col[k] = h
    for k, h in peaks.items():
        for n in (k - 1, k + 1):
            if n not in peaks:
                col[n] = max(col[n], h / 2)
    return col


def _eight_peak_spectrum():
    """Four frames, eight peaks (an EVEN count), all symmetric (shift 0, mag = height):
         frame 0: (275 Hz, 4) (440 Hz, 8)        frame 1: (220 Hz, 4) (550 Hz, 8)
         frame 2: (330 Hz, 1) (440 Hz, 3) (660 Hz, 1)      frame 3: (330 Hz, 1)  + a bump below 0.1 * column max
    """
    f0 = _column({5: 4.0, 8: 8.0})
    f1 = _column({4: 4.0, 10: 8.0})
    f2 = _column({6: 1.0, 8: 3.0, 12: 1.0})
    f2[7] = 0.5                                 # shared neighbour of bins 6 and 8: keeps both symmetric
    f2[9] = 0.5
    f3 = _column({6: 1.0})
    f3[12] = 0.05                               # 0.05 <= 0.1 * max(f3) = 0.1: thresholded to 0, never a peak
    return np.stack([f0, f1, f2, f3], axis=1)   # [17 bins, 4 frames] like librosa's S


def test_parabolic_interpolation_and_localmax_rules():
    from oracle import librosa_restated as lr

    # a = x[i+1] + x[i-1] - 2 x[i], b = (x[i+1] - x[i-1]) / 2, shift = -b / a if |b| < |a| else 0; edges 0
    x = np.array([[2.0], [8.0], [4.0], [1.0], [2.0], [4.0]], dtype=np.float32)
    got = lr._parabolic_interpolation(x)[:, 0]
    # i=1: a = 4 + 2 - 16 = -10, b = 1      -> 0.1        (double division, rounded to float32)
    # i=2: a = 1 + 8 - 8  = 1,   b = -3.5   -> |b| >= |a| -> 0
    # i=3: a = 2 + 4 - 2  = 4,   b = -1     -> 0.25
    # i=4: a = 4 + 1 - 4  = 1,   b = 1.5    -> |b| >= |a| -> 0
    np.testing.assert_array_equal(got, np.array([0, np.float32(0.1), 0, 0.25, 0, 0], dtype=np.float32))
    # localmax: x[i] > x[i-1] and x[i] >= x[i+1]; the first row never; the LAST row iff x[-1] > x[-2]
    col = np.array([5, 1, 3, 3, 2, 4], dtype=np.float32)[:, None]
    assert lr.localmax(col)[:, 0].tolist() == [False, False, True, False, False, True]
    assert lr.localmax(np.array([1, 2, 2], np.float32)[:, None])[:, 0].tolist() == [False, True, False]


def test_piptrack_hand_derived_values():
    from oracle import librosa_restated as lr

    # one asymmetric peak: S[4], S[5], S[6] = 2, 8, 4
    #   shift = float32(0.1); pitch = (5 + shift) * sr / n_fft in double = 5.1000000015 * 55 = 280.50000008 -> f32 280.5
    #   np.gradient at bin 5 = (4 - 2) / 2 = 1; dskew = 0.5 * 1 * shift = f32(0.05); mag = f32(8 + 0.05)
    col = np.zeros(BINS, dtype=np.float32)
    col[4:7] = [2.0, 8.0, 4.0]
    col[14] = 0.5                                  # 0.5 <= 0.1 * 8: below the threshold, not a peak
    col[2] = 7.0                                   # a local maximum at 110 Hz: outside the 150 Hz band
    pitches, mags = lr.piptrack(col[:, None], sr=SR, n_fft=N_FFT)
    shift = np.float32(0.1)
    want_pitch = np.float32((5.0 + float(shift)) * float(SR) / N_FFT)
    want_mag = np.float32(8.0) + np.float32(np.float32(0.5) * np.float32(1.0)) * shift
    assert want_pitch == np.float32(280.5)
    assert np.flatnonzero(pitches[:, 0]).tolist() == [5]
    assert pitches[5, 0] == want_pitch and mags[5, 0] == want_mag and pitches.dtype == np.float32
    # the eight-peak spectrum: symmetric peaks -> shift -0.0 / a = 0, pitch = 55 k exactly, mag = height
    pitches, mags = lr.piptrack(_eight_peak_spectrum(), sr=SR, n_fft=N_FFT)
    found = sorted((t, float(pitches[k, t]), float(mags[k, t])) for k, t in zip(*np.nonzero(pitches)))
    assert found == [(0, 275.0, 4.0), (0, 440.0, 8.0), (1, 220.0, 4.0), (1, 550.0, 8.0),
                     (2, 330.0, 1.0), (2, 440.0, 3.0), (2, 660.0, 1.0), (3, 330.0, 1.0)]


def test_pitch_tuning_histogram_rules():
    from oracle import librosa_restated as lr

    # residual = mod(12 log2(f / 27.5), 1), >= 0.5 folded down by 1; bin [e_i, e_i+1); arg-max = FIRST maximum
    assert lr.pitch_tuning(np.array([440.0, 220.0], np.float32)) == EDGES[50] == 0.0
    assert lr.pitch_tuning(np.array([275.0], np.float32)) == EDGES[36]           # -0.1369 in [-0.14, -0.13)
    assert lr.pitch_tuning(np.array([330.0, 660.0, 440.0], np.float32)) == EDGES[51]   # +0.0196 twice beats 0 once
    # a tie between bin 36 and bin 50: the lower bin wins
    assert lr.pitch_tuning(np.array([440.0, 275.0, 550.0, 220.0], np.float32)) == EDGES[36]
    # 27.5 * 2 ** (k / 12 + 0.5 / 12): residual exactly at the fold -> -0.5 -> first bin
    f = np.float32(27.5 * 2.0 ** (40.5 / 12.0))
    r = np.mod(np.float32(12.0) * np.log2(f / np.float32(27.5)), np.float32(1.0))
    assert lr.pitch_tuning(np.array([f])) == (EDGES[0] if r >= 0.5 else EDGES[99])
    assert lr.pitch_tuning(np.zeros(3, np.float32)) == 0.0                        # empty set (warning filtered upstream)


def test_estimate_tuning_even_count_median_and_tie():
    from oracle import librosa_restated as lr

    # eight peak magnitudes sorted: 1 1 1 3 | 4 4 8 8 -> np.median = (3 + 4) / 2 = 3.5
    # selected (mag >= 3.5): (275,4) (440,8) (220,4) (550,8) -> bins 36, 50, 50, 36: a 2-2 tie -> bin 36 -> -0.14
    # wrong medians give other answers: lower middle (3) adds (440,3): bin 50 wins, 0.0; no threshold at all adds
    # 330 / 660 / 330 as well: bins 50 and 51 tie at 3, 0.0 again
    tuning, dbg = lr.estimate_tuning(_eight_peak_spectrum(), sr=SR, n_fft=N_FFT, return_debug=True)
    assert dbg["n_peaks"] == 8 and dbg["threshold"] == 3.5
    assert sorted(dbg["selected"].tolist()) == [220.0, 275.0, 440.0, 550.0]
    assert tuning == EDGES[36] and abs(tuning + 0.14) < 1e-12
    # odd count: drop frame 3 -> 1 1 3 4 4 8 8 -> median 4 -> the same four selected
    tuning7, dbg7 = lr.estimate_tuning(_eight_peak_spectrum()[:, :3], sr=SR, n_fft=N_FFT, return_debug=True)
    assert dbg7["n_peaks"] == 7 and dbg7["threshold"] == 4.0 and tuning7 == EDGES[36]
    # no peaks at all -> 0.0
    assert lr.estimate_tuning(np.zeros((BINS, 3), np.float32), sr=SR, n_fft=N_FFT) == 0.0


def _peak_lists():
    """(pitch, mag) lists per frame of the eight-peak spectrum, as k_stft_frames would have written them."""
    return [[(275.0, 4.0), (440.0, 8.0)], [(220.0, 4.0), (550.0, 8.0)],
            [(330.0, 1.0), (440.0, 3.0), (660.0, 1.0)], [(330.0, 1.0)]]


@pytest.mark.gpu
def test_gpu_tuning_estimate_on_hand_derived_peak_lists():
    """The same hand-derived cases through the device tuning estimate (k_segment_finish<1>: radix-select
    median incl. the lower middle of an even count, float32 residuals, 100-bin histogram, first maximum),
    fed with peak lists through serhip_tuning_from_peaks."""
    from ser_amd.engine import get_engine

    eng = get_engine()
    tune = lambda frames: eng.tuning_from_peaks(frames, 16000)    # noqa: E731  (the plan's rate does not enter)
    assert tune(_peak_lists()) == np.float32(EDGES[36])            # even count, 2-2 tie -> -0.14
    assert tune(_peak_lists()[:3]) == np.float32(EDGES[36])        # odd count, median 4
    assert tune([[(440.0, 3.0)], [(330.0, 1.0), (660.0, 1.0)]]) == np.float32(EDGES[51])   # median 1: all selected
    assert tune([[(440.0, 8.0), (220.0, 8.0)], [(330.0, 1.0), (660.0, 1.0)]]) == np.float32(EDGES[50])
    assert tune([[], [], []]) == 0.0                               # empty peak set -> 0.0 (edge 50)
    # ties INSIDE the median: 1 1 3 3 | 3 3 8 8 -> median 3 (lower middle == upper middle), selected >= 3
    frames = [[(330.0, 1.0), (660.0, 1.0)], [(275.0, 3.0), (550.0, 3.0), (275.0, 3.0)],
              [(440.0, 3.0)], [(440.0, 8.0), (220.0, 8.0)]]
    assert tune(frames) == np.float32(EDGES[36])                   # bin 36: 3, bin 50: 3 -> first
    # more peaks than the kernel stages in LDS (2048): 3 000 peaks at 330 Hz with mag 1, 3 001 at 275 Hz with mag 2
    many = [[(330.0, 1.0)] * 100 for _ in range(30)] + [[(275.0, 2.0)] * 100 for _ in range(30)] + [[(275.0, 2.0)]]
    assert tune(many) == np.float32(EDGES[36])
    # even count in the large case: median (1 + 2) / 2 = 1.5 -> only the 275 Hz peaks
    assert tune(many[:-1]) == np.float32(EDGES[36])
    # a pitch exactly on the fold
    f = float(np.float32(27.5 * 2.0 ** (40.5 / 12.0)))
    r = np.mod(np.float32(12.0) * np.log2(np.float32(f) / np.float32(27.5)), np.float32(1.0))
    assert tune([[(f, 1.0)]]) == np.float32(EDGES[0] if r >= 0.5 else EDGES[99])
