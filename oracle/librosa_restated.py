"""numpy/scipy restatement of the librosa 0.11.0 routines on the fast path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  librosa is a
third-party dependency of the reference (``uv.lock:867-868``), absent from
``/root/reference`` and from this image, so each function restates librosa's
published algorithm and cites the reference call site that reaches it
(``ser/_internal/utils/dsp.py``).  dtype placement (f64 window x f32 frame,
complex64 STFT, f32 filterbanks, f32 reductions) follows librosa, because the
GPU path is judged against it at rtol 1e-4.

PARITY UNPINNED for ``piptrack`` / ``estimate_tuning`` / ``pitch_tuning`` (no
independent implementation in this image); everything else is cross-checked in
``tests/test_oracle_pins.py``.
"""

from __future__ import annotations

import warnings

import numpy as np
import scipy.fft
import scipy.signal


class ParameterError(Exception):
    """Stand-in for ``librosa.util.exceptions.ParameterError``."""


# --------------------------------------------------------------------------
# STFT  (librosa.core.spectrum.stft; reached from dsp.py:100,108,122)
# --------------------------------------------------------------------------
def stft_frame_count(n_samples: int, n_fft: int, hop_length: int) -> int:
    """Number of STFT columns for ``center=True`` (zero padding n_fft//2 per side)."""
    return 1 + (n_samples + 2 * (n_fft // 2) - n_fft) // hop_length


def stft(
    y: np.ndarray,
    *,
    n_fft: int = 2048,
    hop_length: int | None = None,
) -> np.ndarray:
    """``librosa.stft(y, n_fft=..., hop_length=...)`` with 0.11 defaults.

    win_length = n_fft, hop_length = win_length // 4 when None, periodic Hann
    (float64, ``scipy.signal.get_window('hann', n, fftbins=True)``),
    ``center=True`` with ``pad_mode='constant'`` (zeros, the 0.10+ default),
    float64 window x float32 frame -> double rfft -> rounded to complex64.
    Returns ``(1 + n_fft//2, n_frames)`` complex64.
    """
    y = np.asarray(y)
    if y.ndim != 1:
        raise ParameterError("stft restatement handles mono signals only")
    win_length = n_fft
    if hop_length is None:
        hop_length = int(win_length // 4)
    if hop_length <= 0:
        raise ParameterError(f"hop_length={hop_length} must be a positive integer")
    if n_fft > y.shape[-1]:
        warnings.warn(
            f"n_fft={n_fft} is too large for input signal of length={y.shape[-1]}",
            stacklevel=2,
        )
    window = scipy.signal.get_window("hann", win_length, fftbins=True)  # float64
    padded = np.pad(y, (n_fft // 2, n_fft // 2), mode="constant")
    frames = np.lib.stride_tricks.sliding_window_view(padded, n_fft)[::hop_length]
    # (T, n_fft) float64 product, double-precision real FFT
    spectrum = scipy.fft.rfft(window[np.newaxis, :] * frames, axis=-1)
    out = np.empty((1 + n_fft // 2, frames.shape[0]), dtype=np.complex64)
    out[...] = spectrum.T
    return out


# --------------------------------------------------------------------------
# power_to_db  (librosa.core.spectrum.power_to_db; dsp.py:101-104 and inside mfcc)
# --------------------------------------------------------------------------
def power_to_db(S, *, ref=1.0, amin: float = 1e-10, top_db: float | None = 80.0):
    """``10 log10(max(amin, S)) - 10 log10(max(amin, ref))``, floored at max - top_db."""
    S = np.asarray(S)
    if amin <= 0:
        raise ParameterError("amin must be strictly positive")
    magnitude = np.abs(S) if np.issubdtype(S.dtype, np.complexfloating) else S
    ref_value = ref(magnitude) if callable(ref) else np.abs(ref)
    log_spec = 10.0 * np.log10(np.maximum(amin, magnitude))
    log_spec -= 10.0 * np.log10(np.maximum(amin, ref_value))
    if top_db is not None:
        if top_db < 0:
            raise ParameterError("top_db must be non-negative")
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


# --------------------------------------------------------------------------
# Slaney mel filterbank (librosa.filters.mel; inside melspectrogram / mfcc)
# --------------------------------------------------------------------------
def hz_to_mel(frequencies):
    frequencies = np.asanyarray(frequencies, dtype=np.float64)
    f_min, f_sp = 0.0, 200.0 / 3
    mels = (frequencies - f_min) / f_sp
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if frequencies.ndim:
        log_t = frequencies >= min_log_hz
        mels[log_t] = min_log_mel + np.log(frequencies[log_t] / min_log_hz) / logstep
    elif frequencies >= min_log_hz:
        mels = min_log_mel + np.log(frequencies / min_log_hz) / logstep
    return mels


def mel_to_hz(mels):
    mels = np.asanyarray(mels, dtype=np.float64)
    f_min, f_sp = 0.0, 200.0 / 3
    freqs = f_min + f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if mels.ndim:
        log_t = mels >= min_log_mel
        freqs[log_t] = min_log_hz * np.exp(logstep * (mels[log_t] - min_log_mel))
    elif mels >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (mels - min_log_mel))
    return freqs


def mel_frequencies(n_mels: int, *, fmin: float, fmax: float) -> np.ndarray:
    return mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels))


def fft_frequencies(*, sr: float, n_fft: int) -> np.ndarray:
    return np.fft.rfftfreq(n=n_fft, d=1.0 / sr)


def mel_filter(*, sr: float, n_fft: int, n_mels: int = 128, fmin: float = 0.0,
               fmax: float | None = None) -> np.ndarray:
    """``librosa.filters.mel(sr, n_fft, n_mels=128, htk=False, norm='slaney')`` -> f32."""
    if fmax is None:
        fmax = float(sr) / 2
    weights = np.zeros((n_mels, int(1 + n_fft // 2)), dtype=np.float32)
    fftfreqs = fft_frequencies(sr=sr, n_fft=n_fft)
    mel_f = mel_frequencies(n_mels + 2, fmin=fmin, fmax=fmax)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def melspectrogram(y: np.ndarray, *, sr: int, n_fft: int, hop_length: int = 512,
                   n_mels: int = 128) -> np.ndarray:
    """``librosa.feature.melspectrogram(y=y, sr=sr, n_fft=n_fft)`` (power=2, hop 512)."""
    S = np.abs(stft(y, n_fft=n_fft, hop_length=hop_length)) ** 2.0  # f32
    mel_basis = mel_filter(sr=sr, n_fft=n_fft, n_mels=n_mels)
    return np.einsum("ft,mf->mt", S, mel_basis, optimize=True)


def mfcc(y: np.ndarray, *, sr: int, n_mfcc: int, n_fft: int) -> np.ndarray:
    """``librosa.feature.mfcc(y=y, sr=sr, n_mfcc=40, n_fft=n_fft)`` (dsp.py:108).

    dct_type=2, norm='ortho', lifter=0; log-mel via ``power_to_db`` defaults
    (ref=1, amin=1e-10, top_db=80 against the whole-array max).
    """
    S = power_to_db(melspectrogram(y, sr=sr, n_fft=n_fft))
    return scipy.fft.dct(S, axis=-2, type=2, norm="ortho")[:n_mfcc, :]


# --------------------------------------------------------------------------
# chroma  (librosa.feature.chroma_stft; dsp.py:115 with S = |STFT| magnitude)
# --------------------------------------------------------------------------
def _parabolic_interpolation(x: np.ndarray) -> np.ndarray:
    """librosa.core.pitch._parabolic_interpolation along axis 0 (frequency).

    numba stencil ``a = x[1] + x[-1] - 2*x[0]; b = (x[1] - x[-1]) / 2``: the
    float32 sums/differences are formed first, the integer literals then
    promote to float64 (numba's int64 x float32 rule), so ``-b / a`` is a double
    division whose result is stored back into the float32 output.  Edges 0.
    PARITY UNPINNED (numba promotion recalled, not executed).
    """
    shifts = np.zeros_like(x)
    if x.shape[0] < 3:
        return shifts
    s = (x[2:] + x[:-2]).astype(np.float64)            # f32 add, then widened
    a = s - 2.0 * x[1:-1].astype(np.float64)
    b = (x[2:] - x[:-2]).astype(np.float64) / 2.0      # f32 subtract, then widened
    with np.errstate(divide="ignore", invalid="ignore"):
        val = np.where(np.abs(b) >= np.abs(a), 0.0, -b / a)
    shifts[1:-1] = val.astype(x.dtype)
    return shifts


def localmax(x: np.ndarray) -> np.ndarray:
    """librosa.util.localmax along axis 0: ``x[i] > x[i-1] and x[i] >= x[i+1]``.

    First row is never a maximum; last row is one iff ``x[-1] > x[-2]``.
    """
    lmax = np.zeros(x.shape, dtype=bool)
    if x.shape[0] >= 3:
        lmax[1:-1] = (x[1:-1] > x[:-2]) & (x[1:-1] >= x[2:])
    if x.shape[0] >= 2:
        lmax[-1] = x[-1] > x[-2]
    return lmax


def piptrack(S: np.ndarray, *, sr: int, n_fft: int, fmin: float = 150.0,
             fmax: float = 4000.0, threshold: float = 0.1):
    """``librosa.piptrack(S=S, sr=sr, n_fft=n_fft)`` -> (pitches, mags), both f32 like S."""
    if n_fft // 2 + 1 != S.shape[-2]:          # _spectrogram(): infer n_fft on mismatch
        n_fft = 2 * (S.shape[-2] - 1)
    S = np.abs(S)
    fmin = np.maximum(fmin, 0)
    fmax = np.minimum(fmax, float(sr) / 2)
    fft_freqs = fft_frequencies(sr=sr, n_fft=n_fft)
    avg = np.gradient(S, axis=-2)
    shift = _parabolic_interpolation(S)
    dskew = 0.5 * avg * shift
    pitches = np.zeros_like(S)
    mags = np.zeros_like(S)
    freq_mask = ((fmin <= fft_freqs) & (fft_freqs < fmax))[:, np.newaxis]
    ref_value = threshold * np.max(S, axis=-2)
    ref_value = np.expand_dims(ref_value, -2)
    idx = np.nonzero(freq_mask & localmax(S * (S > ref_value)))
    pitches[idx] = (idx[-2] + shift[idx]) * float(sr) / n_fft
    mags[idx] = S[idx] + dskew[idx]
    return pitches, mags


def hz_to_octs(frequencies, *, tuning: float = 0.0, bins_per_octave: int = 12):
    A440 = 440.0 * 2.0 ** (tuning / bins_per_octave)
    return np.log2(np.asanyarray(frequencies) / (float(A440) / 16))


def pitch_tuning(frequencies, *, resolution: float = 0.01, bins_per_octave: int = 12) -> float:
    """``librosa.pitch_tuning``: histogram arg-max of the semitone residual (left edge)."""
    frequencies = np.atleast_1d(frequencies)
    frequencies = frequencies[frequencies > 0]
    if not np.any(frequencies):
        warnings.warn("Trying to estimate tuning from empty frequency set.", stacklevel=2)
        return 0.0
    residual = np.mod(bins_per_octave * hz_to_octs(frequencies), 1.0)
    residual[residual >= 0.5] -= 1.0
    bins = np.linspace(-0.5, 0.5, int(np.ceil(1.0 / resolution)) + 1)
    counts, tuning = np.histogram(residual, bins)
    return float(tuning[np.argmax(counts)])


def estimate_tuning(S: np.ndarray, *, sr: int, n_fft: int = 2048,
                    bins_per_octave: int = 12, return_debug: bool = False):
    """``librosa.estimate_tuning(S=S, sr=sr, bins_per_octave=12)``.

    ``chroma_stft`` does NOT forward its ``n_fft`` here, so ``n_fft`` keeps the
    2048 default and ``piptrack`` re-infers it from ``S.shape`` on mismatch.
    """
    pitch, mag = piptrack(S, sr=sr, n_fft=n_fft)
    pitch_mask = pitch > 0
    if pitch_mask.any():
        threshold = np.median(mag[pitch_mask])
    else:
        threshold = 0.0
    selected = pitch[(mag >= threshold) & pitch_mask]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tuning = pitch_tuning(selected, resolution=0.01, bins_per_octave=bins_per_octave)
    if return_debug:
        return tuning, {"n_peaks": int(pitch_mask.sum()), "threshold": float(threshold),
                        "selected": selected}
    return tuning


def tiny(x) -> float:
    x = np.asarray(x)
    if np.issubdtype(x.dtype, np.floating) or np.issubdtype(x.dtype, np.complexfloating):
        dtype = x.dtype
    else:
        dtype = np.dtype(np.float32)
    return float(np.finfo(dtype).tiny)


def normalize(S: np.ndarray, *, norm=np.inf, axis: int = 0) -> np.ndarray:
    """``librosa.util.normalize`` (threshold=tiny(S), fill=None)."""
    threshold = tiny(S)
    if not np.isfinite(S).all():
        raise ParameterError("Input must be finite")
    mag = np.abs(S).astype(float)
    if norm == np.inf:
        length = np.max(mag, axis=axis, keepdims=True)
    elif norm > 0:
        length = np.sum(mag**norm, axis=axis, keepdims=True) ** (1.0 / norm)
    else:
        raise ParameterError(f"Unsupported norm: {norm!r}")
    small_idx = length < threshold
    Snorm = np.empty_like(S)
    length[small_idx] = 1.0
    Snorm[:] = S / length
    return Snorm


def chroma_filter(*, sr: float, n_fft: int, tuning: float = 0.0, n_chroma: int = 12,
                  ctroct: float = 5.0, octwidth: float = 2) -> np.ndarray:
    """``librosa.filters.chroma(sr, n_fft, tuning, norm=2, base_c=True)`` -> f32 (12, 1+n_fft//2)."""
    frequencies = np.linspace(0, sr, n_fft, endpoint=False)[1:]
    frqbins = n_chroma * hz_to_octs(frequencies, tuning=tuning, bins_per_octave=n_chroma)
    frqbins = np.concatenate(([frqbins[0] - 1.5 * n_chroma], frqbins))
    binwidthbins = np.concatenate((np.maximum(frqbins[1:] - frqbins[:-1], 1.0), [1]))
    D = np.subtract.outer(frqbins, np.arange(0, n_chroma, dtype="d")).T
    n_chroma2 = np.round(float(n_chroma) / 2)
    D = np.remainder(D + n_chroma2 + 10 * n_chroma, n_chroma) - n_chroma2
    wts = np.exp(-0.5 * (2 * D / np.tile(binwidthbins, (n_chroma, 1))) ** 2)
    wts = normalize(wts, norm=2, axis=0)
    wts *= np.tile(np.exp(-0.5 * (((frqbins / n_chroma - ctroct) / octwidth) ** 2)), (n_chroma, 1))
    wts = np.roll(wts, -3 * (n_chroma // 12), axis=0)
    return np.ascontiguousarray(wts[:, : int(1 + n_fft / 2)], dtype=np.float32)


def chroma_stft(S: np.ndarray, *, sr: int, n_fft: int, return_tuning: bool = False):
    """``librosa.feature.chroma_stft(S=S, sr=sr, n_fft=n_fft)``; S is used AS GIVEN.

    dsp.py:115 passes the STFT *magnitude*, so the chroma projection is of
    magnitudes (not power).  norm=inf per frame, tuning estimated from S.
    """
    if n_fft // 2 + 1 != S.shape[-2]:
        n_fft = 2 * (S.shape[-2] - 1)
    tuning = estimate_tuning(S, sr=sr, bins_per_octave=12)
    chromafb = chroma_filter(sr=sr, n_fft=n_fft, tuning=tuning)
    raw_chroma = np.einsum("cf,ft->ct", chromafb, S, optimize=True)
    out = normalize(raw_chroma, norm=np.inf, axis=-2)
    if return_tuning:
        return out, tuning
    return out


# --------------------------------------------------------------------------
# spectral contrast (librosa.feature.spectral_contrast; dsp.py:129 with S = power dB)
# --------------------------------------------------------------------------
def spectral_contrast(S: np.ndarray, *, sr: int, n_fft: int, fmin: float = 200.0,
                      n_bands: int = 6, quantile: float = 0.02) -> np.ndarray:
    """S is used AS GIVEN (dsp.py passes a dB-scaled array <= 0, so the result is 0)."""
    if n_fft // 2 + 1 != S.shape[-2]:
        n_fft = 2 * (S.shape[-2] - 1)
    freq = fft_frequencies(sr=sr, n_fft=n_fft)
    octa = np.zeros(n_bands + 2)
    octa[1:] = fmin * (2.0 ** np.arange(0, n_bands + 1))
    if np.any(octa[:-1] >= 0.5 * sr):
        raise ParameterError("Frequency band exceeds Nyquist. Reduce either fmin or n_bands.")
    shape = list(S.shape)
    shape[-2] = n_bands + 1
    valley = np.zeros(shape)
    peak = np.zeros_like(valley)
    for k, (f_low, f_high) in enumerate(zip(octa[:-1], octa[1:])):
        current_band = np.logical_and(freq >= f_low, freq <= f_high)
        idx = np.flatnonzero(current_band)
        if k > 0:
            current_band[idx[0] - 1] = True
        if k == n_bands:
            current_band[idx[-1] + 1 :] = True
        sub_band = S[current_band, :]
        if k < n_bands:
            sub_band = sub_band[:-1, :]
        n_idx = np.rint(quantile * np.sum(current_band))
        n_idx = int(np.maximum(n_idx, 1))
        sortedr = np.sort(sub_band, axis=-2)
        valley[k, :] = np.mean(sortedr[:n_idx, :], axis=-2)
        peak[k, :] = np.mean(sortedr[-n_idx:, :], axis=-2)
    return power_to_db(peak) - power_to_db(valley)
