// Host-side construction of the constant tables of the fast feature plan.
//
// Everything is computed in double precision with the same operation order as
// the numpy expressions inside librosa 0.11 (filters.mel, filters.chroma,
// scipy.signal.get_window, scipy.fft.dct ortho) that the reference triggers
// from ser/_internal/utils/dsp.py:100-136, then rounded to float32 where
// librosa rounds.  No GPU call in this file.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

namespace serhip {

constexpr int kNfft = 2048;
constexpr int kHop = 512;
constexpr int kBins = 1025;        // 1 + kNfft/2
constexpr int kMels = 128;
constexpr int kMfcc = 40;
constexpr int kChroma = 12;
constexpr int kTunings = 101;      // np.linspace(-0.5, 0.5, 101) histogram edges
constexpr int kBinStride = 1028;   // padded row length of per-bin tables (16-byte rows)

constexpr double kPi = 3.14159265358979323846;

// np.linspace(start, stop, num) element i (endpoint=True): arange*step + start, last = stop
inline double linspace_at(double start, double stop, int num, int i) {
    if (num == 1) return start;
    if (i == num - 1) return stop;
    const double step = (stop - start) / static_cast<double>(num - 1);
    return static_cast<double>(i) * step + start;
}

// librosa.hz_to_mel / mel_to_hz, Slaney variant (htk=False)
inline double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3;
    const double min_log_hz = 1000.0;
    const double min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    if (f >= min_log_hz) return min_log_mel + std::log(f / min_log_hz) / logstep;
    return f / f_sp;
}
inline double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3;
    const double min_log_hz = 1000.0;
    const double min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    if (m >= min_log_mel) return min_log_hz * std::exp(logstep * (m - min_log_mel));
    return f_sp * m;
}

// mel_frequencies(n_mels + 2, fmin=0, fmax) -> n_mels+2 band edges in Hz
inline std::vector<double> mel_edges(int n_mels, double fmax) {
    std::vector<double> e(n_mels + 2);
    const double lo = hz_to_mel(0.0), hi = hz_to_mel(fmax);
    for (int i = 0; i < n_mels + 2; ++i) e[i] = mel_to_hz(linspace_at(lo, hi, n_mels + 2, i));
    return e;
}

// np.fft.rfftfreq(n, d=1/sr)[k]
inline double rfft_freq(int k, int n_fft, int sr) {
    const double d = 1.0 / static_cast<double>(sr);
    const double val = 1.0 / (static_cast<double>(n_fft) * d);
    return static_cast<double>(k) * val;
}

// one Slaney-normalised triangular weight, rounded like librosa: f32(max(0,min(lower,upper)))
// then *= enorm in double, rounded to f32 again
inline float mel_weight(const double* mel_f, int i, double freq) {
    const double fd0 = mel_f[i + 1] - mel_f[i];
    const double fd1 = mel_f[i + 2] - mel_f[i + 1];
    const double lower = -(mel_f[i] - freq) / fd0;
    const double upper = (mel_f[i + 2] - freq) / fd1;
    double w = lower < upper ? lower : upper;
    if (!(w > 0.0)) w = 0.0;
    const float w32 = static_cast<float>(w);
    const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
    return static_cast<float>(static_cast<double>(w32) * enorm);
}

// dense [n_mels, 1 + n_fft/2] row-major f32 filterbank (librosa.filters.mel, slaney norm)
inline std::vector<float> mel_filterbank(int sr, int n_fft, int n_mels) {
    const int bins = 1 + n_fft / 2;
    std::vector<float> w(static_cast<size_t>(n_mels) * bins);
    const std::vector<double> mel_f = mel_edges(n_mels, static_cast<double>(sr) / 2);
    for (int i = 0; i < n_mels; ++i)
        for (int k = 0; k < bins; ++k)
            w[static_cast<size_t>(i) * bins + k] = mel_weight(mel_f.data(), i, rfft_freq(k, n_fft, sr));
    return w;
}

// scipy.signal.get_window('hann', n, fftbins=True): general_cosine on n+1 points, last dropped
inline std::vector<double> hann_periodic(int n) {
    std::vector<double> w(n);
    for (int i = 0; i < n; ++i) {
        const double fac = linspace_at(-kPi, kPi, n + 1, i);
        w[i] = 0.5 + 0.5 * std::cos(fac);
    }
    return w;
}

// tuning value of histogram edge `idx` (np.linspace(-0.5, 0.5, 101)[idx])
inline double tuning_edge(int idx) { return linspace_at(-0.5, 0.5, kTunings, idx); }

// frqbins[k] of librosa.filters.chroma for k >= 1
inline double chroma_frqbin(int k, int n_fft, int sr, double tuning) {
    const double step = static_cast<double>(sr) / static_cast<double>(n_fft);  // np.linspace step
    const double freq = static_cast<double>(k) * step;
    const double a440 = 440.0 * std::pow(2.0, tuning / 12.0);
    return 12.0 * std::log2(freq / (a440 / 16));
}

// the 12 chroma weights of FFT bin k (one column of librosa.filters.chroma), f32
inline void chroma_column(int k, int n_fft, int sr, double tuning, float* out12) {
    const double f1 = chroma_frqbin(1, n_fft, sr, tuning);
    const double fk = (k == 0) ? f1 - 18.0 : chroma_frqbin(k, n_fft, sr, tuning);
    double width;
    if (k == n_fft - 1) {
        width = 1.0;
    } else {
        const double fn = chroma_frqbin(k + 1, n_fft, sr, tuning);
        width = fn - fk;
        if (!(width > 1.0)) width = 1.0;
    }
    double w[12];
    double ss = 0.0;
    for (int c = 0; c < 12; ++c) {
        double d = fk - static_cast<double>(c);
        // np.remainder(D + 6 + 120, 12) - 6   (python-style modulo, operand is positive)
        double r = std::fmod(d + 6.0 + 120.0, 12.0);
        if (r < 0) r += 12.0;
        d = r - 6.0;
        const double t = 2.0 * d / width;
        w[c] = std::exp(-0.5 * (t * t));
        ss += w[c] * w[c];
    }
    double length = std::pow(ss, 0.5);
    if (length < 2.2250738585072014e-308) length = 1.0;
    const double o = (fk / 12.0 - 5.0) / 2.0;
    const double oct = std::exp(-0.5 * (o * o));
    for (int c = 0; c < 12; ++c) {
        // np.roll(wts, -3, axis=0): row r of the result is row (r + 3) % 12
        const int src = (c + 3) % 12;
        out12[c] = static_cast<float>((w[src] / length) * oct);
    }
}

// bin-major [bins][12] f32 chroma filterbank for one tuning (the transpose of librosa's [12, bins])
inline void chroma_filterbank(int sr, int n_fft, double tuning, float* out) {
    const int bins = 1 + n_fft / 2;
    for (int k = 0; k < bins; ++k) chroma_column(k, n_fft, sr, tuning, out + static_cast<size_t>(k) * 12);
}

// scipy.fft.dct(type=2, norm='ortho') matrix rows 0..n_out-1 over n_in inputs, f32
inline std::vector<float> dct_ortho(int n_out, int n_in) {
    std::vector<float> m(static_cast<size_t>(n_out) * n_in);
    for (int k = 0; k < n_out; ++k) {
        const double s = (k == 0) ? std::sqrt(1.0 / n_in) : std::sqrt(2.0 / n_in);
        for (int n = 0; n < n_in; ++n)
            m[static_cast<size_t>(k) * n_in + n] =
                static_cast<float>(s * std::cos(kPi * (2.0 * n + 1.0) * k / (2.0 * n_in)));
    }
    return m;
}

// piptrack frequency mask (fmin=150 <= f < min(4000, sr/2)) as an inclusive bin range
inline void piptrack_bin_range(int sr, int n_fft, int* kmin, int* kmax) {
    const int bins = 1 + n_fft / 2;
    const double fmin = 150.0;
    double fmax = 4000.0;
    if (static_cast<double>(sr) / 2 < fmax) fmax = static_cast<double>(sr) / 2;
    int lo = bins, hi = -1;
    for (int k = 0; k < bins; ++k) {
        const double f = rfft_freq(k, n_fft, sr);
        if (fmin <= f && f < fmax) {
            if (k < lo) lo = k;
            if (k > hi) hi = k;
        }
    }
    *kmin = lo;
    *kmax = hi;
}

}  // namespace serhip
