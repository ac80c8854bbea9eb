// gfx950 kernels for the parts of the path that are not FFT-shaped:
//   k_dft_frames   segments shorter than 2048 samples, where the reference uses
//                  n_fft = N (any integer 512..2047, dsp.py:96) with hop N//4 for the
//                  explicit STFT and hop 512 inside mfcc/melspectrogram.  A direct
//                  double-precision DFT (the reference's FFT is double, librosa.stft)
//                  with the window, mel weights and piptrack computed on the fly.
//   k_pool_windows segment statistics pooling (pool/stats_pool.py:35-43,
//                  repr/handcrafted.py:118-122), one thread per (window, column),
//                  rows accumulated in numpy's order so results are bit-identical.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_fast.hpp"

namespace serhip {

// device twin of tables.hpp::chroma_column (same double-precision expression order)
__device__ void chroma_column_dev(int k, int n_fft, int sr, double tuning, float* out12) {
    const double step = static_cast<double>(sr) / static_cast<double>(n_fft);
    const double a440 = 440.0 * pow(2.0, tuning / 12.0);
    const double denom = a440 / 16;
    const double f1 = 12.0 * log2((1.0 * step) / denom);
    const double fk = (k == 0) ? f1 - 18.0 : 12.0 * log2((static_cast<double>(k) * step) / denom);
    double width;
    if (k == n_fft - 1) {
        width = 1.0;
    } else {
        const double fn = 12.0 * log2((static_cast<double>(k + 1) * step) / denom);
        width = fn - fk;
        if (!(width > 1.0)) width = 1.0;
    }
    double w[12];
    double ss = 0.0;
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        double d = fk - static_cast<double>(c);
        double r = fmod(d + 6.0 + 120.0, 12.0);
        if (r < 0) r += 12.0;
        d = r - 6.0;
        const double t = 2.0 * d / width;
        w[c] = exp(-0.5 * (t * t));
        ss += w[c] * w[c];
    }
    double length = sqrt(ss);
    if (length < 2.2250738585072014e-308) length = 1.0;
    const double o = (fk / 12.0 - 5.0) / 2.0;
    const double oct = exp(-0.5 * (o * o));
#pragma unroll
    for (int c = 0; c < 12; ++c) out12[c] = static_cast<float>((w[(c + 3) % 12] / length) * oct);
}

// device twin of tables.hpp::mel_weight
__device__ __forceinline__ float mel_weight_dev(const double* __restrict__ mel_f, int i, double freq) {
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

// One workgroup per (segment, job): jobs [0, n_stft) are explicit-STFT columns (hop n/4),
// jobs [n_stft, n_stft + n_mel) are mel-STFT columns (hop 512).
__global__ __launch_bounds__(256) void k_dft_frames(const float* __restrict__ audio,
                                                    const SegDesc* __restrict__ segs, PlanTables tb,
                                                    Scratch sc) {
    __shared__ double s_x[2048];
    __shared__ float s_mag[1028];
    __shared__ float s_red[4];
    __shared__ int s_cnt;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const SegDesc sd = segs[blockIdx.y];
    const int job = blockIdx.x;
    if (job >= sd.n_stft + sd.n_mel) return;
    const bool is_mel = job >= sd.n_stft;
    const int n = sd.n_fft;
    const int nbins = 1 + n / 2;
    const int t = is_mel ? job - sd.n_stft : job;
    const int hop = is_mel ? kHop : n / 4;
    const long long fs = static_cast<long long>(t) * hop - n / 2;
    const float* seg_audio = audio + sd.offset;

    // window (scipy get_window('hann', n, fftbins=True), float64) x frame
    const double wstep = (kPi - (-kPi)) / static_cast<double>(n);
    for (int j = tid; j < n; j += 256) {
        const double fac = static_cast<double>(j) * wstep + (-kPi);
        const double w = 0.5 + 0.5 * cos(fac);
        const long long s = fs + j;
        const float xv = (s >= 0 && s < sd.len) ? seg_audio[s] : 0.0f;
        s_x[j] = w * static_cast<double>(xv);
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();

    // X[k] = sum_j x_j z^j, z = exp(-i theta), theta = 2 pi k / n, by the Goertzel / Clenshaw recurrence
    //   b_j = x_j + 2 cos(theta) b_{j+1} - b_{j+2}   (j = n-1 .. 0),   X[k] = b_0 - b_1 conj(z)
    // in double: two arithmetic instructions per sample and bin and one broadcast LDS read per sample, where the
    // direct sum it replaces (a 32 KB twiddle table in LDS, read at a different address by every lane) took two
    // conflicting LDS reads per sample and bin and was bound by them: 63 -> 11 ms per 70 000 segments of 700 samples.
    // Rounding errors grow like n^2 eps ~ 5e-10 of the frame's scale at worst (theta -> 0), a hundredth of the
    // float32 rounding the reference applies to its double FFT (complex64) -- and four orders below the float32 FFT
    // of the fast path.  A thread owns up to four bins (tid + 256 c): independent chains that hide the latency
    // of the double-precision pipeline.
    float cmax = 0.0f;
    {
        double c2[4], b1[4] = {0.0, 0.0, 0.0, 0.0}, b2[4] = {0.0, 0.0, 0.0, 0.0}, cs[4], sn[4];
        const int chains = (nbins - tid + 255) / 256;   // bins tid, tid + 256, ... below nbins (<= 4: nbins <= 1024)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = tid + 256 * c;
            sincospi(2.0 * static_cast<double>(k) / static_cast<double>(n), &sn[c], &cs[c]);
            c2[c] = 2.0 * cs[c];
        }
        auto sweep = [&](auto nchains) {
            constexpr int kC = decltype(nchains)::value;
            for (int j = n - 1; j >= 1; --j) {   // leaves b1 = b_1, b2 = b_2
                const double xv = s_x[j];
#pragma unroll
                for (int c = 0; c < kC; ++c) {
                    const double b0 = fma(c2[c], b1[c], xv) - b2[c];
                    b2[c] = b1[c];
                    b1[c] = b0;
                }
            }
        };
        const int wave_chains = __builtin_amdgcn_readfirstlane((nbins - (tid & ~63) + 255) / 256);   // the wave's first lane has the most
        if (wave_chains >= 4) sweep(std::integral_constant<int, 4>{});
        else if (wave_chains == 3) sweep(std::integral_constant<int, 3>{});
        else if (wave_chains == 2) sweep(std::integral_constant<int, 2>{});
        else if (wave_chains == 1) sweep(std::integral_constant<int, 1>{});
        const double x0 = s_x[0];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < chains) {
                const int k = tid + 256 * c;
                const double b0 = fma(c2[c], b1[c], x0) - b2[c];
                const double re = fma(-b1[c], cs[c], b0);     // b_0 - b_1 cos(theta)
                const double im = -(b1[c] * sn[c]);           //     - i b_1 sin(theta)
                const float re32 = static_cast<float>(re), im32 = static_cast<float>(im);  // complex64 STFT
                const double m2 = static_cast<double>(re32) * re32 + static_cast<double>(im32) * im32;
                const float mag = static_cast<float>(sqrt(m2));
                s_mag[k] = mag;
                cmax = fmaxf(cmax, mag);
            }
        }
    }
    cmax = wave_max(cmax);
    if (lane == 0) s_red[wave] = cmax;
    __syncthreads();
    cmax = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));

    if (is_mel) {
        if (tid < kMels) {
            const double d = 1.0 / static_cast<double>(tb.sample_rate);
            const double val = 1.0 / (static_cast<double>(n) * d);
            float acc = 0.0f;
            // the triangle of filter `tid` is zero outside (mel_f[tid], mel_f[tid + 2]): walk only the bins around
            // it (one spare bin on each side; the weights are still evaluated exactly and zeros skipped as before,
            // so the sum runs over the same terms in the same order as the full walk, which cost 40x the DFT)
            const int k_lo = max(0, static_cast<int>(floor(tb.mel_f[tid] / val)) - 1);
            const int k_hi = min(nbins - 1, static_cast<int>(ceil(tb.mel_f[tid + 2] / val)) + 1);
            for (int k = k_lo; k <= k_hi; ++k) {
                const float w = mel_weight_dev(tb.mel_f, tid, static_cast<double>(k) * val);
                if (w != 0.0f) {
                    const float m = s_mag[k];
                    acc = fmaf(w, m * m, acc);
                }
            }
            sc.mel[(static_cast<size_t>(sd.mel_base) + t) * kMels + tid] = acc;
        }
        return;
    }

    const size_t row = static_cast<size_t>(sd.stft_base) + t;
    float* mrow = sc.mags + row * kBinStride;
    for (int k = tid; k < nbins; k += 256) mrow[k] = s_mag[k];

    // piptrack: estimate_tuning does not forward n_fft, so piptrack re-infers it from the
    // row count: 2 * (nbins - 1)
    const int n_pip = 2 * (nbins - 1);
    const double d = 1.0 / static_cast<double>(tb.sample_rate);
    const double val = 1.0 / (static_cast<double>(n_pip) * d);
    double fmax = 4000.0;
    if (static_cast<double>(tb.sample_rate) / 2 < fmax) fmax = static_cast<double>(tb.sample_rate) / 2;
    const float ref = 0.1f * cmax;
    v2f* prow = sc.peaks + row * sc.peak_cap;
    for (int k = tid; k < nbins - 1; k += 256) {
        if (k < 1) continue;
        const double f = static_cast<double>(k) * val;
        if (!(150.0 <= f && f < fmax)) continue;
        const float sm = s_mag[k - 1], s0 = s_mag[k], sp = s_mag[k + 1];
        const float tm = sm > ref ? sm : 0.0f;
        const float t0 = s0 > ref ? s0 : 0.0f;
        const float tp = sp > ref ? sp : 0.0f;
        if ((t0 > tm) && (t0 >= tp)) {
            const int pos = atomicAdd(&s_cnt, 1);
            prow[pos] = piptrack_peak<false>(sm, s0, sp, k, static_cast<double>(tb.sample_rate),
                                      static_cast<double>(n_pip));
        }
    }
    __syncthreads();
    if (tid == 0) sc.peak_cnt[row] = s_cnt;
}

// correctly rounded double sqrt: the device sqrt is within 1 ulp; one Markstein correction step
// (exact residual by fma, then a fused update) rounds it the way IEEE-754 / numpy does
__device__ __forceinline__ double sqrt_rn(double x) {
    double y = sqrt(x);
    if (x > 0.0 && y < 1.0e300) {
        const double r = fma(-y, y, x);
        const double h = 0.5 / y;
        y = fma(r, h, y);
    }
    return y;
}

// out[w, c]        = mean over rows [lo, hi) of emb[:, c]                     (both modes)
// out[w, dim + c]  = population std (mode 0 only), numpy's two-pass formula in float64
__global__ __launch_bounds__(256) void k_pool_windows(const float* __restrict__ emb, int dim,
                                                      const int* __restrict__ win_lo,
                                                      const int* __restrict__ win_hi, int n_win, int mode,
                                                      double* __restrict__ out) {
#pragma clang fp contract(off)  // numpy rounds every product and sum separately
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int w = blockIdx.y;
    if (c >= dim || w >= n_win) return;
    const int lo = win_lo[w], hi = win_hi[w];
    const int n = hi - lo;
    if (mode == 1) {
        float s = 0.0f;  // np.add.reduce over axis 0 in float32, row by row
        for (int r = lo; r < hi; ++r) s += emb[static_cast<size_t>(r) * dim + c];
        out[static_cast<size_t>(w) * dim + c] = static_cast<double>(s / static_cast<float>(n));
        return;
    }
    double s = 0.0;
    for (int r = lo; r < hi; ++r) s += static_cast<double>(emb[static_cast<size_t>(r) * dim + c]);
    const double mean = s / static_cast<double>(n);
    double v = 0.0;
    for (int r = lo; r < hi; ++r) {
        const double d = static_cast<double>(emb[static_cast<size_t>(r) * dim + c]) - mean;
        v = __dadd_rn(v, __dmul_rn(d, d));  // no fma contraction: numpy squares, rounds, then adds
    }
    double* orow = out + static_cast<size_t>(w) * 2 * dim;
    orow[c] = mean;
    orow[dim + c] = sqrt_rn(v / static_cast<double>(n));
}

}  // namespace serhip
