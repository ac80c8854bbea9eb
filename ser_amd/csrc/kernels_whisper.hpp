// Whisper-style log-mel front end on gfx950 (config 5 of BASELINE.json):
//   whisper.audio.log_mel_spectrogram(audio, n_mels, padding=0)
// n_fft 400 / hop 160 / periodic Hann / reflect padding / last frame dropped /
// Slaney mel at 16 kHz / log10 / clamp to (clip max - 8) / (x + 4) / 4.
//
// 400 is not a power of two and may not be zero-padded (the bins would move), so each
// frame is a 200-point complex DFT of the even/odd packed samples, done as a 25 x 8
// two-step decomposition through LDS (sums of 25 and 8 terms, so float32 error stays
// at FFT level), then the real-FFT unpack to 201 bins.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_fast.hpp"

namespace serhip {

constexpr int kWNfft = 400;
constexpr int kWHop = 160;
constexpr int kWBins = 201;
constexpr int kWHalf = 200;          // packed complex points per frame
constexpr int kWWaves = 12;          // waves per workgroup (3 per SIMD; ~150 KB of LDS, one workgroup per CU)
constexpr int kWTile = 96;           // frames per workgroup
constexpr int kWMelWCap = 1024;      // sparse mel taps (80 mels: ~400, 128 mels: ~500)

typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2d cmuld(v2d a, v2d b) {
    return v2d{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

struct WhisperTables {
    const v2d* tw200;    // [200] W_200^m   (double: the 1e-5 gate is against a float64 STFT)
    const v2d* tw400;    // [201] W_400^k
    const double* win;   // [400]
    const float* mel_w;  // sparse taps
    const int4* mel_rng; // [n_mels] (first bin, count, offset, 0)
    int mel_nnz;
    int n_mels;
};

// pass 1: log10(max(mel, 1e-10)) into out[clip][m][t] and the running clip maximum.
// The 400-point DFT is accumulated in double: bins 80 dB below the clip maximum still have to
// agree with a float64 STFT to ~1e-4 relative, which a float32 transform does not deliver.
__global__ __launch_bounds__(kWWaves * 64) void k_whisper_frames(const float* __restrict__ audio, long long n_samples,
                                                        int n_frames, WhisperTables wt, float* __restrict__ out,
                                                        unsigned* __restrict__ clip_max) {
    __shared__ v2d s_tw200[kWHalf];
    __shared__ v2d s_tw400[kWBins + 1];
    __shared__ double s_win[kWNfft];
    __shared__ float s_melw[kWMelWCap];
    __shared__ v2d s_z[kWWaves][kWHalf];
    __shared__ v2d s_y[kWWaves][kWHalf];
    __shared__ float s_p[kWWaves][kWBins + 3];
    __shared__ float s_tile[128][kWTile + 1];
    __shared__ float s_red[kWWaves];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int clip = blockIdx.y;
    const int t_tile = blockIdx.x * kWTile;
    const float* x = audio + static_cast<size_t>(clip) * n_samples;

    for (int i = tid; i < kWHalf; i += kWWaves * 64) s_tw200[i] = wt.tw200[i];
    for (int i = tid; i < kWBins; i += kWWaves * 64) s_tw400[i] = wt.tw400[i];
    for (int i = tid; i < kWNfft; i += kWWaves * 64) s_win[i] = wt.win[i];
    for (int i = tid; i < wt.mel_nnz; i += kWWaves * 64) s_melw[i] = wt.mel_w[i];
    __syncthreads();

    const int4 rng0 = (lane < wt.n_mels) ? wt.mel_rng[lane] : int4{0, 0, 0, 0};
    const int4 rng1 = (lane + 64 < wt.n_mels) ? wt.mel_rng[lane + 64] : int4{0, 0, 0, 0};
    v2d* z = s_z[wave];
    v2d* y = s_y[wave];
    float* p = s_p[wave];
    float vmax = -INFINITY;

    for (int f = wave; f < kWTile; f += kWWaves) {
        const int t = t_tile + f;
        if (t >= n_frames) break;
        // ---- framing with reflect padding (200 samples each side) + Hann
        const long long fs = static_cast<long long>(t) * kWHop - kWNfft / 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = lane + 64 * i;
            if (n < kWHalf) {
                long long s0 = fs + 2 * n, s1 = s0 + 1;
                if (s0 < 0) s0 = -s0;
                if (s0 >= n_samples) s0 = 2 * (n_samples - 1) - s0;
                if (s1 < 0) s1 = -s1;
                if (s1 >= n_samples) s1 = 2 * (n_samples - 1) - s1;
                z[n] = v2d{static_cast<double>(x[s0]) * s_win[2 * n], static_cast<double>(x[s1]) * s_win[2 * n + 1]};
            }
        }
        wave_lds_sync();
        // ---- step 1: Y[k1][n2] = sum_{n1<25} z[8 n1 + n2] W_25^(n1 k1), then * W_200^(n2 k1).
        // Output o = lane + 64 i has n2 = lane & 7 for every i, so the four outputs of a lane share each
        // z read; their twiddles W_25^(n1 k1) advance by a complex multiplication (double precision:
        // 24 steps cost ~1e-15) instead of a table read.  LDS reads per lane: 25 + 8 instead of 200.
        {
            const int n2 = lane & 7;
            v2d acc[4], w[4], step[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k1 = min((lane >> 3) + 8 * i, 24);
                acc[i] = v2d{0.0, 0.0};
                w[i] = v2d{1.0, 0.0};
                step[i] = s_tw200[(8 * k1) % kWHalf];   // W_25^k1
            }
            for (int n1 = 0; n1 < 25; ++n1) {
                const v2d zz = z[8 * n1 + n2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] += cmuld(zz, w[i]);
                    w[i] = cmuld(w[i], step[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = lane + 64 * i;
                if (o < kWHalf) y[o] = cmuld(acc[i], s_tw200[n2 * (o >> 3)]);
            }
        }
        wave_lds_sync();
        // ---- step 2: Z[k1 + 25 k2] = sum_{n2<8} Y'[k1][n2] W_8^(n2 k2).  Lane (k1, half) reads the 8
        // values of its k1 once and produces k2 = 4 half + j, j = 0..3: W_8^(n2 k2) = W_8^(n2 j) (-1)^(n2 half),
        // compile-time eighth roots of unity after the odd terms are negated for half = 1.
        v2d zk[4];
        const int k1s = lane % 25, half = lane / 25;
        if (lane < 50) {
            const double r = 0.70710678118654752440;
            const v2d w8[8] = {v2d{1.0, 0.0}, v2d{r, -r}, v2d{0.0, -1.0}, v2d{-r, -r},
                               v2d{-1.0, 0.0}, v2d{-r, r}, v2d{0.0, 1.0}, v2d{r, r}};
            v2d yy[8];
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) {
                yy[n2] = y[k1s * 8 + n2];
                if ((n2 & 1) && half) yy[n2] = -yy[n2];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v2d acc = yy[0];
#pragma unroll
                for (int n2 = 1; n2 < 8; ++n2) acc += cmuld(yy[n2], w8[(n2 * j) & 7]);
                zk[j] = acc;
            }
        }
        if (lane < 50) {   // z was last read in step 1, two barriers ago
#pragma unroll
            for (int j = 0; j < 4; ++j) z[k1s + 25 * (4 * half + j)] = zk[j];
        }
        wave_lds_sync();
        // ---- real-FFT unpack to bins 0..200 and |X|^2
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane + 64 * i;
            if (k < kWBins) {
                const v2d za = z[k % kWHalf];
                const v2d zb = z[(kWHalf - k) % kWHalf];
                const v2d e2 = v2d{za.x + zb.x, za.y - zb.y};
                const v2d o2 = v2d{za.y + zb.y, zb.x - za.x};
                const v2d xk = 0.5 * (e2 + cmuld(o2, s_tw400[k]));
                p[k] = static_cast<float>(xk.x * xk.x + xk.y * xk.y);
            }
        }
        wave_lds_sync();
        // ---- sparse mel, log10
        {
            float a0 = 0.0f, a1 = 0.0f;
            const int nmax = max(rng0.y, rng1.y);
            for (int j = 0; j < nmax; ++j) {
                if (j < rng0.y) a0 = fmaf(s_melw[rng0.z + j], p[rng0.x + j], a0);
                if (j < rng1.y) a1 = fmaf(s_melw[rng1.z + j], p[rng1.x + j], a1);
            }
            if (lane < wt.n_mels) {
                const float l = log10f(fmaxf(a0, 1e-10f));
                s_tile[lane][f] = l;
                vmax = fmaxf(vmax, l);
            }
            if (lane + 64 < wt.n_mels) {
                const float l = log10f(fmaxf(a1, 1e-10f));
                s_tile[lane + 64][f] = l;
                vmax = fmaxf(vmax, l);
            }
        }
        wave_lds_sync();
    }
    vmax = wave_max(vmax);
    if (lane == 0) s_red[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
        float m = s_red[0];
        for (int w = 1; w < kWWaves; ++w) m = fmaxf(m, s_red[w]);
        if (m > -INFINITY) atomicMax(&clip_max[clip], float_key(m));
    }
    // ---- coalesced store of the [n_mels][tile] block (frames are the minor axis)
    const int valid = min(kWTile, n_frames - t_tile);
    float* o = out + static_cast<size_t>(clip) * wt.n_mels * n_frames + t_tile;
    for (int i = tid; i < wt.n_mels * kWTile; i += kWWaves * 64) {
        const int m = i / kWTile, f = i - m * kWTile;
        if (f < valid) o[static_cast<size_t>(m) * n_frames + f] = s_tile[m][f];
    }
}

// pass 2: out = (max(out, clip_max - 8) + 4) / 4
__global__ __launch_bounds__(256) void k_whisper_finalize(float* __restrict__ out, long long per_clip,
                                                          const unsigned* __restrict__ clip_max) {
    const int clip = blockIdx.y;
    const float floor_v = key_float(clip_max[clip]) - 8.0f;
    float* o = out + static_cast<size_t>(clip) * per_clip;
    for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < per_clip;
         i += static_cast<long long>(gridDim.x) * 256) {
        const float v = fmaxf(o[i], floor_v);
        o[i] = (v + 4.0f) / 4.0f;
    }
}

}  // namespace serhip
