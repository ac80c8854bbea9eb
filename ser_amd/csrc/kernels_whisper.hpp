// Whisper-style log-mel front end on gfx950 (config 5 of BASELINE.json):
//   whisper.audio.log_mel_spectrogram(audio, n_mels, padding=0)
// n_fft 400 / hop 160 / periodic Hann / reflect padding / last frame dropped /
// Slaney mel at 16 kHz / log10 / clamp to (clip max - 8) / (x + 4) / 4.
//
// 400 is not a power of two and may not be zero-padded (the bins would move), so each
// frame is a 200-point complex DFT of the even/odd packed samples, done as 5 x 5 x 8
// through LDS in double precision, then the real-FFT unpack to 201 bins.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_fast.hpp"

namespace serhip {

constexpr int kWNfft = 400;
constexpr int kWHop = 160;
constexpr int kWBins = 201;
constexpr int kWHalf = 200;          // packed complex points per frame
constexpr int kWWaves = 16;          // waves per workgroup (4 per SIMD; ~128 KB of LDS, one workgroup per CU)
constexpr int kWTile = 96;           // frames per workgroup
constexpr int kWMelWCap = 1024;      // sparse mel taps (80 mels: ~400, 128 mels: ~500)

typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2d cmuld(v2d a, v2d b) {
    return v2d{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

// forward 5-point DFT in place (W_5 = exp(-2 pi i / 5)), the symmetric form: 16 complex additions and
// 8 real-by-complex multiplications
__device__ __forceinline__ void dft5(v2d (&x)[5]) {
    const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;   // cos(2 pi/5), cos(4 pi/5)
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;    // sin(2 pi/5), sin(4 pi/5)
    const v2d a1 = x[1] + x[4], b1 = x[1] - x[4], a2 = x[2] + x[3], b2 = x[2] - x[3];
    const v2d m1 = x[0] + c1 * a1 + c2 * a2, m2 = x[0] + c2 * a1 + c1 * a2;
    const v2d n1 = s1 * b1 + s2 * b2, n2 = s2 * b1 - s1 * b2;
    x[0] = x[0] + a1 + a2;
    x[1] = v2d{m1.x + n1.y, m1.y - n1.x};   // m1 - i n1
    x[4] = v2d{m1.x - n1.y, m1.y + n1.x};   // m1 + i n1
    x[2] = v2d{m2.x + n2.y, m2.y - n2.x};
    x[3] = v2d{m2.x - n2.y, m2.y + n2.x};
}

struct WhisperTables {
    const v2d* tw200;    // [200] W_200^m   (double: the 1e-5 gate is against a float64 STFT)
    const v2d* tw400;    // [201] W_400^k
    const double* win;   // [400]
    const float* mel_w;  // sparse taps
    const int4* mel_rng; // [128] accumulator slots lane, 64 + lane: (first bin, taps, offset, mel or -1)
    int mel_nnz;
    int n_mels;
    int mel_pair;        // 1: the second slots of lanes 2j, 2j + 1 hold halves of one filter

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
    __shared__ v2d s_z[kWWaves][kWHalf];   // ONE buffer per wave: every stage below works in place (see the stages)
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

    const int4 rng0 = wt.mel_rng[lane];
    const int4 rng1 = wt.mel_rng[64 + lane];
    v2d* z = s_z[wave];
    v2d* y = z;   // stage 1a rewrites exactly the slots it read; 1b and 2 rely on a wave's LDS accesses completing in order
    float* p = s_p[wave];
    float vmax = -INFINITY;

    // Issue priority rises as a wave moves through its frame (0 framing, 1 first DFT stage, 2 second stage, 3 unpack /
    // mel / log) and falls back for the next one: of the four waves of a SIMD the one furthest along goes first, so a
    // frame's chain of six LDS-synchronised stages is not stretched by waves that have only just started theirs
    // (64 x 30 s clips: 0.372-0.386 -> 0.344-0.348 ms, round robin in one call; same idea as k_stft_frames).
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
        __builtin_amdgcn_s_setprio(1);
        // ---- step 1: Y[k1][n2] = sum_{n1<25} z[8 n1 + n2] W_25^(n1 k1), then * W_200^(n2 k1): eight 25-point
        // DFTs (one per n2), each as 5 x 5 Cooley-Tukey with n1 = 5 p + q, k1 = r + 5 s.  Lane (n2, q) takes
        // the 5-point DFT over p and the twiddle W_25^(q r); lane (n2, r) then the 5-point DFT over q.
        // 40 lanes, ~70 double-precision operations each per stage, against 25-term sums with a twiddle
        // recurrence (~800 per lane) before.  The second stage works in place in y: a wave's LDS
        // accesses complete in order, so every lane has read its inputs before any lane writes.
        {
            const int n2 = lane & 7, q5 = lane >> 3;   // q5: q in the first stage, r in the second
            if (lane < 40) {
                v2d x[5];
#pragma unroll
                for (int p5 = 0; p5 < 5; ++p5) x[p5] = z[8 * (5 * p5 + q5) + n2];
                dft5(x);
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const v2d a = r == 0 ? x[0] : cmuld(x[r], s_tw200[(8 * q5 * r) % kWHalf]);   // W_25^(q r)
                    y[(r * 5 + q5) * 8 + n2] = a;
                }
            }
            wave_lds_sync();
            if (lane < 40) {
                v2d x[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) x[q] = y[(q5 * 5 + q) * 8 + n2];
                dft5(x);
#pragma unroll
                for (int sgrp = 0; sgrp < 5; ++sgrp) {
                    const int k1 = q5 + 5 * sgrp;
                    y[k1 * 8 + n2] = cmuld(x[sgrp], s_tw200[n2 * k1]);   // n2 * k1 <= 168
                }
            }
        }
        wave_lds_sync();
        __builtin_amdgcn_s_setprio(2);
        // ---- step 2: Z[k1 + 25 k2] = sum_{n2<8} Y'[k1][n2] W_8^(n2 k2): lane k1 < 25 reads its eight values
        // (128 contiguous bytes) and takes a radix-2 x radix-4 FFT in registers (~60 operations; the
        // 8-term sums with constant roots took 112 on each of 50 lanes).  z was last read in step 1.
        if (lane < 25) {
            const double rh = 0.70710678118654752440;
            v2d yy[8];
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) yy[n2] = y[lane * 8 + n2];
            v2d e[4], o[4];
            {
                const v2d s0 = yy[0] + yy[4], d0 = yy[0] - yy[4], s1 = yy[2] + yy[6], t = yy[2] - yy[6];
                e[0] = s0 + s1;
                e[1] = v2d{d0.x + t.y, d0.y - t.x};   // d0 - i t
                e[2] = s0 - s1;
                e[3] = v2d{d0.x - t.y, d0.y + t.x};   // d0 + i t
            }
            {
                const v2d s0 = yy[1] + yy[5], d0 = yy[1] - yy[5], s1 = yy[3] + yy[7], t = yy[3] - yy[7];
                o[0] = s0 + s1;
                const v2d o1 = v2d{d0.x + t.y, d0.y - t.x};
                const v2d o2 = s0 - s1;
                const v2d o3 = v2d{d0.x - t.y, d0.y + t.x};
                o[1] = v2d{rh * (o1.x + o1.y), rh * (o1.y - o1.x)};     // * W_8^1 = (1 - i) / sqrt 2
                o[2] = v2d{o2.y, -o2.x};                                // * W_8^2 = -i
                o[3] = v2d{rh * (o3.y - o3.x), -rh * (o3.x + o3.y)};    // * W_8^3 = (-1 - i) / sqrt 2
            }
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                z[lane + 25 * k2] = e[k2] + o[k2];
                z[lane + 25 * (k2 + 4)] = e[k2] - o[k2];
            }
        }
        wave_lds_sync();
        __builtin_amdgcn_s_setprio(3);
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
            if (wt.mel_pair) a1 += __shfl_xor(a1, 1);   // lower half of the taps + upper half
            if (rng0.w >= 0) {
                const float l = log10f(fmaxf(a0, 1e-10f));
                s_tile[rng0.w][f] = l;
                vmax = fmaxf(vmax, l);
            }
            if (rng1.w >= 0 && !(wt.mel_pair && (lane & 1))) {
                const float l = log10f(fmaxf(a1, 1e-10f));
                s_tile[rng1.w][f] = l;
                vmax = fmaxf(vmax, l);
            }
        }
        wave_lds_sync();
        __builtin_amdgcn_s_setprio(0);
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
