// Device-side audio preparation: the GPU form of ser/_internal/utils/audio_utils.py:28-60
// (_prepare_audio_buffer -> nan_to_num, _to_mono, _normalize_audio) fused with the sample-format
// conversion that libsndfile performs for the reference's librosa.load (PCM16 -> x / 32768).
//
//   k_prepare_convert    raw interleaved samples -> float32, channel mean in float32 with numpy's
//                        summation order, non-finite -> 0, mono samples written out, per-clip max |x|
//                        gathered with an integer atomicMax on the float bit pattern.  The mean is taken
//                        on the RAW samples and the scrub applied to the mono signal: that is the order of
//                        the reference's primary path (librosa.load mixes down with mono=True before
//                        _prepare_audio_buffer's nan_to_num sees the data, audio_utils.py:104-110), so a
//                        NaN or inf in one channel zeroes that frame.
//   k_prepare_normalise  x / peak (IEEE float32 division, as numpy divides the array by the scalar);
//                        an all-zero clip stays zero
// Both are plain HBM streams: 2 B (PCM16) or 4 B per input sample read, 4 B written, then 4 + 4.
//
//   k_prepare_clip       the two steps in ONE launch for clips of up to kPrepFusedMaxFrames frames: one workgroup
//                        per clip walks the clip twice -- first for the peak, then converting again (the raw bytes
//                        of a clip, 96-192 KB for 3 s, are still in L2) and storing x / peak once.  HBM traffic is
//                        the algorithmic minimum (raw bytes in, float32 out) instead of raw + 3 x float32, and
//                        aligned mono / stereo PCM16 and mono float32 move four frames per lane and instruction.
//                        Same arithmetic, same bits as the pair above.  Long recordings keep the pair (a single
//                        workgroup would stream a whole hour alone).
#pragma once

#include <hip/hip_runtime.h>

namespace serhip {

struct PrepClip {
    long long raw_offset;   // byte offset of the clip's first frame in the raw buffer
    long long out_offset;   // first output sample
    long long frames;       // sample frames
    int channels;
    int pad;
};

template <int kFormat>   // 0: little-endian PCM16, 1: IEEE float32
__device__ __forceinline__ float prep_sample(const unsigned char* __restrict__ raw, long long idx) {
    if (kFormat == 0) {
        const short v = reinterpret_cast<const short*>(raw)[idx];
        return static_cast<float>(v) * (1.0f / 32768.0f);   // exact: power-of-two scale of a 16-bit integer
    }
    return reinterpret_cast<const float*>(raw)[idx];
}
// NaN and +-inf -> 0 (np.nan_to_num with nan = posinf = neginf = 0)
__device__ __forceinline__ float prep_scrub(float x) { return (fabsf(x) <= 3.4028234663852886e38f) ? x : 0.0f; }

// np.add.reduce over a contiguous float32 run of n < 128 elements (numpy's pairwise_sum)
template <int kFormat>
__device__ __forceinline__ float prep_channel_sum(const unsigned char* __restrict__ raw, long long first, int n) {
#pragma clang fp contract(off)
    if (n < 8) {
        float res = 0.0f;
        for (int c = 0; c < n; ++c) res += prep_sample<kFormat>(raw, first + c);
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = prep_sample<kFormat>(raw, first + j);
    int c = 8;
    for (; c < n - (n % 8); c += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += prep_sample<kFormat>(raw, first + c + j);
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; c < n; ++c) res += prep_sample<kFormat>(raw, first + c);
    return res;
}

template <int kFormat>
__global__ __launch_bounds__(256) void k_prepare_convert(const unsigned char* __restrict__ raw,
                                                         const PrepClip* __restrict__ clips,
                                                         float* __restrict__ out, unsigned* __restrict__ clip_peak) {
    const PrepClip c = clips[blockIdx.y];
    const unsigned char* base = raw + c.raw_offset;
    float* dst = out + c.out_offset;
    float peak = 0.0f;
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long f = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; f < c.frames; f += stride) {
        float x;
        if (c.channels == 1) {
            x = prep_scrub(prep_sample<kFormat>(base, f));
        } else {
            x = prep_scrub(prep_channel_sum<kFormat>(base, f * c.channels, c.channels) / static_cast<float>(c.channels));
        }
        dst[f] = x;
        peak = fmaxf(peak, fabsf(x));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) peak = fmaxf(peak, __shfl_xor(peak, o));
    // non-negative floats order like their bit patterns
    if ((threadIdx.x & 63) == 0 && peak > 0.0f) atomicMax(&clip_peak[blockIdx.y], __float_as_uint(peak));
}

constexpr long long kPrepFusedMaxFrames = 1ll << 18;   // 16 s at 16 kHz

// four consecutive frames of a clip whose base address is 4-byte aligned: mono / stereo PCM16 or mono float32
// (one 8- or 16-byte load); the mono value of each frame exactly as the scalar path computes it
template <int kFormat>
__device__ __forceinline__ void prep_quad(const unsigned char* __restrict__ base, long long q, int channels, float (&x)[4]) {
#pragma clang fp contract(off)
    if (kFormat == 0) {
        const float k = 1.0f / 32768.0f;
        if (channels == 1) {
            const uint2 v = *reinterpret_cast<const uint2*>(base + 8 * q);
            const unsigned w[2] = {v.x, v.y};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                x[j] = prep_scrub(static_cast<float>(static_cast<short>(w[j >> 1] >> (16 * (j & 1)))) * k);
        } else {
            const uint4 v = *reinterpret_cast<const uint4*>(base + 16 * q);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = static_cast<float>(static_cast<short>(w[j] & 0xffffu)) * k;
                const float b = static_cast<float>(static_cast<short>(w[j] >> 16)) * k;
                float res = 0.0f;   // numpy's running sum of a run shorter than 8
                res += a;
                res += b;
                x[j] = prep_scrub(res / 2.0f);
            }
        }
    } else {
        const float4 v = *reinterpret_cast<const float4*>(base + 16 * q);
        x[0] = prep_scrub(v.x);
        x[1] = prep_scrub(v.y);
        x[2] = prep_scrub(v.z);
        x[3] = prep_scrub(v.w);
    }
}

template <int kFormat>
__global__ __launch_bounds__(256) void k_prepare_clip(const unsigned char* __restrict__ raw, const PrepClip* __restrict__ clips,
                                                      float* __restrict__ out) {
    __shared__ float s_peak[4];
    const PrepClip c = clips[blockIdx.x];
    const unsigned char* base = raw + c.raw_offset;
    float* dst = out + c.out_offset;
    const int tid = threadIdx.x;
    const bool quads = (kFormat == 0 ? c.channels <= 2 : c.channels == 1) && (reinterpret_cast<unsigned long long>(base) & 3ull) == 0;
    const long long n_quads = quads ? c.frames / 4 : 0;
    auto scalar = [&](long long f) {
        if (c.channels == 1) return prep_scrub(prep_sample<kFormat>(base, f));
        return prep_scrub(prep_channel_sum<kFormat>(base, f * c.channels, c.channels) / static_cast<float>(c.channels));
    };
    // ---- pass 1: the clip's largest |x|
    float peak = 0.0f;
#pragma unroll 4
    for (long long q = tid; q < n_quads; q += 256) {
        float x[4];
        prep_quad<kFormat>(base, q, c.channels, x);
        peak = fmaxf(fmaxf(peak, fmaxf(fabsf(x[0]), fabsf(x[1]))), fmaxf(fabsf(x[2]), fabsf(x[3])));
    }
    for (long long f = 4 * n_quads + tid; f < c.frames; f += 256) peak = fmaxf(peak, fabsf(scalar(f)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) peak = fmaxf(peak, __shfl_xor(peak, o));
    if ((tid & 63) == 0) s_peak[tid >> 6] = peak;
    __syncthreads();
    peak = fmaxf(fmaxf(s_peak[0], s_peak[1]), fmaxf(s_peak[2], s_peak[3]));
    const bool scale = peak > 0.0f;   // an all-zero clip keeps its (zero) samples, like k_prepare_normalise's early return
    // ---- pass 2: convert again (the raw bytes come from L2), divide, store once
#pragma unroll 4
    for (long long q = tid; q < n_quads; q += 256) {
        float x[4];
        prep_quad<kFormat>(base, q, c.channels, x);
        float* d = dst + 4 * q;
        if (scale) {
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = x[j] / peak;
        }
        if ((reinterpret_cast<unsigned long long>(d) & 15ull) == 0) {
            *reinterpret_cast<float4*>(d) = float4{x[0], x[1], x[2], x[3]};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = x[j];
        }
    }
    for (long long f = 4 * n_quads + tid; f < c.frames; f += 256) {
        const float x = scalar(f);
        dst[f] = scale ? x / peak : x;
    }
}

__global__ __launch_bounds__(256) void k_prepare_normalise(const PrepClip* __restrict__ clips, float* __restrict__ out,
                                                           const unsigned* __restrict__ clip_peak) {
    const PrepClip c = clips[blockIdx.y];
    const float peak = __uint_as_float(clip_peak[blockIdx.y]);
    if (!(peak > 0.0f)) return;   // all-zero clip: np.zeros_like
    float* dst = out + c.out_offset;
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long f = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; f < c.frames; f += stride)
        dst[f] = dst[f] / peak;
}

}  // namespace serhip
