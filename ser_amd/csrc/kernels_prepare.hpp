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
