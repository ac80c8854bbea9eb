// k_project_frames: the two filterbank projections of the fast path on the matrix cores.
//
//   chroma_raw[t, c] = sum_k fb_tuning[c, k] * |X|[k, t]        (librosa.feature.chroma_stft, dsp.py:115)
//   mel[t, m]        = sum_k W[m, k]         * |X|^2[k, t]      (melspectrogram inside mfcc/mel, dsp.py:108,122)
//
// One wave owns a tile of 32 frames of one segment and walks the 1025 bins in groups of 16 with
// v_mfma_f32_16x16x4_f32 (exact f32 fma chain): A = 16 frames x 4 bins straight from the magnitude
// rows in HBM (each lane one 16-byte load per group: the contraction index may be permuted freely,
// so lane (frame i, kk) takes bins 16G+4kk..+3 and uses them as 4 consecutive k-steps), B = the
// filterbank pre-arranged on the host in the same order.  The mel filterbank is banded, so a
// 16-filter block only visits the groups its filters touch (~1.3 blocks per group).
// Epilogue: per-frame inf-norm of the chroma row (a 16-lane DPP reduction) and stores of
// chroma_norm[t, 16] and mel[t, 128]; the time means are taken by k_segment_finish<2>.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_fast.hpp"

namespace serhip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kProjTile = 32;   // frames per wave

__device__ __forceinline__ f32x4 mfma4(float4 a, float4 b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    return c;
}

// Mel blocks (16 filters each) are live over a contiguous run of groups [glo, ghi]; the runs advance
// monotonically and at most two are live at once (serhip_plan_fast checks this), so the kernel keeps
// two accumulator slots and retires the older one -- storing its 32 x 16 tile -- when its run ends.
__global__ __launch_bounds__(256) void k_project_frames(const SegDesc* __restrict__ segs, PlanTables tb, Scratch sc) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const SegDesc sd = segs[blockIdx.y];
    const int t0 = (blockIdx.x * 4 + wave) * kProjTile;
    if (t0 >= sd.n_stft) return;   // no workgroup barrier in this kernel: waves leave independently
    const bool want_chroma = (tb.flags & 2u) != 0, want_mel = (tb.flags & 5u) != 0;
    const int i = lane & 15, kk = lane >> 4;
    const int last = sd.n_stft - 1;
    const float* a0p = sc.mags + (static_cast<size_t>(sd.stft_base) + min(t0 + i, last)) * kBinStride + 4 * kk;
    const float* a1p = sc.mags + (static_cast<size_t>(sd.stft_base) + min(t0 + 16 + i, last)) * kBinStride + 4 * kk;
    const float4* cb = tb.chroma_b + (want_chroma ? static_cast<size_t>(sc.tuning_idx[blockIdx.y]) * kGroups * 64 : 0) + lane;
    const float4* mb = tb.mel_b + lane;
    float* mel_out = sc.mel + (static_cast<size_t>(sd.mel_base) + t0 + 4 * kk) * kMels + i;

    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 c0 = zero, c1 = zero;
    f32x4 lo0 = zero, lo1 = zero, hi0 = zero, hi1 = zero;   // mel blocks cur and cur + 1
    int cur = 0;
    int4 blk_lo = tb.mel_blk[0], blk_hi = tb.mel_blk[1];     // {glo, ghi, table offset, 0}

    // store the retiring block's tile: C layout = column j = lane & 15 (filter), rows 4*kk + r (frames)
    auto retire = [&](int b, const f32x4& v0, const f32x4& v1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (t0 + 4 * kk + r < sd.n_stft) mel_out[static_cast<size_t>(r) * kMels + 16 * b] = v0[r];
            if (t0 + 16 + 4 * kk + r < sd.n_stft) mel_out[static_cast<size_t>(16 + r) * kMels + 16 * b] = v1[r];
        }
    };

#pragma unroll 1
    for (int G = 0; G < kGroups; ++G) {
        // group 64 holds only bin 1024 (+ three zero pads): lanes kk > 0 re-read the kk = 0 chunk and
        // zero it, so no load leaves the row
        const bool tail = (G == kGroups - 1);
        const int aoff = 16 * G - (tail ? 4 * kk : 0);
        float4 a0 = *reinterpret_cast<const float4*>(a0p + aoff);
        float4 a1 = *reinterpret_cast<const float4*>(a1p + aoff);
        if (tail && kk != 0) a0 = a1 = float4{0.f, 0.f, 0.f, 0.f};
        if (want_chroma) {
            const float4 b = cb[G * 64];
            c0 = mfma4(a0, b, c0);
            c1 = mfma4(a1, b, c1);
        }
        if (want_mel) {
            while (cur < 8 && G > blk_lo.y) {   // wave-uniform
                retire(cur, lo0, lo1);
                lo0 = hi0;
                lo1 = hi1;
                hi0 = hi1 = zero;
                ++cur;
                blk_lo = blk_hi;
                blk_hi = tb.mel_blk[min(cur + 1, 8)];   // entry 8 is a sentinel that never starts
            }
            const float4 p0 = float4{a0.x * a0.x, a0.y * a0.y, a0.z * a0.z, a0.w * a0.w};   // |X|^2 of the rounded |X|
            const float4 p1 = float4{a1.x * a1.x, a1.y * a1.y, a1.z * a1.z, a1.w * a1.w};
            if (cur < 8 && G >= blk_lo.x) {
                const float4 w = mb[static_cast<size_t>(blk_lo.z + G - blk_lo.x) * 64];
                lo0 = mfma4(p0, w, lo0);
                lo1 = mfma4(p1, w, lo1);
            }
            if (cur < 7 && G >= blk_hi.x) {
                const float4 w = mb[static_cast<size_t>(blk_hi.z + G - blk_hi.x) * 64];
                hi0 = mfma4(p0, w, hi0);
                hi1 = mfma4(p1, w, hi1);
            }
        }
    }
    if (want_mel) {
        while (cur < 8) {
            retire(cur, lo0, lo1);
            lo0 = hi0;
            lo1 = hi1;
            hi0 = hi1 = zero;
            ++cur;
        }
    }

    // ---- chroma epilogue: per-frame inf-norm over the 16 lanes that hold one frame's classes
    if (want_chroma) {
        const float tiny = 1.17549435e-38f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const f32x4 c = half ? c1 : c0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + 16 * half + 4 * kk + r;
                const float raw = c[r];
                float mx = fabsf(raw);
#pragma unroll
                for (int o = 8; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
                if (mx < tiny) mx = 1.0f;
                if (t < sd.n_stft) sc.chroma_norm[(static_cast<size_t>(sd.stft_base) + t) * 16 + i] = raw / mx;
            }
        }
    }
}

}  // namespace serhip
