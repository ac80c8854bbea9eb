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
// Epilogue: per-frame inf-norm of the chroma row (a 16-lane reduction) -> chroma_norm[t, 16]; per
// retired mel block and 16-frame tile the power sums, dB sums, dB minima and the dB maximum, and the
// dB rows mel[t, 128] only where top_db might clip them; k_segment_finish<2> takes the time means.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_fast.hpp"

namespace serhip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// 16-frame MFMA tiles per wave: 2 = 32 frames per wave, three waves per workgroup.  The code is written for
// 1..3 (all parity-green; measured 1.52 / 1.29 / 1.29 ms per 12 500 clips: one tile per wave pays the per-group
// costs -- B reads, barrier, program word -- per 16 frames, three tiles leave a third fewer waves); this is a
// design constant, not a build switch.  Either way a workgroup covers 96 frames = one 3 s clip and fetches each
// B row once.
constexpr int kProjHalves = 2;
static_assert(kProjHalves >= 1 && kProjHalves <= 3, "one to three tiles per wave");
constexpr int kProjTile = 16 * kProjHalves;   // frames per wave
constexpr int kProjWaves = 6 / kProjHalves;   // waves per workgroup
constexpr int kProjBLoads = (3 + kProjWaves - 1) / kProjWaves;   // B rows a wave fetches per group: 1, or 2 with two waves

// v[l] (+|min) v[l ^ 16] and v[l ^ 32] in every lane without an LDS round trip: v_permlane16_swap /
// v_permlane32_swap (gfx950) of two copies of v leave {rows 0,0,2,2 | rows 1,1,3,3} and {lower half twice | upper
// half twice}.  The operands are the same two numbers a __shfl_xor butterfly would combine, so the sums are
// bit-identical to it.
__device__ __forceinline__ float sum_xor16(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_xor32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float min_xor16(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float min_xor32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
template <bool k32>
__device__ __forceinline__ double sum_xor_rows(double v) {
    const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v));
    const unsigned lo = static_cast<unsigned>(b), hi = static_cast<unsigned>(b >> 32);
    const auto rl = k32 ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false) : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = k32 ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false) : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double a = __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(rh[0]) << 32) | rl[0]));
    const double c = __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(rh[1]) << 32) | rl[1]));
    return a + c;
}
// max over each row of 16 lanes, in every lane of the row (the first four steps of wave_max)
__device__ __forceinline__ float row_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_mov<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_mov<0x141>(v));   // row_half_mirror
    return fmaxf(v, dpp_mov<0x140>(v));   // row_mirror
}

// magnitude fragment load (default cache policy: streaming `nt` loads measured +5 %, the second half of each
// line is then fetched again)
__device__ __forceinline__ float4 load_mags(const float4* p) { return *p; }

__device__ __forceinline__ f32x4 mfma4(float4 a, float4 b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    return c;
}

// Mel blocks (16 filters each) are live over a contiguous run of groups [glo, ghi]; the runs advance
// monotonically and at most two are live at once (serhip_plan_fast checks this), so the kernel keeps
// two accumulator slots ("lo" = oldest live block, "hi" = the next) and retires the older one --
// storing its 32 x 16 tile -- when its run ends.  The host compiles this into a per-group program
// (PlanTables::mel_prog): which B rows the two slots use (row 0 is all zeros), which slots are live,
// and how many blocks retire after the group.  All operands of group G + 1 are requested before the
// MFMAs of group G are issued.
// The per-group program travels as a kernel argument (1 KB by value): indexed by the wave-uniform group number it is
// read with scalar loads from the kernarg segment -- no LDS copy, no v_readfirstlane, and nothing on the vector
// memory counter that the software pipeline's waits count (the first form kept a copy in LDS: +1.8 %).
struct ProjProgram {
    int4 g[kGroups];   // {row of slot lo, row of slot hi, live mask (1 lo, 2 hi), retires}
};
__global__ __launch_bounds__(kProjWaves * 64) void k_project_frames(const SegDesc* __restrict__ segs, PlanTables tb, Scratch sc,
                                                                    ProjProgram prog) {
    // B rows, shared by the workgroup's waves (all work on one segment, so on one tuning).  A stage holds the rows of
    // a PAIR of groups: [stage][group parity][0 chroma | 1 mel slot lo | 2 mel slot hi][lane]
    __shared__ float4 s_b[2][2][3][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SegDesc sd = segs[blockIdx.y];
    if (blockIdx.x * kProjWaves * kProjTile >= sd.n_stft) return;   // whole workgroup
    const int t0 = (blockIdx.x * kProjWaves + wave) * kProjTile;
    const bool active = t0 < sd.n_stft;   // a wave without a tile still fetches B rows and joins the barriers
    const bool want_chroma = (tb.flags & 2u) != 0, want_mel = (tb.flags & 5u) != 0;
    const int i = lane & 15, kk = lane >> 4;
    // A operands: the 16-frame tiles of this wave; group G of a tile starts at float4 index
    // mags_group_offset(G) and lane (frame i, kk) takes float4 i*8 + kk from there: an even group and the
    // odd one after it are the two halves of the same sixteen 128-byte lines (see mags_tile_index).  Frame
    // slots past the segment's end inside a tile were never written: their rows of C are garbage and are
    // not stored.  A wave loads only the tiles that hold frames (`live` below): the loop is instantiated per tile
    // count, so inside one instantiation every step issues the same loads and the compiler's waits stay exact
    // vmcnt(n) counts; a wave without a tile only fetches its share of the B rows and joins the barriers.
    const size_t f0 = static_cast<size_t>(sd.stft_base) + (active ? t0 : blockIdx.x * kProjWaves * kProjTile);
    const float4* ap[kProjHalves];
    ap[0] = reinterpret_cast<const float4*>(sc.mags) + (f0 >> 4) * kTileF4 + i * 8 + kk;
#pragma unroll
    for (int h = 1; h < kProjHalves; ++h) ap[h] = (active && t0 + 16 * h < sd.n_stft) ? ap[0] + h * kTileF4 : ap[0];
    const float4* cb = tb.chroma_b + (want_chroma ? static_cast<size_t>(sc.tuning_idx[blockIdx.y]) * kGroups * 64 : 0) + lane;
    const float4* mb = tb.mel_b + lane;
    float* mel_out = sc.mel + (static_cast<size_t>(sd.mel_base) + t0 + 4 * kk) * kMels + i;

    // top_db clips dB values more than 80 below the segment's maximum.  That maximum is at most
    // 10 log10(cmax^2 * mel_wsum_max) (cmax: a bound of the segment's largest |X|, from k_stft_frames), so a block
    // of dB values that all lie within 80 dB of this bound can never be clipped: k_segment_finish<2> will
    // take the tile's sum as it is and the rows need not be written.  The 0.01 dB cover the rounding of
    // the f32 sums and of log10f (< 3e-4 dB).
    const float cmax = __uint_as_float(sc.seg_cmax[blockIdx.y]);
    const float store_below = 10.0f * log10f(fmaxf(1e-10f, cmax * cmax * tb.mel_wsum_max)) + 0.01f - 80.0f;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const float4 zero4 = float4{0.f, 0.f, 0.f, 0.f};
    f32x4 c[kProjHalves], lo[kProjHalves], hi[kProjHalves];
#pragma unroll
    for (int h = 0; h < kProjHalves; ++h) c[h] = lo[h] = hi[h] = zero;
    int cur = 0;   // block held by slot lo

    // Retire slot lo (block cur).  C layout: column j = lane & 15 (filter), rows 4*kk + r (frames).  Per
    // 16-frame tile it leaves what k_segment_finish<2> reduces: the power in dB per frame
    // (power_to_db of mfcc, dsp.py:108), the power summed over the tile's frames (mel mean) and the
    // tile's largest dB value (top_db reference).  Sums run in a fixed order: results do not depend on
    // what else is in the batch.
    auto retire = [&]() {
#pragma unroll
        for (int h = 0; h < kProjHalves; ++h) {
            const int tb0 = t0 + 16 * h;
            if (tb0 < sd.n_stft) {   // wave-uniform
                const f32x4 v = lo[h];
                float psum = 0.0f, dmax = -INFINITY, dmin = INFINITY;
                double dsum = 0.0;
                float db[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    db[r] = 10.0f * log10f(fmaxf(1e-10f, v[r]));
                    if (tb0 + 4 * kk + r < sd.n_stft) {
                        psum += v[r];
                        dsum += static_cast<double>(db[r]);
                        dmax = fmaxf(dmax, db[r]);
                        dmin = fminf(dmin, db[r]);
                    }
                }
                if (__any(dmin < store_below)) {   // wave-uniform: some value of this block may be clipped
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (tb0 + 4 * kk + r < sd.n_stft)
                            mel_out[static_cast<size_t>(16 * h + r) * kMels + 16 * cur] = db[r];
                }
                // the four lanes (kk = 0..3) that hold one filter's 16 frames
                psum = sum_xor32(sum_xor16(psum));
                dsum = sum_xor_rows<true>(sum_xor_rows<false>(dsum));
                dmin = min_xor32(min_xor16(dmin));
                dmax = wave_max(dmax);
                const size_t tile = (f0 >> 4) + h;
                if (kk == 0) {
                    sc.mel_part[tile * kMels + 16 * cur + i] = psum;
                    sc.mel_dbsum[tile * kMels + 16 * cur + i] = dsum;
                    sc.mel_dbmin[tile * kMels + 16 * cur + i] = dmin;
                }
                if (lane == 0) sc.mel_dbmax[tile * 8 + cur] = dmax;
            }
        }
#pragma unroll
        for (int h = 0; h < kProjHalves; ++h) {
            lo[h] = hi[h];
            hi[h] = zero;
        }
        ++cur;
    };
    // B rows of a group: 0 chroma, 1 mel slot lo, 2 mel slot hi.  Load j of wave w fetches row w + j * kProjWaves;
    // a wave with nothing to fetch there (no such row, an idle mel slot, chroma off) still loads a row some other
    // wave of this group needs (an L1 hit) -- never the shared all-zero row 0, which every workgroup on the chip
    // would hammer in L2 -- so that every wave issues the same number of loads.
    auto fetch_b = [&](int G, const int4& p, int j) {
        const int row_id = wave + j * kProjWaves;
        const int live = want_mel ? p.z : 0;
        const float4* row = want_chroma ? cb + G * 64 : mb + static_cast<size_t>(max(p.x, p.y)) * 64;
        if (row_id == 1 && (live & 1)) row = mb + static_cast<size_t>(p.x) * 64;
        if (row_id == 2 && (live & 2)) row = mb + static_cast<size_t>(p.y) * 64;
        return *row;
    };
    // park the fetched rows of a group in LDS (row ids past 2 are the duplicates above)
    auto park_b = [&](float4 (*stage_group)[64], const float4 (&b)[kProjBLoads]) {
#pragma unroll
        for (int j = 0; j < kProjBLoads; ++j)
            if (wave + j * kProjWaves < 3) stage_group[wave + j * kProjWaves][lane] = b[j];
    };
    auto prog_at = [&](int G) { return prog.g[G]; };   // scalar loads (G is wave-uniform)
    // MFMAs of one group; its B rows sit in stage (G / 2) & 1, slot G & 1
    // `live` (a compile-time tag) = the number of this wave's tiles that hold frames of the segment: a wave loads and
    // multiplies only those.  (The first form had every wave issue the loads of all kProjHalves tiles, a missing tile
    // replaced by the first one: with 47-frame clips half of a workgroup's magnitude requests were such duplicates,
    // and the request path is what bounds this kernel.)
    auto compute = [&](auto live, int G, const float4 (&a)[kProjHalves], const int4& p) {
        constexpr int kLive = decltype(live)::value;
        if constexpr (kLive == 0) return;
        const float4* sb = &s_b[(G >> 1) & 1][G & 1][0][lane];
        if (want_chroma) {
            const float4 bc = sb[0];
#pragma unroll
            for (int h = 0; h < kLive; ++h) c[h] = mfma4(a[h], bc, c[h]);
        }
        if (want_mel) {
            float4 pw[kProjHalves];   // |X|^2 of the rounded |X|
#pragma unroll
            for (int h = 0; h < kLive; ++h) pw[h] = float4{a[h].x * a[h].x, a[h].y * a[h].y, a[h].z * a[h].z, a[h].w * a[h].w};
            if (p.z & 1) {   // wave-uniform
                const float4 bl = sb[64];
#pragma unroll
                for (int h = 0; h < kLive; ++h) lo[h] = mfma4(pw[h], bl, lo[h]);
            }
            if (p.z & 2) {
                const float4 bh = sb[128];
#pragma unroll
                for (int h = 0; h < kLive; ++h) hi[h] = mfma4(pw[h], bh, hi[h]);
            }
            for (int n = p.w; n > 0; --n) retire();
        }
    };
    auto load_a = [&](auto live, float4 (&dst)[kProjHalves], int G) {
#pragma unroll
        for (int h = 0; h < decltype(live)::value; ++h) dst[h] = load_mags(ap[h] + mags_group_offset(G));
    };

    // Software pipeline over register sets, no register copies: E for even groups, O and P alternately for odd groups.
    // One step = the even group G and the odd group G + 1, whose B rows sit in stage (G / 2) & 1.  The step requests
    // the B rows of the next pair first, multiplies group G, requests the A rows of BOTH groups G + 2 and G + 3 (the
    // two halves of the same 128-byte lines: requested back to back the second load finds the lines the first one
    // brought; a step apart it found them evicted, +17 % kernel time), multiplies group G + 1 with the odd set loaded
    // a step ago (`ao`; the even step refills `an`, the other odd set), and only then parks the new B rows in the
    // other stage -- their loads have had two groups of MFMAs to land -- behind ONE barrier per pair (per group in
    // the first form: +1.6 %; per four groups: equal): everybody finished reading that stage before the previous
    // step's barrier.  Every wave of the workgroup runs the same number of steps and barriers whatever its `live`.
    auto run = [&](auto live) {
        float4 aE[kProjHalves], aO[kProjHalves], aP[kProjHalves];
#pragma unroll
        for (int h = 0; h < kProjHalves; ++h) aE[h] = aO[h] = aP[h] = zero4;
        load_a(live, aE, 0);
        load_a(live, aO, 1);
        float4 bE[kProjBLoads], bO[kProjBLoads];
#pragma unroll
        for (int j = 0; j < kProjBLoads; ++j) {
            bE[j] = fetch_b(0, prog_at(0), j);
            bO[j] = fetch_b(1, prog_at(1), j);
        }
        park_b(s_b[0][0], bE);
        park_b(s_b[0][1], bO);
        __syncthreads();
        auto pair_step = [&](int G, const float4 (&ao)[kProjHalves], float4 (&an)[kProjHalves]) {
            const int G3 = min(G + 3, kGroups - 1);         // the last step fetches a B row nobody uses
#pragma unroll
            for (int j = 0; j < kProjBLoads; ++j) {
                bE[j] = fetch_b(G + 2, prog_at(G + 2), j);  // G + 2 <= 64
                bO[j] = fetch_b(G3, prog_at(G3), j);
            }
            // issue priority: up from the first MFMA cluster of the step to the end of the second (the A loads between
            // them included), down for the B fetch, the parking and the barrier -- a wave that has its operands gets to
            // the matrix pipe ahead of waves that are only requesting theirs (1.23 -> 1.21 ms round robin in one call;
            // grading by progress through the bin groups, or prologue / epilogue first: no change)
            __builtin_amdgcn_s_setprio(2);
            compute(live, G, aE, prog_at(G));
            load_a(live, aE, G + 2);
            load_a(live, an, G + 3);                        // G + 3 = 65 at the end: the unused half of the last block
            compute(live, G + 1, ao, prog_at(G + 1));
            __builtin_amdgcn_s_setprio(0);
            const int next = ((G >> 1) + 1) & 1;
            park_b(s_b[next][0], bE);
            park_b(s_b[next][1], bO);
            __syncthreads();
        };
        static_assert((kGroups - 1) % 4 == 0, "the loop below takes four groups per trip");
#pragma unroll 1
        for (int G = 0; G < kGroups - 1; G += 4) {
            pair_step(G, aO, aP);
            pair_step(G + 2, aP, aO);
        }
        // group 64 (kGroups is odd) holds only bin 1024 and three zero pads, in the kk = 0 chunk; the other
        // chunks of that group were never written
        if (kk != 0) {
#pragma unroll
            for (int h = 0; h < kProjHalves; ++h) aE[h] = zero4;
        }
        compute(live, kGroups - 1, aE, prog_at(kGroups - 1));
    };
    // tiles of this wave that hold frames (wave-uniform): 0 for a wave past the segment's end
    const int live_tiles = active ? min(kProjHalves, (sd.n_stft - t0 + 15) >> 4) : 0;
    static_assert(kProjHalves <= 3, "the dispatch below lists the tile counts");
    if (live_tiles >= kProjHalves) run(std::integral_constant<int, kProjHalves>{});
    else if (live_tiles == 0) run(std::integral_constant<int, 0>{});
    else if (live_tiles == 1) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, (kProjHalves > 2 ? 2 : 1)>{});
    if (active && want_mel)
        while (cur < 8) retire();
    if (!active) return;

    // ---- chroma epilogue: per-frame inf-norm over the 16 lanes that hold one frame's classes
    if (want_chroma) {
        const float tiny = 1.17549435e-38f;
#pragma unroll
        for (int half = 0; half < kProjHalves; ++half) {
            const f32x4 cc = c[half];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + 16 * half + 4 * kk + r;
                const float raw = cc[r];
                float mx = row_max(fabsf(raw));
                if (mx < tiny) mx = 1.0f;
                if (t < sd.n_stft) sc.chroma_norm[(static_cast<size_t>(sd.stft_base) + t) * 16 + i] = raw / mx;
            }
        }
    }
}

}  // namespace serhip
