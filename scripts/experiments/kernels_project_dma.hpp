// k_project_frames: the two filterbank projections of the fast path on the matrix cores.
//
//   chroma_raw[t, c] = sum_k fb_tuning[c, k] * |X|[k, t]        (librosa.feature.chroma_stft, dsp.py:115)
//   mel[t, m]        = sum_k W[m, k]         * |X|^2[k, t]      (melspectrogram inside mfcc/mel, dsp.py:108,122)
//
// One wave owns a tile of 32 frames of one segment and walks the 1025 bins in groups of 16 with
// v_mfma_f32_16x16x4_f32 (exact f32 fma chain): A = 16 frames x 4 bins of the magnitude rows (each
// lane 16 bytes per group: the contraction index may be permuted freely, so lane (frame i, kk) takes
// bins 16G+4kk..+3 and uses them as 4 consecutive k-steps), B = the filterbank pre-arranged on the
// host in the same order.  Both arrive by LDS-DMA as whole lines (see "Operand transport").  The mel filterbank is banded, so a
// 16-filter block only visits the groups its filters touch (~1.3 blocks per group).
// Epilogue: per-frame inf-norm of the chroma row (a 16-lane reduction) -> chroma_norm[t, 16]; per
// retired mel block and 16-frame tile the power sums, dB sums, dB minima and the dB maximum, and the
// dB rows mel[t, 128] only where top_db might clip them; k_segment_finish<2> takes the time means.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_fast.hpp"

namespace serhip {

// Diagnostic build (-DSERHIP_K2_STAMPS=1): per-phase cycle sums of k_project_frames, summed over all waves into
// g_k2_stamps and printed by the host after the launch.
#ifndef SERHIP_K2_STAMPS
#define SERHIP_K2_STAMPS 0
#endif
#if SERHIP_K2_STAMPS
__device__ unsigned long long g_k2_stamps[16];
#define K2_STAMP(i)                                                              \
    {                                                                            \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();           \
        stamp_acc[i] += static_cast<unsigned>(t_now - stamp_prev);               \
        stamp_prev = t_now;                                                      \
    }
#else
#define K2_STAMP(i)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kProjTile = 32;   // frames per wave
constexpr int kProjWaves = 3;   // waves per workgroup: one B row each to fetch; 96 frames = one 3 s clip

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
// and how many blocks retire after the group.
//
// Operand transport.  The kernel is bound by the magnitude stream (4.16 KB per frame against ~18 MFMAs per
// 16 bins).  Fragment-shaped loads straight to registers (lane (i, kk): 16 bytes of frame i -> 16 lines x 64 B
// per wave-instruction) cost the texture path one line per lane group and ran at 4.6 TB/s of fetches with a
// step time that did not depend on occupancy; so the operands travel as whole 128-byte lines by LDS-DMA
// (global_load_lds_dwordx4: 1 KB = 8 lines per wave-instruction, no VGPR destination) and the fragments are
// cut from LDS:
//   * A: per step one pair block (32 bins = the even group G and the odd group G + 1) of the wave's two
//     16-frame tiles, 4 pieces of 8 frames x 128 B, into a private ring of kProjPairs slots.  LDS-DMA writes
//     lane l's 16 bytes at 16 l, so the image is swizzled through the SOURCE address: slot (f, c ^ ((f >> 1) & 7))
//     holds chunk c of frame f.  Every 8 lanes still fetch one whole line, and the fragment read of lane
//     (i, kk) -- chunk kk of frame i for the even group, chunk kk + 4 for the odd one -- is conflict-free over
//     ds_read_b128's four 16-lane groups ({0-3,12-15,20-27}, ...: each holds all 16 frames once, chunk kk for
//     frames 0-3 and 12-15, chunk kk ^ 1 for frames 4-11).  Ordered by the wave's own vmcnt.
//   * B: wave w fetches row w (0 chroma | 1 mel slot lo | 2 mel slot hi) of both groups of the NEXT pair into
//     a ring of 2 slots shared by the three waves, ordered by vmcnt + the step's barrier.
// hipcc does not count asm memory operations, so the waits are hand-counted.  The only VMEM loads in the loop
// are these, issued unconditionally by every wave in every step -- 2 B pieces at the top, 4 A pieces after the
// LDS reads -- and they complete in order: `s_waitcnt vmcnt(4)` at the end of step P has retired B(P + 1) and,
// older than it, A(P + 1); only A(P + 2) stays in flight across the barrier (the compiler's own stores in
// retire() only make the wait more conservative).  Slot reuse: the A slot of pair P is refilled in step P
// behind an lgkmcnt(0) that retires its reads; the B slot refilled at the top of step P is that of pair P - 1,
// which every wave finished reading before the barrier that ended step P - 1.  A wave drains its DMAs
// (vmcnt(0)) before it exits: a piece landing later would fall into the LDS of the workgroup that inherits it.
constexpr int kProjPairs = 2;   // A ring: pair blocks per wave (4 KB each)
static_assert(kProjPairs >= 2, "the pair in hand plus at least one in flight");

__device__ __forceinline__ const char* wave_uniform(const void* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
    const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
    return reinterpret_cast<const char*>((static_cast<unsigned long long>(hi) << 32) | lo);
}
// LDS byte address of a __shared__ object (the low half of its flat address), wave-uniform
__device__ __forceinline__ unsigned lds_address(const void* p) {
    return __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<unsigned long long>(p)));
}
// lane l: 16 bytes from sbase + voff(l) to LDS byte lds_dst + 16 l.  M0 carries the LDS base; it is
// compiler-reserved, so it is written and restored inside the one statement.
__device__ __forceinline__ void lds_dma_16(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// the same with 4 bytes per lane (256 B per wave-instruction)
__device__ __forceinline__ void lds_dma_4(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

__global__ __launch_bounds__(kProjWaves * 64) void k_project_frames(const SegDesc* __restrict__ segs, PlanTables tb, Scratch sc) {
    __shared__ float4 s_a[kProjWaves][kProjPairs][2][128];    // [wave][slot][tile half][swizzled (frame, chunk)]
    __shared__ float4 s_b[2][2][3][64];                       // [slot][group parity][row][lane]
    __shared__ int4 s_prog[kGroups];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SegDesc sd = segs[blockIdx.y];
    if (blockIdx.x * kProjWaves * kProjTile >= sd.n_stft) return;   // whole workgroup
    const int t0 = (blockIdx.x * kProjWaves + wave) * kProjTile;
    const bool active = t0 < sd.n_stft;   // a wave without a tile still fetches B rows and joins the barriers
    const bool want_chroma = (tb.flags & 2u) != 0, want_mel = (tb.flags & 5u) != 0;
    const int i = lane & 15, kk = lane >> 4;
    // A operands: the two 16-frame tiles of this wave; group G of a tile starts at float4 index
    // mags_group_offset(G) and lane (frame i, kk) takes float4 i*8 + kk from there: an even group and the
    // odd one after it are the two halves of the same sixteen 128-byte lines (see mags_tile_index).  Frame
    // slots past the segment's end inside a tile were never written: their rows of C are garbage and are
    // not stored.  A second tile that lies wholly past the end is replaced by the first; a wave without a
    // tile reads the workgroup's first tile and discards it.
    const size_t f0 = static_cast<size_t>(sd.stft_base) + (active ? t0 : blockIdx.x * kProjWaves * kProjTile);
#if SERHIP_EXP_A_HOT   // timing experiment: every wave reads the first tiles (L2 hits)
    const char* a0s = wave_uniform(reinterpret_cast<const float4*>(sc.mags) + (wave + 3 * (blockIdx.x & 7)) * 2 * kTileF4);
#else
    const char* a0s = wave_uniform(reinterpret_cast<const float4*>(sc.mags) + (f0 >> 4) * kTileF4);
#endif
    const char* a1s = (active && t0 + 16 < sd.n_stft) ? a0s + kTileBytes : a0s;
    // DMA source offsets inside a pair block for the pieces of frames 0-7 and 8-15, and the fragment's LDS offset
    const unsigned a_src0 = static_cast<unsigned>((lane >> 3) * 128 + (((lane & 7) ^ (lane >> 4)) << 4));
    const unsigned a_src1 = static_cast<unsigned>((8 + (lane >> 3)) * 128 + (((lane & 7) ^ ((lane >> 4) + 4)) << 4));
    const int a_frag = i * 8 + (kk ^ ((i >> 1) & 7));   // float4 index for the even group; ^ 4 for the odd one
#if SERHIP_EXP_B_HOT   // timing experiment: one tuning's rows for every segment (L2 hits)
    const float4* cb = tb.chroma_b + (sc.tuning_idx[blockIdx.y] & 0);
#else
    const float4* cb = tb.chroma_b + (want_chroma ? static_cast<size_t>(sc.tuning_idx[blockIdx.y]) * kGroups * 64 : 0);
#endif
    const float4* mb = tb.mel_b;
    float* mel_out = sc.mel + (static_cast<size_t>(sd.mel_base) + t0 + 4 * kk) * kMels + i;

    // top_db clips dB values more than 80 below the segment's maximum.  That maximum is at most
    // 10 log10(cmax^2 * mel_wsum_max) (cmax: the segment's largest |X|, from k_stft_frames), so a block
    // of dB values that all lie within 80 dB of this bound can never be clipped: k_segment_finish<2> will
    // take the tile's sum as it is and the rows need not be written.  The 0.01 dB cover the rounding of
    // the f32 sums and of log10f (< 3e-4 dB).
    const float cmax = __uint_as_float(sc.seg_cmax[blockIdx.y]);
    const float store_below = 10.0f * log10f(fmaxf(1e-10f, cmax * cmax * tb.mel_wsum_max)) + 0.01f - 80.0f;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const float4 zero4 = float4{0.f, 0.f, 0.f, 0.f};
    f32x4 c0 = zero, c1 = zero;
    f32x4 lo0 = zero, lo1 = zero, hi0 = zero, hi1 = zero;
    int cur = 0;   // block held by slot lo

    // Retire slot lo (block cur).  C layout: column j = lane & 15 (filter), rows 4*kk + r (frames).  Per
    // 16-frame half tile it leaves what k_segment_finish<2> reduces: the power in dB per frame
    // (power_to_db of mfcc, dsp.py:108), the power summed over the tile's frames (mel mean) and the
    // tile's largest dB value (top_db reference).  Sums run in a fixed order: results do not depend on
    // what else is in the batch.
    auto retire = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int tb0 = t0 + 16 * h;
            if (tb0 < sd.n_stft) {   // wave-uniform
                const f32x4 v = h ? lo1 : lo0;
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
#if SERHIP_EXP_NO_ROWS   // timing experiments with garbage operands must not trigger the dB-row stores
                if (false) {
#else
                if (__any(dmin < store_below)) {   // wave-uniform: some value of this block may be clipped
#endif
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
        lo0 = hi0;
        lo1 = hi1;
        hi0 = hi1 = zero;
        ++cur;
    };
    // the per-group program, read as scalars
    auto prog_at = [&](int G) {
        const int4 v = s_prog[G];
        return int4{__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y),
                    __builtin_amdgcn_readfirstlane(v.z), __builtin_amdgcn_readfirstlane(v.w)};
    };
    // wave w fetches B row w of a group (0 chroma, 1 mel slot lo, 2 mel slot hi).  A wave with nothing to
    // fetch (an idle mel slot, chroma off) still loads a row some other wave of this group needs
    // -- never the shared all-zero row 0, which every workgroup on the chip would hammer in L2.
    auto b_row = [&](int G, const int4& p) -> const char* {
        const int live = want_mel ? p.z : 0;
        const float4* row = want_chroma ? cb + G * 64 : mb + static_cast<size_t>(max(p.x, p.y)) * 64;
        if (wave == 1 && (live & 1)) row = mb + static_cast<size_t>(p.x) * 64;
        if (wave == 2 && (live & 2)) row = mb + static_cast<size_t>(p.y) * 64;
        return wave_uniform(row);
    };
    const unsigned lds_a = lds_address(&s_a[wave][0][0][0]);
    const unsigned lds_b = lds_address(&s_b[0][0][wave][0]);
    // pair blocks past the last one are clamped to it: the steps at the end re-request it, so that the number of
    // operations per step never changes
    auto request_a = [&](int P, int slot) {
        const unsigned goff = static_cast<unsigned>(min(P, kPairBlocks - 1)) * 2048u;
        const unsigned dst = lds_a + static_cast<unsigned>(slot) * 4096u;
        lds_dma_16(a0s + goff, a_src0, dst);
        lds_dma_16(a0s + goff, a_src1, dst + 1024u);
        lds_dma_16(a1s + goff, a_src0, dst + 2048u);
#if !SERHIP_EXP_K2_A34   // timing experiment: three quarters of the magnitude bytes
        lds_dma_16(a1s + goff, a_src1, dst + 3072u);
#endif
    };
    auto request_b = [&](int P, int slot) {
#if SERHIP_EXP_B_CHROMA_ONLY   // timing experiment: one chroma row per wave 0/1, nothing for mel
        if (wave < 2) {
            const int G = min(2 * P + wave, kGroups - 1);
            lds_dma_16(wave_uniform(cb + G * 64), static_cast<unsigned>(lane) * 16u, lds_b + static_cast<unsigned>(slot) * 6144u);
        }
        return;
#endif
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int G = min(2 * P + e, kGroups - 1);
#if SERHIP_EXP_B_SMALL   // timing experiment: a quarter of the B bytes
            lds_dma_4(b_row(G, prog_at(G)), static_cast<unsigned>(lane) * 4u,
                      lds_b + static_cast<unsigned>(slot) * 6144u + static_cast<unsigned>(e) * 3072u);
#else
            lds_dma_16(b_row(G, prog_at(G)), static_cast<unsigned>(lane) * 16u,
                       lds_b + static_cast<unsigned>(slot) * 6144u + static_cast<unsigned>(e) * 3072u);
#endif
        }
    };
    // MFMAs of one group
    auto compute = [&](const float4& a0, const float4& a1, const float4& bc, const float4& bl, const float4& bh, const int4& p) {
#if SERHIP_EXP_NO_MFMA   // timing experiment
        asm volatile("" ::"v"(a0.x), "v"(a1.x), "v"(bc.x), "v"(bl.x), "v"(bh.x));
        return;
#endif
        if (want_chroma) {
            c0 = mfma4(a0, bc, c0);
            c1 = mfma4(a1, bc, c1);
        }
        if (want_mel) {
            const float4 p0 = float4{a0.x * a0.x, a0.y * a0.y, a0.z * a0.z, a0.w * a0.w};   // |X|^2 of the rounded |X|
            const float4 p1 = float4{a1.x * a1.x, a1.y * a1.y, a1.z * a1.z, a1.w * a1.w};
            if (p.z & 1) {   // wave-uniform
                lo0 = mfma4(p0, bl, lo0);
                lo1 = mfma4(p1, bl, lo1);
            }
            if (p.z & 2) {
                hi0 = mfma4(p0, bh, hi0);
                hi1 = mfma4(p1, bh, hi1);
            }
#if !SERHIP_EXP_NO_RETIRE
            for (int n = p.w; n > 0; --n) retire();
#endif
        }
    };

    for (int j = threadIdx.x; j < kGroups; j += kProjWaves * 64) s_prog[j] = tb.mel_prog[j];
    asm volatile("" ::"v"(store_below), "v"(mel_out));   // the plain loads above are consumed before the first DMA
    __syncthreads();
    request_b(0, 0);
#pragma unroll
    for (int g = 0; g < kProjPairs; ++g) request_a(g, g);
#if SERHIP_EXP_K2_A34
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (kProjPairs - 1)) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kProjPairs - 1)) : "memory");
#endif
    __builtin_amdgcn_s_barrier();

    int sa = 0;   // A ring slot of pair P
#if SERHIP_K2_STAMPS
    unsigned stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
#endif
#if SERHIP_EXP_SHORT_LOOP   // timing experiment: what the kernel costs outside its main loop
    constexpr int kLoopEnd = 2;
#else
    constexpr int kLoopEnd = kPairBlocks;
#endif
#pragma unroll 1
    for (int P = 0; P < kLoopEnd; ++P) {
        request_b(P + 1, (P + 1) & 1);
#if SERHIP_K2_STAMPS == 1
        K2_STAMP(0)
#endif
        const float4* fa = &s_a[wave][sa][0][0];
        float4 e0 = fa[a_frag], o0 = fa[a_frag ^ 4], e1 = fa[128 + a_frag], o1 = fa[128 + (a_frag ^ 4)];
        const float4* fb = &s_b[P & 1][0][0][lane];
        const float4 ebc = fb[0], ebl = fb[64], ebh = fb[128], obc = fb[192], obl = fb[256], obh = fb[320];
        // pair P + kProjPairs goes into the slot just read
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if SERHIP_K2_STAMPS == 1
        K2_STAMP(1)
#endif
        request_a(P + kProjPairs, sa);
#if SERHIP_K2_STAMPS == 1
        K2_STAMP(2)
#endif
        if (active) {
            // group 64 (kGroups is odd) holds only bin 1024 and three zero pads, in chunk 0; the other chunks of
            // that group were never written, and there is no group 65
            if (P == kPairBlocks - 1 && kk != 0) e0 = e1 = zero4;
            compute(e0, e1, ebc, ebl, ebh, prog_at(2 * P));
            if (P < kPairBlocks - 1) compute(o0, o1, obc, obl, obh, prog_at(2 * P + 1));
        }
        K2_STAMP(3)
#if SERHIP_EXP_K2_A34
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#endif
#if SERHIP_K2_STAMPS == 1
        K2_STAMP(4)
#endif
        __builtin_amdgcn_s_barrier();
        K2_STAMP(5)
        sa = sa + 1 == kProjPairs ? 0 : sa + 1;
    }
#if SERHIP_K2_STAMPS
    if (lane == 0) {
        for (int j = 0; j < 6; ++j) atomicAdd(&g_k2_stamps[j], static_cast<unsigned long long>(stamp_acc[j]));
        atomicAdd(&g_k2_stamps[6], 1ull);
        atomicAdd(&g_k2_stamps[7], active ? 1ull : 0ull);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may outlive the workgroup's LDS
    if (active && want_mel)
        while (cur < 8) retire();
    if (!active) return;

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
                float mx = row_max(fabsf(raw));
                if (mx < tiny) mx = 1.0f;
                if (t < sd.n_stft) sc.chroma_norm[(static_cast<size_t>(sd.stft_base) + t) * 16 + i] = raw / mx;
            }
        }
    }
}

}  // namespace serhip
