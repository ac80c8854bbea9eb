// gfx950 kernels of the fast feature path (segments with >= 2048 samples, n_fft = 2048,
// hop 512) -- the GPU form of ser/_internal/utils/dsp.py:96-136.
//
//   k_stft_frames      persistent, one wave64 per STFT frame: framing + Hann + 2048-pt real FFT
//                      (1024-pt complex, radix 16 x 16 x 4, two LDS transposes, the radix-4
//                      across quads with DPP) and |X|.  Writes per frame: the magnitudes in the
//                      tile order k_project_frames reads, the (pitch, mag) piptrack peak list;
//                      per segment the largest magnitude.
//   k_segment_finish<1> one workgroup per segment, before the projection: tuning estimate
//                      (median of the peak magnitudes by radix select + 100-bin histogram).
//   k_project_frames   (kernels_project.hpp) chroma and mel projections on the matrix cores.
//   k_segment_finish<2> one workgroup per segment, after the projection: top_db clip and time
//                      mean of the log-mel -> DCT (MFCC), mel mean, chroma mean, contrast zeros.
//   k_segment_finish<0> the same two steps plus an on-device chroma filterbank for the generic
//                      path (kernels_generic.hpp: segments shorter than 2048 samples).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "fft_math.hpp"
#include "tables.hpp"

namespace serhip {

// This header holds the shipped form only.  The A/B and timing-only variants behind DESIGN.md's tables of
// k_stft_frames experiments (cache policies of the stores, one atomic per frame, hot loads / hot stores, three
// quarters of the magnitude bytes, ...) live in scripts/experiments/variants/ as a patch against this file
// (scripts/experiments/README.md) -- several of them produce garbage on purpose and must not be one -D away
// from a product build.  One switch remains:
//   SERHIP_K1_STAMPS=1   diagnostic build: per-phase shader-cycle sums of every wave of k_stft_frames, taken with
//                        s_memtime at the points where the wave already waits for lgkmcnt(0) (s_memtime shares that
//                        counter with LDS), written to Scratch::mel (unused by this kernel) and printed by the host
//                        (DESIGN.md section 4, round 2).  Results stay correct; never in a timed or shipped build.
#ifndef SERHIP_K1_STAMPS
#define SERHIP_K1_STAMPS 0
#endif
#if SERHIP_K1_STAMPS
#define K1_STAMP(i)                                                              \
    {                                                                            \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();           \
        stamp_acc[i] += static_cast<unsigned>(t_now - stamp_prev);               \
        stamp_prev = t_now;                                                      \
    }
#else
#define K1_STAMP(i)
#endif
// k_stft_frames runs ONE workgroup per CU: its waves share the 21 KB of twiddle/window tables in LDS and
// each owns an 8.5 KB transpose buffer.  16 waves = 4 per SIMD at <= 128 VGPRs fill the 160 KB of LDS;
// measured on MI355X (12.5k clips): 8 waves 4.2 ms, 12 waves 2.9 ms, 16 waves 2.8 ms.
constexpr int kWaves = 16;             // waves per workgroup in k_stft_frames
constexpr int kUnitFramesMax = 128;    // STFT frames per work unit of k_stft_frames (small calls use less)
constexpr int kBufC = 1088;            // complex slots of one wave's LDS buffer (16 rows x stride 68)
constexpr int kGroups = 65;            // 16-bin groups covering bins 0..1039 (1025 used)
constexpr int kPeakLdsCap = 2048;      // peaks staged in LDS by k_segment_finish
// bit of PlanTables::flags set for ONE launch of k_segment_finish<1>: k_tuning_waves ran before it on the same
// segments and sc.tuning_idx holds its results (>= 0) or -1 for the segments it left
constexpr unsigned kFlagTuningWavesRan = 1u << 30;

// per-segment descriptor (device copy of what the host planned)
struct SegDesc {
    long long offset;     // first sample in the audio buffer
    long long out_row;    // row of the output matrix
    int len;              // samples in the segment (before the >=512 zero padding)
    int n_fft;            // 2048 on the fast path, max(len,512) on the generic path
    int n_stft;           // STFT columns (hop n_fft/4)
    int n_mel;            // mel-STFT columns (hop 512)
    int stft_base;        // first row in the magnitude / peak scratch
    int mel_base;         // first row in the mel scratch
};

// work unit of k_stft_frames: frames [t_begin, min(t_begin + unit_frames, n_stft)) of segment seg.  The
// kernel is persistent: workgroup b walks the units the host gave it (balanced by frame count: a plain
// b, b + gridDim.x, ... walk puts all long units on the same workgroups when segment lengths alternate
// with a period that divides the grid) and its waves draw frame after frame of the concatenation of
// those units from a counter in LDS (no workgroup barrier after set-up; the frames in flight in a
// workgroup stay neighbours, so the 75 % they share is read from HBM once).
// The array the kernel gets starts with gridDim.x + 1 header entries whose `seg` field is the index of
// workgroup b's first unit in the list that follows (the last one: the number of units).
struct WorkUnit {
    int seg;              // index into the chunk's descriptor array
    int t_begin;
};

// constant tables of one plan (device pointers)
struct PlanTables {
    const v2f* twA;       // [16][64]  W_1024^(lane*k1)
    const v2f* twB;       // [16][4]   W_64^(m2*j1)
    const v2f* twU;       // [8][64]   W_2048^(lane+64r)
    const v2f* win;       // [16][64]  Hann pairs (w[128n1+2l], w[128n1+2l+1])
    // B operands of k_project_frames in v_mfma_f32_16x16x4_f32 order: group G = 16 bins, lane
    // (kk = lane>>4, j = lane&15) holds float4 {w[16G+4kk+u][j]}, u = 0..3
    const float4* chroma_b;  // [101 tunings][65 groups][64 lanes]
    const float4* mel_b;     // rows of 64 lanes: row 0 all zeros, then per 16-filter block its live groups
    const int4* mel_prog;    // [65] per group: {row of slot lo, row of slot hi, live mask (1 lo, 2 hi), retires}
    const float* dct;     // [40][128]
    const double* edges;  // [101] histogram edges np.linspace(-0.5, 0.5, 101)
    const double* mel_f;  // [130] mel band edges in Hz (generic path)
    int kmin, kmax;       // piptrack bin range (inclusive) for n_fft 2048
    int sample_rate;
    unsigned flags;
    float mel_wsum_max;   // largest row sum of the mel bank: mel power <= (largest |X|)^2 * mel_wsum_max
};

// per-chunk scratch (device pointers)
struct Scratch {
    float* mags;          // generic path: [frames][kBinStride]; fast path: tiled, see mags_tile_index
    float* mel;           // [frames][128]  generic path: mel power; fast path: mel power in dB (k_project_frames)
    float* mel_part;      // [frames/16][128]  fast path: mel power summed over the 16 frames of a tile
    float* mel_dbmax;     // [frames/16][8]    fast path: max dB per (tile, 16-filter block)
    double* mel_dbsum;    // [frames/16][128]  fast path: dB values summed over the 16 frames of a tile
    float* mel_dbmin;     // [frames/16][128]  fast path: smallest dB value of the tile
    v2f* peaks;           // [frames][peak_cap]  (pitch, mag)
    int* peak_cnt;        // [frames] fast path: peak count | frame-max key (see kPeakCountMask); generic path: the count
    float* chroma_norm;   // [frames][16]  per-frame inf-normalised chroma (classes 12..15 unused)
    int* tuning_idx;      // [segments of the chunk]
    unsigned* seg_cmax;   // [segments of the chunk] bit pattern of the segment's largest |X| (atomicMax, zeroed per call)
    int peak_cap;         // peak slots per frame
};

// Fast-path magnitudes are stored in the order k_project_frames consumes them: frame slots in tiles of
// 16, and inside a tile [block of 32 bins][frame][32 bins].  One frame owns whole 128-byte lines (a
// [16 bins]-per-frame layout would share every line between two frames, i.e. two waves of
// k_stft_frames: measured, the half-line stores bound that kernel at 1.9 TB/s), and the A operands of
// an even/odd pair of MFMA groups are the two interleaved halves of one 2 KB block.  Returns the float4
// index of bins 4q..4q+3 of frame slot f.  Every segment starts on a tile boundary (the host pads
// stft_base to a multiple of 16).
constexpr int kPairBlocks = (kGroups + 1) / 2;          // 33 blocks of 32 bins (the last one half used)
constexpr int kTileF4 = kPairBlocks * 128;              // float4 slots per 16-frame tile
constexpr int kTileBytes = kTileF4 * 16;                // 67 584 B per 16 frames
__host__ __device__ __forceinline__ size_t mags_tile_index(size_t f, int q) {
    return (f >> 4) * kTileF4 + static_cast<size_t>(q >> 3) * 128 + (f & 15) * 8 + (q & 7);
}
// float4 offset, inside a tile, of MFMA group G (16 bins) for lane (frame 0, kk 0)
// k_stft_frames packs two things into a frame's peak_cnt word: the peak count in the low 10 bits and, above them,
// the bit pattern of an UPPER bound of the frame's largest magnitude (rounded up to a multiple of 1024 in bit
// space, i.e. by at most 2^-13 relative).  k_segment_finish<1> reads the counts anyway and takes the maximum of the
// keys as the segment's bound for k_project_frames' row-store test -- which only needs a bound, so the results do
// not change -- instead of one atomicMax per frame from the FFT kernel (5 % of that kernel).
constexpr unsigned kPeakCountMask = 1023u;
__host__ __device__ __forceinline__ int mags_group_offset(int G) { return (G >> 1) * 128 + (G & 1) * 4; }

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}

// make this wave's LDS writes visible to its own later reads (one wave owns the buffer)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// max over the 64 lanes, returned to every lane.  DPP only (quad swaps, half-row and row mirrors, then
// the gfx9 row broadcasts): six v_max_f32 and a readlane instead of six ds_bpermute round trips.
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_mov<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_mov<0x141>(v));   // row_half_mirror
    v = fmaxf(v, dpp_mov<0x140>(v));   // row_mirror: every row of 16 now holds its max
    // row_bcast:15 (rows 1 and 3 take lane 15 of the row before), row_bcast:31 (rows 2-3 take lane 31)
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x142, 0xA, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x143, 0xC, 0xF, false)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// the same reduction on unsigned integers (bit patterns of non-negative floats)
__device__ __forceinline__ unsigned wave_max_bits(unsigned v) {
    auto dpp = [](unsigned x, auto ctrl) {
        return static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    v = max(v, dpp(v, std::integral_constant<int, 0xB1>{}));
    v = max(v, dpp(v, std::integral_constant<int, 0x4E>{}));
    v = max(v, dpp(v, std::integral_constant<int, 0x141>{}));
    v = max(v, dpp(v, std::integral_constant<int, 0x140>{}));
    v = max(v, static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x142, 0xA, 0xF, false)));
    v = max(v, static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x143, 0xC, 0xF, false)));
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

// piptrack arithmetic for one local maximum (librosa.piptrack, called from estimate_tuning
// inside chroma_stft; dsp.py:115).  sm/s0/sp = S[i-1], S[i], S[i+1] (f32 magnitudes).
// The parabolic shift is a double division of float32 sums, as numba evaluates the stencil.
template <bool kPow2>
__device__ __forceinline__ v2f piptrack_peak(float sm, float s0, float sp, int i, double bin_hz_num,
                                             double n_fft_d) {
#pragma clang fp contract(off)  // numpy / numba round every product and sum separately
    const float sum = sp + sm;
    const float dif = sp - sm;
    const double a = static_cast<double>(sum) - 2.0 * static_cast<double>(s0);
    const double b = static_cast<double>(dif) / 2.0;
    float shift = 0.0f;
    if (!(fabs(b) >= fabs(a))) shift = static_cast<float>(-b / a);
    const float avg = dif / 2.0f;                 // np.gradient interior point, f32
    const float dskew = __fmul_rn(0.5f * avg, shift);   // separately rounded, as numpy does
    // (i + shift) * sr / n_fft; for a power-of-two n_fft the division is an exact scaling
    const double scaled = (static_cast<double>(i) + static_cast<double>(shift)) * bin_hz_num;
    const double pitch = kPow2 ? scaled * (1.0 / n_fft_d) : scaled / n_fft_d;
    return v2f{static_cast<float>(pitch), __fadd_rn(s0, dskew)};
}

// One 8-byte sample pair per lane from `sbase + voff + kImm` (bytes), issued as inline asm so that the
// compiler does not know a load is in flight: it then inserts no s_waitcnt of its own, and the kernel waits
// with the exact count it needs (k_stft_frames, "sample loads").  global_load_dwordx2 needs 4-byte alignment
// only (unaligned access mode is on under ROCm), so segments at odd sample offsets take this path too.
template <int kImm>
__device__ __forceinline__ void load_pair_async(v2f& dst, const float* sbase, unsigned voff) {
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(kImm));
}
// One-lane stores of the store block: issued with EXEC narrowed to lane 0 inside the asm statement instead of an
// `if (lane == 0)` around a plain store.  The branch the compiler puts around such an `if` (s_cbranch_execz) is
// never taken here, but a path-insensitive count of the block's operations cannot know that; and letting all 64
// lanes store the same value to the same place moves 64 times the data into the memory pipeline (measured: 2 % of
// the kernel).  Vector stores; the scalar data cache is not written.
#define SERHIP_LANE0_NT ""
__device__ __forceinline__ void store_lane0(int* dst_uniform, int value) {
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "global_store_dword %1, %2, %3" SERHIP_LANE0_NT "\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(0), "v"(value), "s"(dst_uniform) : "memory");
}
// the same for a run of lanes: lanes whose bit is set in `mask` (wave-uniform, never 0) store 8 bytes to
// row_uniform + voff.  One vector-memory operation on every path, whatever the mask.
__device__ __forceinline__ void store_pairs_masked(v2f* row_uniform, unsigned voff, v2f value, unsigned long long mask) {
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %4\n\t"
                 "global_store_dwordx2 %1, %2, %3\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(voff), "v"(value), "s"(row_uniform), "s"(mask) : "memory");
}
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_lane0(float4* dst_uniform, v4f value) {
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "global_store_dwordx4 %1, %2, %3" SERHIP_LANE0_NT "\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(0), "v"(value), "s"(dst_uniform) : "memory");
}

// Wait until at most kYounger vector-memory operations of this wave are outstanding.  No register operands on
// purpose: tying the loaded registers to this statement ("+v") made the register allocator copy them into fresh
// registers in FRONT of the wait -- a read of registers still in flight.  Instead nothing may be scheduled across
// the statement (sched_barrier), every use of the loaded pairs follows it in program order, and
// scripts/check_inflight_regs.py proves on the ISA that no instruction touches those registers before it.
template <int kYounger>
__device__ __forceinline__ void wait_samples() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kYounger) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------
// k_stft_frames
// ---------------------------------------------------------------------------------------
// kProj: any of mfcc / chroma / mel is wanted (the magnitudes are stored); kChroma: the piptrack peaks are.
// Compile-time so that a frame's store count is a constant (see "sample loads").
template <bool kProj, bool kChroma>
__global__ __launch_bounds__(kWaves * 64) void k_stft_frames(const float* __restrict__ audio,
                                                             const SegDesc* __restrict__ segs,
                                                             const WorkUnit* __restrict__ units,
                                                             int unit_frames, PlanTables tb, Scratch sc) {
    // The four tables are read from LDS once per frame.  They are kept as PAIRS of consecutive entries
    // (float4 {t[2j], t[2j+1]} per lane) so that every read is one ds_read_b128: 1 KiB per wave-instruction
    // in 4 LDS cycles.  The plain [entry][lane] v2f layout made the compiler fuse two 8-byte reads into
    // ds_read2st64_b64, which moves the same bytes in 8 cycles (MI355X_MICROARCH.md, LDS table).
    __shared__ __attribute__((aligned(16))) float4 s_twA4[8 * 64];
    __shared__ __attribute__((aligned(16))) float4 s_twU4[4 * 64];
    __shared__ __attribute__((aligned(16))) float4 s_win4[8 * 64];
    __shared__ __attribute__((aligned(16))) float4 s_twB4[8 * 4];
    __shared__ __attribute__((aligned(16))) v2f s_buf[kWaves][kBufC];
    __shared__ int s_next;   // next undrawn position in the concatenation of this workgroup's units

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the unit cursor below is scalar

    {   // entry e of lane l -> half (e & 1) of pair (e >> 1)
        v2f* a2 = reinterpret_cast<v2f*>(s_twA4);
        v2f* w2 = reinterpret_cast<v2f*>(s_win4);
        v2f* u2 = reinterpret_cast<v2f*>(s_twU4);
        v2f* b2 = reinterpret_cast<v2f*>(s_twB4);
        for (int i = tid; i < 16 * 64; i += kWaves * 64) {
            const int e = i >> 6, l = i & 63, at = ((e >> 1) * 64 + l) * 2 + (e & 1);
            a2[at] = tb.twA[i];
            w2[at] = tb.win[i];
        }
        for (int i = tid; i < 8 * 64; i += kWaves * 64) {
            const int e = i >> 6, l = i & 63;
            u2[((e >> 1) * 64 + l) * 2 + (e & 1)] = tb.twU[i];
        }
        if (tid < 64) {
            const int e = tid >> 2, l = tid & 3;   // twB is [16][4]
            b2[((e >> 1) * 4 + l) * 2 + (e & 1)] = tb.twB[tid];
        }
    }
    if (tid == 0) s_next = 0;
    __syncthreads();

    v2f* buf = s_buf[wave];
    float* fbuf = reinterpret_cast<float*>(buf);
    float* sM = fbuf;            // |X|, bins 0..1024, natural order

    const int q = lane & 3;
    const float sg1 = (q & 2) ? -1.0f : 1.0f;
    const float sg2 = (q & 1) ? -1.0f : 1.0f;
    const int j2 = ((q & 1) << 1) | (q >> 1);
    // stage-B row of this lane's quad.  Any bijection quad -> row works for the arithmetic; this one makes the
    // transposed read below conflict-free as ds_read_b128: that instruction serves the lane groups
    // {quads 0,3,5,6}, {1,2,4,7} (and the same + 8) in one LDS cycle each, a row starts 8 * row banks further
    // (row stride 68 complex = 136 dwords), and a quad covers 16 banks -- so the rows of a group must be
    // 2 apart: quads 0,3,5,6 -> rows 0,2,4,6 and quads 1,2,4,7 -> rows 1,3,5,7.
    const int k1b = (lane >> 2) ^ (((lane >> 3) ^ (lane >> 4)) & 1);
    // column of this lane in the stage-A rows: within every 8 columns (c = 8j + 4e + q -> 8j + 2q + e) the two
    // columns 4(2j) + q and 4(2j+1) + q that one stage-B lane needs sit side by side
    const int colA = (lane & ~7) | ((lane & 3) << 1) | ((lane >> 2) & 1);
    const double sr_d = static_cast<double>(tb.sample_rate);
    constexpr bool want_chroma = kChroma;
    constexpr bool want_proj = kProj;   // any of mfcc / chroma / mel needs the magnitudes

    // ---- cursor over this workgroup's units (wave-uniform, scalar registers).  `pos` is this wave's
    // position in the concatenation of the workgroup's units (drawn from s_next, so it only grows),
    // `base` the position of the current unit's first frame, `n` its frame count.
    const WorkUnit* list = units + gridDim.x + 1;
    int u = units[blockIdx.x].seg;             // every workgroup has at least one unit
    const int u_end = units[blockIdx.x + 1].seg;
    WorkUnit wu = list[u];
    SegDesc sd = segs[wu.seg];
    int n = min(unit_frames, sd.n_stft - wu.t_begin);
    int base = 0;
    // draw the next position; the value is read back (readfirstlane) only where it is needed, so the LDS
    // round trip of the atomic hides behind the frame in hand
    auto draw = [&]() {
        int p = 0;
        if (lane == 0) p = __hip_atomic_fetch_add(&s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return p;
    };
    int pos = __builtin_amdgcn_readfirstlane(draw());
    auto seek = [&]() -> bool {   // move to the unit that holds `pos`; false when the units are used up
        while (pos >= base + n) {
            base += n;
            if (++u >= u_end) return false;
            wu = list[u];
            sd = segs[wu.seg];
            n = min(unit_frames, sd.n_stft - wu.t_begin);
        }
        return true;
    };
    // ---- sample loads.  Frame t covers segment samples [t*512 - 1024, t*512 + 1024); lane l holds the 16 sample
    // pairs 128*n1 + 2*l; out-of-range samples are the zero padding of librosa.stft(center=True).
    // The 16 loads of the NEXT frame are issued as soon as the unpack of the frame in hand
    // has released the transform's registers, i.e. before that frame's stores; they are waited for at the top
    // of the next transform.  vmcnt counts loads and stores together, in order, and a store's acknowledgement
    // takes as long as a load from HBM: a compiler-placed wait (vmcnt(0) as soon as a conditional store sits
    // between load and use) made every frame wait for its predecessor's stores -- ~2 500 of a frame's 15 500
    // wave-cycles (in-kernel stamps).  Here every frame issues, AFTER its prefetch loads, a fixed number of
    // stores (finish_frame, "store block"), so `vmcnt(that number)` means: the loads have landed, the stores
    // may still be in flight.
    // Every frame takes exactly 16 global_load_dwordx2.  An interior frame reads sbase + 8*lane + 512*n1; a
    // frame that overlaps the zero padding of librosa.stft(center=True) reads pair-wise clamped addresses and
    // repairs the values after the wait (fix_padded_frame).
    auto load_frame_async = [&, lane_outer = lane](int t, v2f (&dst)[16]) {
        int lane = lane_outer;
        asm volatile("" : "+v"(lane));
        const int fs = t * kHop - kNfft / 2;   // < 2^31: the host rejects longer segments
        const float* seg_audio = audio + sd.offset;
        if ((fs >= 0) & (fs + kNfft <= sd.len)) {
            const float* b0 = seg_audio + fs;
            const float* b1 = b0 + 1024;       // immediate offsets reach 4 KiB
            const unsigned voff = 8u * static_cast<unsigned>(lane);
            load_pair_async<0>(dst[0], b0, voff);
            load_pair_async<512>(dst[1], b0, voff);
            load_pair_async<1024>(dst[2], b0, voff);
            load_pair_async<1536>(dst[3], b0, voff);
            load_pair_async<2048>(dst[4], b0, voff);
            load_pair_async<2560>(dst[5], b0, voff);
            load_pair_async<3072>(dst[6], b0, voff);
            load_pair_async<3584>(dst[7], b0, voff);
            load_pair_async<0>(dst[8], b1, voff);
            load_pair_async<512>(dst[9], b1, voff);
            load_pair_async<1024>(dst[10], b1, voff);
            load_pair_async<1536>(dst[11], b1, voff);
            load_pair_async<2048>(dst[12], b1, voff);
            load_pair_async<2560>(dst[13], b1, voff);
            load_pair_async<3072>(dst[14], b1, voff);
            load_pair_async<3584>(dst[15], b1, voff);
        } else {
            // pair (s, s+1), s = fs + 128 n1 + 2 lane, read from the clamped pair start sc = clamp(s, 0, len-2)
            // (always inside the segment: len >= 2048 here), relative to a wave-uniform base inside the segment
            const int b = min(max(fs, 0), sd.len - kNfft);
            const float* b0 = seg_audio + b;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const int sx = fs + 128 * n1 + 2 * lane;
                const unsigned voff = 4u * static_cast<unsigned>(min(max(sx, 0), sd.len - 2) - b);
                load_pair_async<0>(dst[n1], b0, voff);
            }
        }
    };
    // after the wait: pair n1 holds (x[sc], x[sc+1]); wanted is (x[s], x[s+1]) with zeros outside [0, len)
    auto fix_padded_frame = [&, lane_outer = lane](int t, v2f (&dst)[16]) {
        const int lane = lane_outer;
        const int fs = t * kHop - kNfft / 2;
        if ((fs >= 0) & (fs + kNfft <= sd.len)) return;
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int sx = fs + 128 * n1 + 2 * lane;
            const bool whole = sx >= 0 && sx <= sd.len - 2;          // then sc == s: both samples as loaded
            const bool last = sx == sd.len - 1;                      // sc = len - 2: x[s] is the second one
            dst[n1] = v2f{whole ? dst[n1].x : (last ? dst[n1].y : 0.0f), whole ? dst[n1].y : 0.0f};
        }
    };
    const v2f rot3 = (q == 3) ? v2f{0.0f, -1.0f} : v2f{1.0f, 0.0f};   // lane 3 of a quad turns by -i
    // bit r: bin lane + 64 r lies in the piptrack range [kmin, kmax] (one VGPR instead of a set of
    // hoisted scalar conditions per stripe, which spilled SGPRs)
    unsigned in_range = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (lane + 64 * r >= tb.kmin && lane + 64 * r <= tb.kmax) in_range |= 1u << r;
#if SERHIP_K1_STAMPS
    unsigned stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned stamp_frames = 0;
    unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long stamp_first = stamp_prev;
    const unsigned long long stamp_real_first = __builtin_amdgcn_s_memrealtime();
#endif
    // one frame, consumed from v (the raw sample pairs loaded by load_frame)
    // s_win holds HALF the Hann window: the FFT is linear and halving is exact, so the unpack below needs no
    // 0.5 factors.  The window is read from LDS BEFORE the wait for the samples (nothing is scheduled across that
    // wait, so the order in the source is the order in the code): its LDS latency runs under the vmcnt wait.
    auto read_window = [&](float4 (&w)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = s_win4[j * 64 + lane];
    };
    auto transform_frame = [&](v2f (&v)[16], const float4 (&w)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[2 * j] = v[2 * j] * v2f{w[j].x, w[j].y};
            v[2 * j + 1] = v[2 * j + 1] * v2f{w[j].z, w[j].w};
        }
        K1_STAMP(9)

        // ---- stage A: 16-pt DFT over n1 (registers), twiddle W_1024^(lane*k1), transpose
        fft16(v);
#pragma unroll
        for (int k1 = 0; k1 < 16; k1 += 2) {
            const float4 w = s_twA4[(k1 >> 1) * 64 + lane];
            if (k1 > 0) v[SERHIP_REV4(k1)] = cmul(v[SERHIP_REV4(k1)], v2f{w.x, w.y});
            v[SERHIP_REV4(k1 + 1)] = cmul(v[SERHIP_REV4(k1 + 1)], v2f{w.z, w.w});
        }
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) buf[k1 * 68 + colA] = v[SERHIP_REV4(k1)];
        wave_lds_sync();
        K1_STAMP(1)
        {
            const float4* b4 = reinterpret_cast<const float4*>(buf) + k1b * 34 + q;
#pragma unroll
            for (int j = 0; j < 8; ++j) {   // columns 4 m1 + q for m1 = 2j, 2j + 1: one 16-byte read
                const float4 t = b4[4 * j];
                v[2 * j] = v2f{t.x, t.y};
                v[2 * j + 1] = v2f{t.z, t.w};
            }
        }
        wave_lds_sync();
        __builtin_amdgcn_s_setprio(1);   // (frame_priority below)
        K1_STAMP(2)

        // ---- stage B: 16-pt DFT over m1, twiddle W_64^(m2*j1), radix-4 across the quad
        fft16(v);
        // two j1 chains are written interleaved: a DPP move needs two independent instructions between it
        // and the VALU write of its source, and the compiler pads with s_nop what the source order does
        // not give it (the kernel is bound by instruction issue: every s_nop costs a slot)
#pragma unroll
        for (int j1 = 0; j1 < 16; j1 += 2) {
            v2f xa = v[SERHIP_REV4(j1)], xb = v[SERHIP_REV4(j1 + 1)];
            const float4 wb = s_twB4[(j1 >> 1) * 4 + q];
            if (j1 > 0) xa = cmul(xa, v2f{wb.x, wb.y});
            xb = cmul(xb, v2f{wb.z, wb.w});
            const v2f pa = v2f{dpp_mov<0x4E>(xa.x), dpp_mov<0x4E>(xa.y)};   // lane ^ 2
            const v2f pb = v2f{dpp_mov<0x4E>(xb.x), dpp_mov<0x4E>(xb.y)};
            const v2f ta = pa + sg1 * xa;
            const v2f tb2 = pb + sg1 * xb;
            const v2f ua = cmul(ta, rot3);
            const v2f ub = cmul(tb2, rot3);
            const v2f qa = v2f{dpp_mov<0xB1>(ua.x), dpp_mov<0xB1>(ua.y)};   // lane ^ 1
            const v2f qb = v2f{dpp_mov<0xB1>(ub.x), dpp_mov<0xB1>(ub.y)};
            // Z[k], k = k1 + 16*j1 + 256*j2, stored at k + 4*(k>>8)
            buf[k1b + 16 * j1 + 260 * j2] = qa + sg2 * ua;
            buf[k1b + 16 * (j1 + 1) + 260 * j2] = qb + sg2 * ub;
        }
        wave_lds_sync();
        K1_STAMP(3)
    };
    // second half of a frame: Z (in this wave's LDS buffer) -> |X|, stores, peaks.  Needs no registers
    // of the transform, so the next frame's samples are requested before it starts.
    int run_seg = -1;        // segment of the frames whose largest magnitude run_max holds (see finish_frame)
    unsigned run_max = 0;
    auto finish_frame = [&, lane_outer = lane](size_t row, int seg, auto&& request_next) {
        const int lane = lane_outer;
        // the transform's registers are free from here on: the next frame's samples are requested now, ahead of the
        // unpack (1.2 % faster than behind it)
        request_next();
        // ---- real-FFT unpack: bins k = lane + 64r and 1024 - k from Z[k], Z[1024-k]
        float mlo[8], mhi[8];
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            const float4 wu = s_twU4[rp * 64 + lane];   // W_2048^(lane + 64 r) for r = 2 rp, 2 rp + 1
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = 2 * rp + h;
                const int k = lane + 64 * r;
                const int kk = (1024 - k) & 1023;
                const v2f za = buf[k + 4 * (r >> 2)];
                const v2f zb = buf[kk + 4 * (kk >> 8)];
                const v2f e2 = add_conj(za, zb);           // E  (the window is pre-halved)
                const v2f o2 = sub_conj_neg_i(za, zb);     // O
                const v2f t2 = cmul(o2, h ? v2f{wu.z, wu.w} : v2f{wu.x, wu.y});
                const v2f xlo = e2 + t2;
                const v2f xhi = e2 - t2;
                mlo[r] = __builtin_amdgcn_sqrtf(xlo.x * xlo.x + xlo.y * xlo.y);   // v_sqrt_f32, <= 1 ulp
                mhi[r] = __builtin_amdgcn_sqrtf(xhi.x * xhi.x + xhi.y * xhi.y);
            }
        }
        float m512 = 0.0f;
        if (lane == 0) {
            const v2f z = buf[512 + 8];
            m512 = 2.0f * __builtin_amdgcn_sqrtf(z.x * z.x + z.y * z.y);   // Z is half-scaled
        }
        wave_lds_sync();  // every lane has read Z before the buffer is reused
        K1_STAMP(4)

        // column maximum on the bit patterns: magnitudes are non-negative, so unsigned integer order is
        // float order and v_max3_u32 needs none of the NaN canonicalisation fmaxf drags in
        unsigned cmax_bits = __float_as_uint(m512);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int k = lane + 64 * r;
            sM[k] = mlo[r];
            sM[1024 - k] = mhi[r];
            cmax_bits = max(cmax_bits, max(__float_as_uint(mlo[r]), __float_as_uint(mhi[r])));
        }
        if (lane == 0) sM[512] = m512;
        wave_lds_sync();
        __builtin_amdgcn_s_setprio(3);
        K1_STAMP(5)

        // ---- the segment's largest magnitude bounds its mel power from above: k_project_frames uses it to tell
        // which dB rows can never be clipped by top_db and need not be stored; the frame's own is piptrack's ref
        if (want_proj || want_chroma) cmax_bits = wave_max_bits(cmax_bits);
        K1_STAMP(0)
        // ---- piptrack peaks (bins kmin..kmax), threshold 0.1 * column max.  Pass 1 finds the local
        // maxima and compacts their bin numbers into LDS; pass 2 gives each lane one peak, so the
        // double-precision interpolation runs once per 64 peaks instead of once per bin stripe.
        int base = 0;
        v2f first_peak = v2f{0.0f, 0.0f};
        v2f* prow = sc.peaks + row * sc.peak_cap;
        if (want_chroma) {
            const float cmax = __uint_as_float(cmax_bits);
            const float ref = 0.1f * cmax;
            int* cand = reinterpret_cast<int*>(fbuf + 1032);   // behind sM
            // librosa thresholds S below ref to 0 before localmax; with s0 > ref that is the same as
            // comparing the raw neighbours (a neighbour <= ref is < s0 either way)
            auto scan_stripe = [&](int r, float s0) {
                const int k = lane + 64 * r;
                const bool cand_ok = (s0 > ref) & (((in_range >> r) & 1u) != 0u);
                if (__ballot(cand_ok) == 0ull) return;   // wave-uniform: most stripes hold nothing above ref
                // both neighbours are read unconditionally and the tests combined without short-circuit
                // branches: one LDS round trip per stripe instead of two nested ones
                const float sm = sM[max(k - 1, 0)], sp = sM[k + 1];
                const bool is_peak = cand_ok & (s0 > sm) & (s0 >= sp);
                const unsigned long long mask = __ballot(is_peak);
                if (is_peak) cand[base + __popcll(mask & ((1ull << lane) - 1ull))] = k;
                base += __popcll(mask);
            };
#pragma unroll
            for (int r = 0; r < 8; ++r) scan_stripe(r, mlo[r]);
            for (int r = 8; 64 * r <= tb.kmax; ++r) scan_stripe(r, sM[lane + 64 * r]);   // sample rates < 16 kHz
            wave_lds_sync();
        K1_STAMP(6)
            // peaks 64, 65, ... of a frame (rare) are stored here, ahead of the store block
            for (int i = lane + 64; i < base; i += 64) {
                const int k = cand[i];
                prow[i] = piptrack_peak<true>(sM[k - 1], sM[k], sM[k + 1], k, sr_d, static_cast<double>(kNfft));
            }
            if (lane < base) {
                const int k = cand[lane];
                first_peak = piptrack_peak<true>(sM[k - 1], sM[k], sM[k + 1], k, sr_d, static_cast<double>(kNfft));
            }
        }
        if (want_proj) {
            // the segment's largest magnitude.  With chroma on it travels in the frame's peak-count word (store
            // block below) and k_segment_finish<1> reduces it.  Without chroma that kernel does not run: the
            // maximum is kept per wave over the consecutive frames it draws from one segment and sent as one
            // atomic per such run, ahead of the store block (an atomic is acknowledged late and vmcnt counts in
            // order, so each one delays the wave's next sample wait).
            if constexpr (!kChroma) {
                if (seg != run_seg) {
                    if (run_seg >= 0 && lane == 0) atomicMax(&sc.seg_cmax[run_seg], run_max);
                    run_seg = __builtin_amdgcn_readfirstlane(seg);
                    run_max = __builtin_amdgcn_readfirstlane(cmax_bits);
                } else {
                    run_max = __builtin_amdgcn_readfirstlane(max(run_max, cmax_bits));
                }
            }
        }
        // ---- store block: the LAST vector-memory operations of a frame, and always the same number of them
        // (kStoreBlock, by the kernel's template flags), so that the next frame can wait for its samples with an
        // exact count (see "sample loads").  No branch around any of them, so the count holds on every path
        // (scripts/check_inflight_regs.py): one-lane stores narrow EXEC inside their asm statement (store_lane0).
        if (want_chroma) {
            // only the lanes that hold a peak store (lane 0 always does, a zero pair into slot 0 of a frame without
            // peaks: nobody reads it, the count says 0): EXEC is set from the count inside the asm statement
            {
                const int n = __builtin_amdgcn_readfirstlane(min(max(base, 1), 64));
                const unsigned long long mask = n == 64 ? ~0ull : (1ull << n) - 1ull;
                unsigned slot = static_cast<unsigned>(lane);
                asm volatile("" : "+v"(slot));   // opaque: the byte offset is recomputed here, not kept in a register across the frame loop
                store_pairs_masked(prow, slot * 8u, lane < base ? first_peak : v2f{0.0f, 0.0f}, mask);
            }
            store_lane0(sc.peak_cnt + row, __builtin_amdgcn_readfirstlane(static_cast<int>(
                            ((cmax_bits + kPeakCountMask) & ~kPeakCountMask) | static_cast<unsigned>(base))));
        }
        if (want_proj) {
            // magnitudes for k_project_frames: a linear copy of sM, float4 chunk (lane + 64u) per lane
            // (conflict-free ds_read_b128; eight whole 128-byte lines per store instruction); bin 1024 is
            // stored with three zero pads so the last 16-bin group reads defined values
            const float4* mp = reinterpret_cast<const float4*>(sM);
            float4* m4 = reinterpret_cast<float4*>(sc.mags);
            const size_t mrow = row;
#pragma unroll
            // streaming (nt) stores: the rows are read once, by another kernel, long after they have left the L2;
            // store acknowledgements come back sooner and with them the in-order vmcnt of the next sample loads
            // (measured: kernel -3.7 %)
            for (int u = 0; u < 4; ++u) {
                const float4 val = mp[lane + 64 * u];
                __builtin_nontemporal_store(v4f{val.x, val.y, val.z, val.w}, reinterpret_cast<v4f*>(m4 + mags_tile_index(mrow, lane + 64 * u)));
            }
            store_lane0(m4 + mags_tile_index(mrow, 256), v4f{mhi[0], 0.0f, 0.0f, 0.0f});   // lane 0 holds bin 1024
        }
        wave_lds_sync();  // sM reads done before the next frame overwrites the buffer
        K1_STAMP(7)
    };
    // vector-memory operations in the store block of finish_frame
    constexpr int kStoreBlock = (kChroma ? 2 : 0) + (kProj ? 5 : 0);
    v2f v[16];
    bool more = seek();
    if (more) {
        load_frame_async(wu.t_begin + (pos - base), v);
        // the first frame has no predecessor whose store block would follow its loads: stand-ins, so that the
        // one wait below is right for every frame (the stored value is what the frame will store there anyway
        // or overwrite: its own peak count slot)
        {
            volatile int* slot = sc.peak_cnt + static_cast<size_t>(sd.stft_base) + static_cast<size_t>(wu.t_begin + (pos - base));
#pragma unroll
            for (int i = 0; i < kStoreBlock; ++i) *slot = 0;   // volatile: kStoreBlock separate store instructions
        }
    }
    // frame_priority: a wave raises its issue priority as it moves through a frame -- 0 for the window and stage A,
    // 1 from stage B, 2 from the unpack (the next frame's loads go out first thing there), 3 for the peak scan and the
    // store block, back to 0 for the next frame.  The four waves of a SIMD are at different points of their frames;
    // letting the one that is furthest along issue first gets its vector-memory operations (the longest latencies of
    // the frame) out earliest and frees its LDS buffer soonest.  Measured round robin in one call against the
    // arbiter's default (oldest wave first): 1.874 -> 1.820 ms per 12 500 clips (-2.9 %); the reverse grading
    // (young frames first) 1.91 ms; high priority for the transform alone +0.6 %, for the finish alone -0.9 %.
    while (more) {
        const int t = wu.t_begin + (pos - base);
        const size_t row = static_cast<size_t>(sd.stft_base) + static_cast<size_t>(t);
        const int drawn = draw();
        const int seg = wu.seg;
        float4 win[8];
        read_window(win);
        // the samples of this frame: everything but the store block that followed their loads has to be done
        wait_samples<kStoreBlock>();
        fix_padded_frame(t, v);
        transform_frame(v, win);
        __builtin_amdgcn_s_setprio(2);
        finish_frame(row, seg, [&]() {
            pos = __builtin_amdgcn_readfirstlane(drawn);
            more = seek();
            if (more) load_frame_async(wu.t_begin + (pos - base), v);
        });
        __builtin_amdgcn_s_setprio(0);
        K1_STAMP(8)
#if SERHIP_K1_STAMPS
        ++stamp_frames;
#endif
    }
    if (kProj && !kChroma && run_seg >= 0 && lane == 0) atomicMax(&sc.seg_cmax[run_seg], run_max);
#if SERHIP_K1_STAMPS
    if (lane == 0) {
        unsigned* dst = reinterpret_cast<unsigned*>(sc.mel) + (blockIdx.x * kWaves + wave) * 16;
        for (int i = 0; i < 10; ++i) dst[i] = stamp_acc[i];
        dst[10] = stamp_frames;
        dst[11] = static_cast<unsigned>(__builtin_amdgcn_s_memtime() - stamp_first);
        dst[12] = static_cast<unsigned>(__builtin_amdgcn_s_memrealtime() - stamp_real_first);   // 100 MHz ticks
        dst[13] = static_cast<unsigned>(stamp_real_first);
    }
#endif
}

// ---------------------------------------------------------------------------------------
// k_segment_finish helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned float_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// semitone residual of librosa.pitch_tuning in float32: mod(12*log2(f/27.5), 1), folded to
// [-0.5, 0.5).  log2 is evaluated in double and rounded once (correctly rounded f32 log2).
__device__ __forceinline__ float tuning_residual(float pitch) {
    const float qv = pitch / 27.5f;
    const float o = static_cast<float>(log2(static_cast<double>(qv)));
    const float v = 12.0f * o;
    float r = v - floorf(v);
    if (r >= 0.5f) r -= 1.0f;
    return r;
}

// histogram bin of np.histogram(residual, np.linspace(-0.5, 0.5, 101)): [e_i, e_i+1), last closed
__device__ __forceinline__ int tuning_bin(float r, const double* edges) {
    const double x = static_cast<double>(r);
    int i = static_cast<int>((x + 0.5) * 100.0);
    i = max(0, min(99, i));
    while (i > 0 && x < edges[i]) --i;
    while (i < 99 && x >= edges[i + 1]) ++i;
    return i;
}

// kMode 0: whole finish for one segment (generic path); 1: tuning estimate only; 2: reductions only
template <int kMode>
struct FinishShared {
    float fb[kMode == 0 ? kBins * kChroma + 4 : 4];   // chroma filterbank of the selected tuning, [bin][12]
    v2f peaks[kMode == 2 ? 1 : kPeakLdsCap];
    unsigned hist[kMode == 2 ? 1 : 2048];
    float red[64];
    int ired[8];
    double lmean[kMode == 1 ? 1 : kMels];          // the tuning-only kernel keeps none of the mel / chroma sums
    double melsum[2][kMode == 1 ? 1 : kMels];      // (25 KB instead of 30 KB; its time did not change: 0.20 ms)
    double lsum[2][kMode == 1 ? 1 : kMels];
    float chroma_acc[kMode == 1 ? 1 : 4][16];
    unsigned sel_prefix, sel_k;
    unsigned sel_cnt, sel_cum;   // pass 0 of select_kth: keys in the selected bucket / in the buckets below it
    double edges[kMode == 2 ? 1 : kTunings];   // histogram edges (a copy: two dependent reads per peak)
    int n_peaks;
    int tuning_idx;
    float thr;
};

// iterate over all peaks of the segment; F is called as f(pitch_mag)
template <typename Sh, typename F>
__device__ __forceinline__ void for_each_peak(const Sh& sh, bool in_lds, const Scratch& sc,
                                              const SegDesc& sd, int peak_cap, F f) {
    if (in_lds) {
        for (int i = threadIdx.x; i < sh.n_peaks; i += blockDim.x) f(sh.peaks[i]);
    } else {
        // Peaks that do not fit the LDS staging (noisy or long segments) are re-read from their rows in
        // every pass.  Wave w owns rows w, w + nw, ...: one strided load brings the counts of 64 of its
        // rows, then 4 rows x 4 chunks of 64 peaks are requested before the first one is consumed
        // (a row-by-row walk with its two dependent loads per row was ~6x slower on white noise).
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
        const int nw = blockDim.x >> 6;
        for (int tb = wave; tb < sd.n_stft; tb += 64 * nw) {
            const int my_t = tb + lane * nw;
            const int my_cnt = my_t < sd.n_stft ? static_cast<int>(static_cast<unsigned>(sc.peak_cnt[static_cast<size_t>(sd.stft_base) + my_t]) & kPeakCountMask) : 0;
            const int rows = min(64, (sd.n_stft - tb + nw - 1) / nw);
            for (int r0 = 0; r0 < rows; r0 += 4) {
                int c[4];
                const v2f* prow[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    // lanes past `rows` hold 0; r0 + u < 64 because rows <= 64 and r0 is a multiple of 4
                    c[u] = __builtin_amdgcn_readlane(my_cnt, r0 + u);
                    prow[u] = sc.peaks + (static_cast<size_t>(sd.stft_base) + tb + static_cast<size_t>(r0 + u) * nw) * peak_cap;
                }
                const int cmax = max(max(c[0], c[1]), max(c[2], c[3]));
                for (int j0 = 0; j0 < cmax; j0 += 256) {
                    v2f val[4][4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int j = j0 + 64 * k + lane;
                            if (j < c[u]) val[u][k] = prow[u][j];
                        }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int j = j0 + 64 * k + lane;
                            if (j < c[u]) f(val[u][k]);
                        }
                }
            }
        }
    }
}

// k-th smallest peak magnitude (0-based) by radix select on the order-preserving key, 11 + 11 + 10
// bits (wide digits keep the LDS histogram atomics spread: peak magnitudes share few exponents).
// All threads return it.  blockDim.x must be 256.
//
// Peaks that did not fit the LDS staging (in_lds false) are read from their rows in HBM.  After the
// first digit only the keys of ONE bucket still matter: if that bucket fits, it is compacted into
// sh.peaks by one more pass over the rows and the remaining digits (and below_threshold) run from LDS
// -- three passes over HBM per segment (first digit, compaction, final histogram) instead of five.
// Then *subset is set, sh.n_peaks is the bucket's size, and *lower_cnt / *lower_max describe the keys
// of the lower buckets (their number, their largest magnitude).
template <typename Sh>
__device__ float select_kth(Sh& sh, bool in_lds, const Scratch& sc, const SegDesc& sd,
                            int peak_cap, unsigned k, bool* subset, unsigned* lower_cnt, float* lower_max) {
    unsigned prefix = 0, mask = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    *subset = false;
    *lower_cnt = 0;
    *lower_max = -INFINITY;
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
        const unsigned digit_mask = pass == 2 ? 1023u : 2047u;
        for (int i = tid; i < 2048; i += 256) sh.hist[i] = 0;
        __syncthreads();
        for_each_peak(sh, in_lds, sc, sd, peak_cap, [&](v2f pm) {
            const unsigned key = float_key(pm.y);
            if ((key & mask) == prefix) atomicAdd(&sh.hist[(key >> shift) & digit_mask], 1u);
        });
        __syncthreads();
        // thread t owns buckets 8t..8t+7; block-wide inclusive scan of the per-thread totals
        unsigned c[8];
        unsigned mine = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c[j] = sh.hist[8 * tid + j];
            mine += c[j];
        }
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) sh.ired[wave] = static_cast<int>(incl);
        __syncthreads();
        unsigned before = 0;
        for (int w = 0; w < wave; ++w) before += static_cast<unsigned>(sh.ired[w]);
        incl += before;
        const unsigned excl = incl - mine;
        if (excl <= k && k < incl) {   // exactly one thread
            unsigned cum = excl;
            int b = 7;
            bool found = false;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (!found) {
                    if (cum + c[j] > k) {   // first bucket whose cumulative count exceeds k
                        found = true;
                        b = j;
                    } else {
                        cum += c[j];
                    }
                }
            }
            sh.sel_prefix = prefix | (static_cast<unsigned>(8 * tid + b) << shift);
            sh.sel_k = k - cum;
            sh.sel_cnt = c[b];
            sh.sel_cum = cum;
        }
        __syncthreads();
        prefix = sh.sel_prefix;
        k = sh.sel_k;
        mask |= digit_mask << shift;
        if (pass == 0 && !in_lds && sh.sel_cnt <= static_cast<unsigned>(kPeakLdsCap)) {
            *lower_cnt = sh.sel_cum;
            __syncthreads();   // everyone has read sel_cnt / sel_cum
            if (tid == 0) sh.n_peaks = 0;
            __syncthreads();
            float lmax = -INFINITY;
            for_each_peak(sh, false, sc, sd, peak_cap, [&](v2f pm) {
                const unsigned top = float_key(pm.y) & mask;
                if (top == prefix) {
                    sh.peaks[atomicAdd(&sh.n_peaks, 1)] = pm;
                } else if (top < prefix) {
                    lmax = fmaxf(lmax, pm.y);
                }
            });
            lmax = wave_max(lmax);
            if (lane == 0) sh.red[wave] = lmax;
            __syncthreads();
            *lower_max = fmaxf(fmaxf(sh.red[0], sh.red[1]), fmaxf(sh.red[2], sh.red[3]));
            *subset = true;
            in_lds = true;
            __syncthreads();   // sh.red free again
        }
    }
    return key_float(prefix);
}

// largest peak magnitude strictly below thr and how many there are (for the lower middle of an even count)
template <typename Sh>
__device__ void below_threshold(Sh& sh, bool in_lds, const Scratch& sc, const SegDesc& sd, int peak_cap,
                                float thr, int* count, float* largest) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int n = 0;
    float m = -INFINITY;
    for_each_peak(sh, in_lds, sc, sd, peak_cap, [&](v2f pm) {
        if (pm.y < thr) {
            ++n;
            m = fmaxf(m, pm.y);
        }
    });
    n = wave_sum_int(n);
    m = wave_max(m);
    __syncthreads();   // sh.ired / sh.red free
    if (lane == 0) {
        sh.ired[wave] = n;
        sh.red[wave] = m;
    }
    __syncthreads();
    *count = sh.ired[0] + sh.ired[1] + sh.ired[2] + sh.ired[3];
    *largest = fmaxf(fmaxf(sh.red[0], sh.red[1]), fmaxf(sh.red[2], sh.red[3]));
    __syncthreads();
}

// on-the-fly chroma column for the generic path lives in kernels_generic.hpp
__device__ void chroma_column_dev(int k, int n_fft, int sr, double tuning, float* out12);

// ---------------------------------------------------------------------------------------
// k_segment_finish: one workgroup (256 threads) per segment
//   <0> generic path (segments < 2048 samples): mel/MFCC, tuning, on-device filterbank, chroma
//   <1> fast path, before k_project_frames: tuning estimate -> sc.tuning_idx[segment]
//   <2> fast path, after  k_project_frames: mel/MFCC reductions + time mean of the per-frame chroma
// ---------------------------------------------------------------------------------------
template <int kMode>
__global__ __launch_bounds__(256, 2) void k_segment_finish(const SegDesc* __restrict__ segs, PlanTables tb,
                                                        Scratch sc, float* __restrict__ out,
                                                        float* __restrict__ tuning_out, int out_dim) {
    __shared__ FinishShared<kMode> sh;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const SegDesc sd = segs[blockIdx.x];
    float* orow = out + sd.out_row * out_dim;
    const bool want_mfcc = tb.flags & 1u, want_chroma = tb.flags & 2u, want_mel = tb.flags & 4u,
               want_contrast = tb.flags & 8u;
    const int off_mfcc = 0;
    const int off_chroma = off_mfcc + (want_mfcc ? kMfcc : 0);
    const int off_mel = off_chroma + (want_chroma ? kChroma : 0);
    const int off_contrast = off_mel + (want_mel ? kMels : 0);
    const int nbins = 1 + sd.n_fft / 2;
    // kMode 2: this thread's first six values of the chroma time mean (a 3 s clip has six per thread) are requested now,
    // so that their latency passes under the mel / MFCC reductions instead of after them
    float chroma_pre[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (kMode == 2) {
        if (want_chroma) {
            const float* cn = sc.chroma_norm + static_cast<size_t>(sd.stft_base) * 16;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int t = (tid >> 4) + 16 * u;
                if (t < sd.n_stft) chroma_pre[u] = cn[static_cast<size_t>(t) * 16 + (tid & 15)];
            }
        }
    }
    if constexpr (kMode == 1) {
        // k_tuning_waves ran first and took every segment whose peaks fit one wave's staging: what it left
        // (index -1: more than kTwPeakCap peaks or kTwFrameCap frames) is this kernel's
        if (want_chroma && (tb.flags & kFlagTuningWavesRan) && sc.tuning_idx[blockIdx.x] >= 0) return;
    }

    // ---- mel mean and MFCC (power_to_db, top_db=80 vs the segment max, DCT of the time mean)
    if (kMode != 1 && (want_mfcc || want_mel)) {
        const int m = tid & 127, half = tid >> 7;
        const float* mel = sc.mel + static_cast<size_t>(sd.mel_base) * kMels;
        // time sums are accumulated in double (the reference's float32 pairwise sums are within
        // rounding of the exact value; a float32 running sum here would not be)
        double s = 0.0;
        float lmax = -INFINITY;
        // kMode 2: the dB minimum and dB sum of this thread's first four tiles (a 3 s clip has three per thread) are
        // requested here, together with the power sums, instead of behind the floor and behind each other
        float pre_min[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        double pre_sum[4] = {0.0, 0.0, 0.0, 0.0};
        if constexpr (kMode == 2) {
            // k_project_frames left dB values in `mel`, 16-frame power sums and per-tile dB maxima
            const int n_tiles = (sd.n_mel + 15) >> 4;
            const size_t tile0 = static_cast<size_t>(sd.mel_base) >> 4;
            if (want_mfcc) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int tl = half + 2 * u;
                    if (tl < n_tiles) {
                        pre_min[u] = sc.mel_dbmin[(tile0 + tl) * kMels + m];
                        pre_sum[u] = sc.mel_dbsum[(tile0 + tl) * kMels + m];
                    }
                }
            }
            for (int tl = half; tl < n_tiles; tl += 2) s += static_cast<double>(sc.mel_part[(tile0 + tl) * kMels + m]);
            for (int j = tid; j < n_tiles * 8; j += 256) lmax = fmaxf(lmax, sc.mel_dbmax[tile0 * 8 + j]);
        } else {
            for (int t = half; t < sd.n_mel; t += 2) {
                const float v = mel[static_cast<size_t>(t) * kMels + m];
                s += static_cast<double>(v);
                lmax = fmaxf(lmax, 10.0f * log10f(fmaxf(1e-10f, v)));
            }
        }
        sh.melsum[half][m] = s;
        lmax = wave_max(lmax);
        if (lane == 0) sh.red[wave] = lmax;
        __syncthreads();
        lmax = fmaxf(fmaxf(sh.red[0], sh.red[1]), fmaxf(sh.red[2], sh.red[3]));
        const float floor_db = lmax - 80.0f;
        if constexpr (kMode == 2) {
            // sum over time of max(dB, floor).  k_project_frames left, per 16-frame tile and filter, the sum
            // and the minimum of the dB values: a tile whose minimum reaches the floor contributes its sum
            // as it is (nothing in it is clipped), and only the others -- silence, fades -- are re-read
            // from the dB rows.  In the common case the rows are written but never read back.
            double ls = 0.0;
            if (want_mfcc) {
                const int n_tiles = (sd.n_mel + 15) >> 4;
                const size_t tile0 = static_cast<size_t>(sd.mel_base) >> 4;
                auto tile_sum = [&](int tl, float dbmin, double dbsum) {
                    if (dbmin >= floor_db) return dbsum;
                    double part = 0.0;
                    const int t_end = min(16 * tl + 16, sd.n_mel);
                    for (int t = 16 * tl; t < t_end; ++t)
                        part += static_cast<double>(fmaxf(mel[static_cast<size_t>(t) * kMels + m], floor_db));
                    return part;
                };
#pragma unroll
                for (int u = 0; u < 4; ++u) {          // the prefetched tiles, in the same order as before
                    const int tl = half + 2 * u;
                    if (tl < n_tiles) ls += tile_sum(tl, pre_min[u], pre_sum[u]);
                }
                for (int tl = half + 8; tl < n_tiles; tl += 2) {
                    const size_t at = (tile0 + tl) * kMels + m;
                    ls += tile_sum(tl, sc.mel_dbmin[at], sc.mel_dbsum[at]);
                }
            }
            sh.lsum[half][m] = ls;
        } else {
            double ls = 0.0;
            if (want_mfcc) {
                for (int t = half; t < sd.n_mel; t += 2) {
                    const float v = mel[static_cast<size_t>(t) * kMels + m];
                    ls += static_cast<double>(fmaxf(10.0f * log10f(fmaxf(1e-10f, v)), floor_db));
                }
            }
            sh.lsum[half][m] = ls;
        }
        __syncthreads();
        const double inv_t = 1.0 / static_cast<double>(sd.n_mel);
        if (tid < kMels) {
            const double lsum = sh.lsum[0][tid] + sh.lsum[1][tid];
            sh.lmean[tid] = lsum * inv_t;
            if (want_mel) orow[off_mel + tid] = static_cast<float>((sh.melsum[0][tid] + sh.melsum[1][tid]) * inv_t);
        }
        __syncthreads();
        if constexpr (kMode == 2) {
            // 40 x 128 DCT on all four waves: thread (coefficient, quarter of the mel axis)
            const int k = tid & 63, quarter = tid >> 6;
            double acc = 0.0;
            if (want_mfcc && k < kMfcc) {
                const float* d = tb.dct + k * kMels + 32 * quarter;
#pragma unroll 8
                for (int j = 0; j < 32; ++j) acc = fma(static_cast<double>(d[j]), sh.lmean[32 * quarter + j], acc);
            }
            double* partial = &sh.melsum[0][0];   // 256 doubles; melsum itself was consumed above
            partial[tid] = acc;
            __syncthreads();
            if (want_mfcc && tid < kMfcc)
                orow[off_mfcc + tid] = static_cast<float>((partial[tid] + partial[64 + tid]) +
                                                          (partial[128 + tid] + partial[192 + tid]));
        } else if (want_mfcc && tid < kMfcc) {
            double acc = 0.0;
            const float* d = tb.dct + tid * kMels;
            for (int j = 0; j < kMels; ++j) acc = fma(static_cast<double>(d[j]), sh.lmean[j], acc);
            orow[off_mfcc + tid] = static_cast<float>(acc);
        }
    }
    if (kMode != 1 && want_contrast && tid < 7) orow[off_contrast + tid] = 0.0f;
    if (!want_chroma) {
        if (kMode != 2 && tuning_out && tid == 0) tuning_out[sd.out_row] = 0.0f;
        return;
    }
    if constexpr (kMode == 2) {
        // time mean of the per-frame normalised chroma written by k_project_frames:
        // thread (c = tid & 15, part = tid >> 4) sums frames part, part + 16, ...
        const int c = tid & 15, part = tid >> 4;
        const float* cn = sc.chroma_norm + static_cast<size_t>(sd.stft_base) * 16;
        float acc = 0.0f;
#pragma unroll
        for (int u = 0; u < 6; ++u)                                   // requested at the top of the kernel
            if (part + 16 * u < sd.n_stft) acc += chroma_pre[u];
        for (int t = part + 96; t < sd.n_stft; t += 16) acc += cn[static_cast<size_t>(t) * 16 + c];
        __syncthreads();
        float* slots = reinterpret_cast<float*>(sh.melsum);   // 256 floats fit in melsum (2 x 128 doubles)
        slots[part * 16 + c] = acc;
        __syncthreads();
        if (tid < kChroma) {
            float s = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) s += slots[q * 16 + tid];
            orow[off_chroma + tid] = s / static_cast<float>(sd.n_stft);
        }
        return;
    }

    // ---- tuning estimate: median of peak magnitudes, histogram of residuals above it
    if (tid == 0) sh.n_peaks = 0;
    if constexpr (kMode != 2) {
        for (int i = tid; i < kTunings; i += blockDim.x) sh.edges[i] = tb.edges[i];
    }
    __syncthreads();
    // per-frame counts: loaded in parallel; for segments of <= 1024 frames they are kept (sh.hist is
    // free until the select) so that the gather below needs no dependent global loads
    const bool few_frames = sd.n_stft <= 1024;
    int total = 0;
    unsigned frame_max = 0;   // largest frame-max key of the segment (kPeakCountMask)
    for (int t = tid; t < sd.n_stft; t += blockDim.x) {
        const unsigned word = static_cast<unsigned>(sc.peak_cnt[static_cast<size_t>(sd.stft_base) + t]);
        const int c = static_cast<int>(word & kPeakCountMask);
        frame_max = max(frame_max, word & ~kPeakCountMask);
        total += c;
        if (few_frames) sh.hist[t] = static_cast<unsigned>(c);
    }
    total = wave_sum_int(total);
    frame_max = wave_max_bits(frame_max);
    if (lane == 0) {
        sh.ired[wave] = total;
        sh.ired[4 + wave] = static_cast<int>(frame_max);
    }
    __syncthreads();
    const int n_peaks = sh.ired[0] + sh.ired[1] + sh.ired[2] + sh.ired[3];
    if constexpr (kMode == 1) {
        if (tid == 0)
            sc.seg_cmax[blockIdx.x] = max(max(static_cast<unsigned>(sh.ired[4]), static_cast<unsigned>(sh.ired[5])),
                                          max(static_cast<unsigned>(sh.ired[6]), static_cast<unsigned>(sh.ired[7])));
    }
    __syncthreads();
    const bool in_lds = n_peaks <= kPeakLdsCap;
    if (in_lds && few_frames) {
        // exclusive scan of the counts (thread t owns frames 4t..4t+3) -> sh.hist[1024 + frame]
        unsigned c4[4], mine = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = 4 * tid + j;
            c4[j] = idx < sd.n_stft ? sh.hist[idx] : 0u;
            mine += c4[j];
        }
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) sh.ired[wave] = static_cast<int>(incl);
        __syncthreads();
        unsigned excl = incl - mine;
        for (int w = 0; w < wave; ++w) excl += static_cast<unsigned>(sh.ired[w]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = 4 * tid + j;
            if (idx < sd.n_stft) sh.hist[1024 + idx] = excl;
            excl += c4[j];
        }
        __syncthreads();
        // gather: 16 lanes per frame, 4 frames in flight per lane group, all loads independent
        const int g = tid >> 4, l16 = tid & 15;
        for (int t0 = g; t0 < sd.n_stft; t0 += 64) {
            v2f val[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + 16 * u;
                dst[u] = -1;
                if (t < sd.n_stft && l16 < static_cast<int>(sh.hist[t])) {
                    val[u] = sc.peaks[(static_cast<size_t>(sd.stft_base) + t) * sc.peak_cap + l16];
                    dst[u] = static_cast<int>(sh.hist[1024 + t]) + l16;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst[u] >= 0) sh.peaks[dst[u]] = val[u];
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // frames with more than 16 peaks
                const int t = t0 + 16 * u;
                if (t < sd.n_stft) {
                    const int c = static_cast<int>(sh.hist[t]), o = static_cast<int>(sh.hist[1024 + t]);
                    for (int j = l16 + 16; j < c; j += 16)
                        sh.peaks[o + j] = sc.peaks[(static_cast<size_t>(sd.stft_base) + t) * sc.peak_cap + j];
                }
            }
        }
        if (tid == 0) sh.n_peaks = n_peaks;
        __syncthreads();
    } else if (in_lds) {
        for (int t = wave; t < sd.n_stft; t += 4) {
            const size_t row = static_cast<size_t>(sd.stft_base) + t;
            const int c = static_cast<int>(static_cast<unsigned>(sc.peak_cnt[row]) & kPeakCountMask);
            if (c == 0) continue;
            int base = 0;
            if (lane == 0) base = atomicAdd(&sh.n_peaks, c);
            base = __shfl(base, 0);
            for (int j = lane; j < c; j += 64) sh.peaks[base + j] = sc.peaks[row * sc.peak_cap + j];
        }
        __syncthreads();
    } else if (tid == 0) {
        sh.n_peaks = n_peaks;
    }
    int tuning_idx = 50;  // edge 50 is exactly 0.0: librosa's answer for an empty peak set
    if (n_peaks > 0) {
        bool subset;
        unsigned lower_cnt;
        float lower_max;
        float thr = select_kth(sh, in_lds, sc, sd, sc.peak_cap, static_cast<unsigned>(n_peaks / 2), &subset,
                               &lower_cnt, &lower_max);
        if ((n_peaks & 1) == 0) {
            // the element of rank n/2 - 1: the largest value below thr if exactly n/2 values are below it,
            // else thr itself (a tie).  With the selected bucket compacted into LDS, the keys of the
            // lower buckets come in through their count and their maximum.
            int n_below;
            float below;
            below_threshold(sh, in_lds || subset, sc, sd, sc.peak_cap, thr, &n_below, &below);
            n_below += static_cast<int>(lower_cnt);
            below = fmaxf(below, lower_max);
            const float lo = (n_below == n_peaks / 2) ? below : thr;
            thr = (lo + thr) / 2.0f;  // np.median of an even count: f32 mean of the middle pair
        }
        for (int i = tid; i < 256; i += blockDim.x) sh.hist[i] = 0;
        __syncthreads();
        for_each_peak(sh, in_lds, sc, sd, sc.peak_cap, [&](v2f pm) {
            if (pm.y >= thr && pm.x > 0.0f) atomicAdd(&sh.hist[tuning_bin(tuning_residual(pm.x), sh.edges)], 1u);
        });
        __syncthreads();
        if (wave == 0) {
            // np.argmax: the first of the largest counts.  One wave, two bins per lane; the key orders by
            // count, then by smaller bin number (counts are far below 2^24)
            unsigned key = 0;
            if (lane < 100) key = (sh.hist[lane] << 8) | static_cast<unsigned>(255 - lane);
            if (lane + 64 < 100) key = max(key, (sh.hist[lane + 64] << 8) | static_cast<unsigned>(255 - (lane + 64)));
            key = wave_max_bits(key);
            if (lane == 0) sh.tuning_idx = 255 - static_cast<int>(key & 255u);
        }
        __syncthreads();
        tuning_idx = sh.tuning_idx;
    }
    if (tuning_out && tid == 0) tuning_out[sd.out_row] = static_cast<float>(tb.edges[tuning_idx]);
    if constexpr (kMode == 1) {
        if (tid == 0) sc.tuning_idx[blockIdx.x] = tuning_idx;
        return;
    }

    if constexpr (kMode == 0) {
    // ---- chroma filterbank of that tuning, built on the device -> LDS, bin-major [k][12]
    {
        const double tuning = tb.edges[tuning_idx];
        for (int k = tid; k < nbins; k += blockDim.x) {
            float col[12];
            chroma_column_dev(k, sd.n_fft, tb.sample_rate, tuning, col);
#pragma unroll
            for (int c = 0; c < 12; ++c) sh.fb[k * kChroma + c] = col[c];
        }
    }
    __syncthreads();

    // ---- chroma = fb @ |STFT|, per-frame inf-norm, time mean.
    // A wave takes 4 frames per pass: lane l holds bins l, l+64, ... of each frame, every
    // filterbank weight read from LDS is applied to the 4 frames, and the 4 x 16 partial sums
    // are reduced with a halving butterfly (63 shuffles) that leaves lane l with the complete
    // sum of (frame l>>4, chroma class l&15).
    const float tiny = 1.17549435e-38f;
    const int fslot = lane >> 4;
    float acc_c = 0.0f;
    for (int t0 = 4 * wave; t0 < sd.n_stft; t0 += 16) {
        const float* mrow[4];
#pragma unroll
        for (int f = 0; f < 4; ++f)
            mrow[f] = sc.mags + (static_cast<size_t>(sd.stft_base) + min(t0 + f, sd.n_stft - 1)) * kBinStride;
        float v[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = 0.0f;
        {
            for (int k = lane; k < nbins; k += 64) {
                float x[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) x[f] = mrow[f][k];
                const float4* wp = reinterpret_cast<const float4*>(sh.fb + k * kChroma);
                const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2];
                const float w[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int c = 0; c < 12; ++c) v[f * 16 + c] = fmaf(w[c], x[f], v[f * 16 + c]);
            }
        }
        // halving butterfly: after the stage with offset o, lane keeps the half selected by (lane & o)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const bool up = (lane & o) != 0;
#pragma unroll
            for (int i = 0; i < o; ++i) {
                const float keep = up ? v[i + o] : v[i];
                const float send = up ? v[i] : v[i + o];
                v[i] = keep + __shfl_xor(send, o);
            }
        }
        // v[0] = raw chroma of (frame t0 + fslot, class lane & 15); classes 12..15 are zero pads
        float raw = v[0];
        float mx = fabsf(raw);
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (mx < tiny) mx = 1.0f;
        if (t0 + fslot < sd.n_stft) acc_c += raw / mx;
    }
    // combine the 4 frame slots of the wave, then the waves
    acc_c += __shfl_xor(acc_c, 16);
    acc_c += __shfl_xor(acc_c, 32);
    if (lane < 16) sh.chroma_acc[wave][lane] = acc_c;
    __syncthreads();
    if (tid < kChroma) {
        const float s = (sh.chroma_acc[0][tid] + sh.chroma_acc[1][tid]) + (sh.chroma_acc[2][tid] + sh.chroma_acc[3][tid]);
        orow[off_chroma + tid] = s / static_cast<float>(sd.n_stft);
    }
    }  // kMode == 0
}

// ---------------------------------------------------------------------------------------
// k_tuning_waves: the tuning estimate of k_segment_finish<1> with ONE WAVE per segment (four segments per
// workgroup) for the common case -- at most kTwPeakCap peaks in at most kTwFrameCap frames.  A wave needs no
// workgroup barrier: its LDS staging is its own, counts are prefix-summed with wave scans, the median is found
// by a 32-step bitwise search whose counts are ballots (one v_cmp + one s_bcnt1 per 64 keys), the histogram is
// the same 100-bin LDS histogram.  Same discrete decisions as k_segment_finish<1> (order statistics and counts
// do not depend on the order of the keys), so the same tuning index.  Writes sc.tuning_idx[seg] (or -1: left
// to k_segment_finish<1>, which runs next and skips the segments done here), sc.seg_cmax[seg], tuning_out.
// ---------------------------------------------------------------------------------------
constexpr int kTwWaves = 4;
// Calls with fewer segments than this go to k_segment_finish<1> alone: one wave working through a segment takes
// twice as long as four waves sharing it (25 against 13 us), which is what a one-recording call waits for, and
// only a batch that fills the chip several times over is bound by throughput instead.
constexpr int kTwMinSegments = 2048;
constexpr int kTwPeakCap = 1024;
constexpr int kTwFrameCap = 1024;
constexpr int kTwKeysPerLane = kTwPeakCap / 64;
constexpr int kTwTopBits = 13;         // steps of the median search that run over all keys (sign, exponent, 4 mantissa bits)

struct TuningWaveShared {
    double edges[kTunings];
    unsigned keys[kTwWaves][kTwPeakCap];
    float pitch[kTwWaves][kTwPeakCap];
    unsigned hist[kTwWaves][128];
    int cnt[kTwWaves][64];
    int off[kTwWaves][64];
};

__global__ __launch_bounds__(kTwWaves * 64) void k_tuning_waves(const SegDesc* __restrict__ segs, int n_seg, PlanTables tb,
                                                                Scratch sc, float* __restrict__ tuning_out) {
    __shared__ TuningWaveShared sh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kTunings; i += kTwWaves * 64) sh.edges[i] = tb.edges[i];
    __syncthreads();                       // the only workgroup barrier: everything below is per wave
    const int seg = blockIdx.x * kTwWaves + wave;
    if (seg >= n_seg) return;
    const SegDesc sd = segs[seg];
    const size_t base = static_cast<size_t>(sd.stft_base);
    if (sd.n_stft > kTwFrameCap) {
        if (lane == 0) sc.tuning_idx[seg] = -1;
        return;
    }
    // ---- counts of all frames: total and the segment's frame-max key
    int total = 0;
    unsigned frame_max = 0;
    for (int t = lane; t < sd.n_stft; t += 64) {
        const unsigned word = static_cast<unsigned>(sc.peak_cnt[base + t]);
        total += static_cast<int>(word & kPeakCountMask);
        frame_max = max(frame_max, word & ~kPeakCountMask);
    }
    total = wave_sum_int(total);
    if (total > kTwPeakCap) {
        if (lane == 0) sc.tuning_idx[seg] = -1;
        return;
    }
    frame_max = wave_max_bits(frame_max);
    if (lane == 0) sc.seg_cmax[seg] = frame_max;
    const int n = total;
    int tuning_idx = 50;   // edge 50 is exactly 0.0: librosa's answer for an empty peak set
    if (n > 0) {
        unsigned* keys = sh.keys[wave];
        float* pitch = sh.pitch[wave];
        // ---- gather: 64 frames per round (lane = frame for the scan), then 16 lanes per frame
        int placed = 0;
        for (int t0 = 0; t0 < sd.n_stft; t0 += 64) {
            const int t = t0 + lane;
            const int c = t < sd.n_stft ? static_cast<int>(static_cast<unsigned>(sc.peak_cnt[base + t]) & kPeakCountMask) : 0;
            int incl = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            sh.cnt[wave][lane] = c;
            sh.off[wave][lane] = placed + incl - c;
            wave_lds_sync();
            const int in_round = min(64, sd.n_stft - t0);
            const int g = lane >> 4, l16 = lane & 15;
            // eight frames in flight per 16-lane group, all loads independent (a frame-by-frame walk is a chain of
            // dependent global round trips: 143 us per 12 500 segments against 9x us this way)
            for (int f0 = g; f0 < in_round; f0 += 32) {
                v2f val[8];
                int dst[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int f = f0 + 4 * u;
                    dst[u] = -1;
                    if (f < in_round && l16 < sh.cnt[wave][f]) {
                        val[u] = sc.peaks[(base + t0 + f) * sc.peak_cap + l16];
                        dst[u] = sh.off[wave][f] + l16;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (dst[u] >= 0) {
                        keys[dst[u]] = float_key(val[u].y);
                        pitch[dst[u]] = val[u].x;
                    }
#pragma unroll
                for (int u = 0; u < 8; ++u) {   // frames with more than 16 peaks
                    const int f = f0 + 4 * u;
                    if (f < in_round) {
                        const int cf = sh.cnt[wave][f], of = sh.off[wave][f];
                        const v2f* row = sc.peaks + (base + t0 + f) * sc.peak_cap;
                        for (int j = l16 + 16; j < cf; j += 16) {
                            const v2f pm = row[j];
                            keys[of + j] = float_key(pm.y);
                            pitch[of + j] = pm.x;
                        }
                    }
                }
            }
            placed += __builtin_amdgcn_readlane(incl, 63);
            wave_lds_sync();               // cnt / off are rewritten by the next round
        }
        // ---- my keys (lane owns slots lane, lane + 64, ...); slots past n hold a key no threshold exceeds
        const int per_lane = (n + 63) >> 6;
        unsigned k[kTwKeysPerLane];
#pragma unroll
        for (int i = 0; i < kTwKeysPerLane; ++i) {
            const int idx = lane + 64 * i;
            k[i] = (i < per_lane && idx < n) ? keys[idx] : 0xFFFFFFFFu;
        }
        auto count_below = [&](unsigned t) {
            int c = 0;
#pragma unroll
            for (int i = 0; i < kTwKeysPerLane; ++i)
                if (i < per_lane) c += __builtin_popcountll(__ballot(k[i] < t));
            return c;
        };
        // ---- sorted[n / 2]: the largest v with at most n / 2 keys strictly below it, built bit by bit
        // The first kTwTopBits steps run over all keys; the keys that still share the prefix then (a slice of an
        // octave: a few dozen of ~600) are compacted to one per lane and the remaining steps cost one compare each
        // (the search was a third of this kernel: 42 us per 12 500 segments in a timing-only ablation).
        const int kth = n >> 1;
        unsigned v = 0;
        int bit = 31;
        for (; bit > 31 - kTwTopBits; --bit) {
            const unsigned t = v | (1u << bit);
            if (count_below(t) <= kth) v = t;
        }
        {
            const unsigned span = 1u << (bit + 1);                 // keys with this prefix lie in [v, v + span)
            const int c_lo = count_below(v);
            const bool wraps = v > 0xFFFFFFFFu - (span - 1u);      // v + span would pass 2^32: the prefix is all ones
            const int c_hi = wraps ? n : count_below(v + span);    // (the slots past n hold 0xFFFFFFFF and are never below)
            const int m = c_hi - c_lo;
            if (m <= 64) {
                unsigned* sub = sh.hist[wave];                     // 128 slots, free until the histogram
                int placed_keys = 0;
#pragma unroll
                for (int i = 0; i < kTwKeysPerLane; ++i) {
                    if (i < per_lane) {
                        const bool mine = lane + 64 * i < n && k[i] >= v && (wraps || k[i] < v + span);
                        const unsigned long long mask = __ballot(mine);
                        if (mine) sub[placed_keys + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = k[i];
                        placed_keys += __builtin_popcountll(mask);
                    }
                }
                wave_lds_sync();
                const unsigned ks = lane < m ? sub[lane] : 0xFFFFFFFFu;
                wave_lds_sync();                                   // read before the histogram zeroes these slots
                for (; bit >= 0; --bit) {
                    const unsigned t = v | (1u << bit);
                    if (c_lo + __builtin_popcountll(__ballot(ks < t)) <= kth) v = t;
                }
            } else {
                for (; bit >= 0; --bit) {
                    const unsigned t = v | (1u << bit);
                    if (count_below(t) <= kth) v = t;
                }
            }
        }
        float thr = key_float(v);
        if ((n & 1) == 0) {
            // the element of rank n/2 - 1: the largest key below v if exactly n/2 keys are below it, else v (a tie)
            unsigned below = 0;
#pragma unroll
            for (int i = 0; i < kTwKeysPerLane; ++i)
                if (i < per_lane && k[i] < v) below = max(below, k[i]);
            below = wave_max_bits(below);
            const float lo = (count_below(v) == kth) ? key_float(below) : thr;
            thr = (lo + thr) / 2.0f;       // np.median of an even count: f32 mean of the middle pair
        }
        // ---- histogram of the residuals of the peaks at or above the median
        unsigned* hist = sh.hist[wave];
        hist[lane] = 0;
        hist[lane + 64] = 0;
        wave_lds_sync();
        // the pitches of the peaks at or above the median (about half) are first compacted to the front of `pitch`
        // (in place: round i writes below 64 i + 64, all of which has been read), so that the residuals -- a
        // double-precision log2 each -- are evaluated with every lane busy
        int n_pass = 0;
#pragma unroll
        for (int i = 0; i < kTwKeysPerLane; ++i) {
            if (i < per_lane) {
                const int idx = lane + 64 * i;
                const float p = idx < n ? pitch[idx] : 0.0f;
                const bool pass = idx < n && key_float(k[i]) >= thr && p > 0.0f;
                const unsigned long long mask = __ballot(pass);
                if (pass) pitch[n_pass + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = p;
                n_pass += __builtin_popcountll(mask);
            }
        }
        wave_lds_sync();
        for (int j = lane; j < n_pass; j += 64) atomicAdd(&hist[tuning_bin(tuning_residual(pitch[j]), sh.edges)], 1u);
        wave_lds_sync();
        // np.argmax: the first of the largest counts (two bins per lane; the key orders by count, then by smaller bin)
        unsigned key = 0;
        if (lane < 100) key = (hist[lane] << 8) | static_cast<unsigned>(255 - lane);
        if (lane + 64 < 100) key = max(key, (hist[lane + 64] << 8) | static_cast<unsigned>(255 - (lane + 64)));
        key = wave_max_bits(key);
        tuning_idx = 255 - static_cast<int>(key & 255u);
    }
    if (lane == 0) {
        sc.tuning_idx[seg] = tuning_idx;
        if (tuning_out) tuning_out[sd.out_row] = static_cast<float>(tb.edges[tuning_idx]);
    }
}

}  // namespace serhip
