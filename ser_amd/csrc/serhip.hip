// libserhip.so -- C ABI (include/serhip.h) over the gfx950 kernels.
// Host side: argument validation, per-call segment planning (which kernel, how many
// frames, where in the scratch), workspace management, launches.  No torch types.
#include "../../include/serhip.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
// RCCL is loaded with dlopen on first use and called through pointers: only its types are needed at compile
// time.  Where the RCCL headers are absent the few types this file names are declared here, with the values
// of rccl.h (nccl.h 2.x ABI), so that the library still builds and reports SERHIP_E_UNSUPPORTED at run time
// if librccl cannot be loaded.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat32 = 7 } ncclDataType_t;
}
#endif

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "kernels_fast.hpp"
#include "kernels_generic.hpp"
#include "kernels_mlp.hpp"
#include "kernels_prepare.hpp"
#include "kernels_project.hpp"
#include "kernels_whisper.hpp"
#include "tables.hpp"
#include "flac_decoder.hpp"
#include "wav_reader.hpp"

using namespace serhip;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(SERHIP_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int feature_dim_of(uint32_t flags) {
    return ((flags & SERHIP_FEAT_MFCC) ? kMfcc : 0) + ((flags & SERHIP_FEAT_CHROMA) ? kChroma : 0) +
           ((flags & SERHIP_FEAT_MEL) ? kMels : 0) + ((flags & SERHIP_FEAT_CONTRAST) ? 7 : 0);
}

// host tables of the fast plan (n_fft 2048)
struct HostTables {
    std::vector<v2f> twA, twB, twU, win;
    std::vector<float> mel_dense, dct;
    std::vector<float> chroma_b;   // [101][65][64][4]  MFMA B operand order (kernels_project.hpp)
    std::vector<float> mel_b;      // per 16-filter block, the 16-bin groups it touches, [64][4] each
    int mel_glo[8], mel_ghi[8], mel_off[8];
    std::vector<double> edges, mel_f, window;
    int kmin = 0, kmax = -1;
};

void build_host_tables(int sr, HostTables& h, bool with_chroma) {
    h.twA.resize(16 * 64);
    for (int k1 = 0; k1 < 16; ++k1)
        for (int l = 0; l < 64; ++l) {
            const double a = -2.0 * kPi * static_cast<double>((l * k1) % 1024) / 1024.0;
            h.twA[k1 * 64 + l] = v2f{static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a))};
        }
    h.twB.resize(16 * 4);
    for (int j1 = 0; j1 < 16; ++j1)
        for (int m2 = 0; m2 < 4; ++m2) {
            const double a = -2.0 * kPi * static_cast<double>((m2 * j1) % 64) / 64.0;
            h.twB[j1 * 4 + m2] = v2f{static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a))};
        }
    h.twU.resize(8 * 64);
    for (int r = 0; r < 8; ++r)
        for (int l = 0; l < 64; ++l) {
            const double a = -2.0 * kPi * static_cast<double>(l + 64 * r) / 2048.0;
            h.twU[r * 64 + l] = v2f{static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a))};
        }
    h.window = hann_periodic(kNfft);
    h.win.resize(16 * 64);
    for (int n1 = 0; n1 < 16; ++n1)
        for (int l = 0; l < 64; ++l)
            h.win[n1 * 64 + l] = v2f{0.5f * static_cast<float>(h.window[128 * n1 + 2 * l]),        // half window:
                                     0.5f * static_cast<float>(h.window[128 * n1 + 2 * l + 1])};   // see k_stft_frames
    h.mel_dense = mel_filterbank(sr, kNfft, kMels);
    // B operands of k_project_frames: lane (kk = lane>>4, j = lane&15) of group G holds the weights of
    // filter/class j for bins 16G + 4kk + u, u = 0..3.  Block b = filters 16b..16b+15 keeps only the
    // contiguous run of groups in which one of its filters is nonzero (the Slaney bank is banded).
    h.mel_b.assign(256, 0.0f);   // row 0: zeros, the operand of an idle accumulator slot
    for (int b = 0; b < 8; ++b) {
        int glo = -1, ghi = -1;
        for (int m = 16 * b; m < 16 * b + 16; ++m)
            for (int k = 0; k < kBins; ++k)
                if (h.mel_dense[static_cast<size_t>(m) * kBins + k] != 0.0f) {
                    if (glo < 0 || k / 16 < glo) glo = k / 16;
                    if (k / 16 > ghi) ghi = k / 16;
                }
        if (glo < 0) {   // an all-zero block (cannot happen for 128 Slaney filters; keep the kernel's test false)
            h.mel_glo[b] = 1;
            h.mel_ghi[b] = 0;
            h.mel_off[b] = 0;
            continue;
        }
        h.mel_glo[b] = glo;
        h.mel_ghi[b] = ghi;
        h.mel_off[b] = static_cast<int>(h.mel_b.size() / 256);
        for (int G = glo; G <= ghi; ++G)
            for (int l = 0; l < 64; ++l)
                for (int u = 0; u < 4; ++u) {
                    const int k = 16 * G + 4 * (l >> 4) + u;
                    const int m = 16 * b + (l & 15);
                    h.mel_b.push_back(k < kBins ? h.mel_dense[static_cast<size_t>(m) * kBins + k] : 0.0f);
                }
    }
    h.dct = dct_ortho(kMfcc, kMels);
    h.edges.resize(kTunings);
    for (int i = 0; i < kTunings; ++i) h.edges[i] = tuning_edge(i);
    h.mel_f = mel_edges(kMels, static_cast<double>(sr) / 2);
    piptrack_bin_range(sr, kNfft, &h.kmin, &h.kmax);
    if (with_chroma) {
        h.chroma_b.assign(static_cast<size_t>(kTunings) * kGroups * 256, 0.0f);
        std::vector<float> fb(static_cast<size_t>(kBins) * kChroma);
        for (int i = 0; i < kTunings; ++i) {
            chroma_filterbank(sr, kNfft, h.edges[i], fb.data());   // bin-major [k][12]
            float* dst = h.chroma_b.data() + static_cast<size_t>(i) * kGroups * 256;
            for (int G = 0; G < kGroups; ++G)
                for (int l = 0; l < 64; ++l)
                    for (int u = 0; u < 4; ++u) {
                        const int k = 16 * G + 4 * (l >> 4) + u, c = l & 15;
                        if (k < kBins && c < kChroma) dst[(G * 64 + l) * 4 + u] = fb[static_cast<size_t>(k) * kChroma + c];
                    }
        }
    }
}

}  // namespace

// The scratch of serhip_fast_features_noalloc: device memory that exists from the moment the code object is
// loaded on a GPU until the process ends -- no hipMalloc at call time.  In host code this name is only a
// stand-in for the symbol: the device address is resolved once per context (= per device) with
// hipGetSymbolAddress in serhip_create and every use goes through that pointer (serhip_ctx::na_scratch).
constexpr size_t kNoallocScratchBytes = SERHIP_NOALLOC_SCRATCH_BYTES;    // 48 MiB per GPU
constexpr int kNoallocMaxSegments = SERHIP_NOALLOC_MAX_SEGMENTS;        // segments per call
constexpr int kNoallocMaxFrameSlots = SERHIP_NOALLOC_MAX_FRAME_SLOTS;   // STFT frame slots per call
constexpr size_t kNoallocStagingBytes = 64 << 10;   // pinned host staging for descriptors + work units
static __device__ __align__(256) unsigned char g_noalloc_scratch[kNoallocScratchBytes];

namespace {

struct Chunk {
    int64_t first, count;    // range in the descriptor array
    int max_frames;
    bool generic;
    size_t first_unit = 0, n_units = 0;   // fast chunks: header + units in the work-unit array, unit count
    unsigned wgs = 0;                     // workgroups of the persistent k_stft_frames
    int unit_frames = kUnitFramesMax;
};

// host plan of one serhip_fast_features call.  Lives in the context and is reused call after call, so a
// steady stream of calls does not allocate on the host either.
struct CallPlan {
    std::vector<SegDesc> descs, generic;
    std::vector<Chunk> chunks;
    std::vector<WorkUnit> units, plain;
    std::vector<int> owner, per_wg, fill;
    std::vector<std::pair<long long, int>> heap;
    size_t max_chunk_frames = 0, gen_frames_max = 0, gen_mel_max = 0, max_chunk_segs = 0;
    void reserve(size_t segs, size_t frame_slots, size_t cus) {
        descs.reserve(segs);
        generic.reserve(segs);
        chunks.reserve(8);
        const size_t max_units = frame_slots / 16 + segs;
        units.reserve(max_units + cus + 2);
        plain.reserve(max_units);
        owner.reserve(max_units);
        per_wg.reserve(cus + 2);
        fill.reserve(cus + 2);
        heap.reserve(cus + 2);
    }
};

}  // namespace

struct serhip_ctx {
    int device = 0;
    void* ws = nullptr;          // device workspace (scratch + descriptors)
    size_t ws_bytes = 0;
    // pinned host staging for descriptors: a ring of slots, each with the event of the upload that last read it.
    // A call takes the next slot and waits for THAT slot's event -- recorded kStagingSlots uploads ago -- so a
    // stream of calls (the slab pipeline: prepare + features per slab) never waits on the host for the upload it
    // has just enqueued behind a long H2D copy.  `staging` / `staging_done` name the slot of the current call.
    static constexpr int kStagingSlots = 4;
    void* st_buf[kStagingSlots] = {nullptr, nullptr, nullptr, nullptr};
    size_t st_bytes[kStagingSlots] = {0, 0, 0, 0};
    hipEvent_t st_done[kStagingSlots] = {nullptr, nullptr, nullptr, nullptr};
    int st_next = 0;
    void* staging = nullptr;
    hipEvent_t staging_done = nullptr;
    // the workspace is shared by every call on the context: the last user's stream, and the event a call on
    // another stream records there and waits for (WorkspaceUse)
    hipEvent_t ws_done = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_marked = false;
    size_t chunk_frames = 0;     // frames of scratch per chunk (0 = default)
    int cu_count = 256;          // compute units: grid size of the persistent k_stft_frames
    // serhip_fast_features skips the descriptor upload when the workspace still holds the same plan
    // (fixed-length clip batches repeat it call after call)
    uint64_t ws_calls = 0, ws_allocs = 0;       // ensure_workspace calls / (re)allocations so far
    std::vector<char> head_copy;                // descriptors + work units last uploaded
    uint64_t head_call = 0, head_alloc = 0;     // ws_calls / ws_allocs at that upload
    hipStream_t head_stream = nullptr;
    CallPlan call_plan;                         // host plan of the current serhip_fast_features call
    // serhip_fast_features_noalloc: device address of g_noalloc_scratch on this context's GPU, pinned staging
    // and events created in serhip_create, a plan whose vectors were reserved there
    void* na_scratch = nullptr;
    void* na_staging = nullptr;
    hipEvent_t na_staging_done = nullptr, na_done = nullptr;
    bool na_staging_used = false, na_marked = false;
    hipStream_t na_stream = nullptr;
    CallPlan na_plan;
    void* whisper_blob[2] = {nullptr, nullptr};   // tables for n_mels 80 / 128
    WhisperTables whisper_tb[2] = {};
    // optional per-kernel timing with HIP events (serhip_profile_*)
    bool profiling = false;
    struct Timed {
        int kind;
        hipEvent_t start, stop;
    };
    std::vector<Timed> timed;          // launches recorded since the last read
    std::vector<hipEvent_t> event_pool;
};

struct serhip_plan {
    serhip_ctx* ctx = nullptr;
    int sample_rate = 0;
    uint32_t flags = 0;
    int dim = 0;
    void* dev_blob = nullptr;    // one allocation holding every table
    PlanTables tb{};
    ProjProgram proj_prog{};     // host copy of the per-group program: a kernel argument of k_project_frames
    int peak_cap = 0;
};

namespace {

int ensure_workspace(serhip_ctx* ctx, size_t bytes) {
    ++ctx->ws_calls;   // every user of the workspace comes through here
    if (bytes <= ctx->ws_bytes) return SERHIP_OK;
    ++ctx->ws_allocs;
    if (ctx->ws) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    const size_t want = align_up(bytes + bytes / 8, 1 << 20);
    hipError_t e = hipMalloc(&ctx->ws, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc(&ctx->ws, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(SERHIP_E_NOMEM, "workspace hipMalloc of " + std::to_string(bytes) + " bytes failed");
        }
        ctx->ws_bytes = bytes;
    } else {
        ctx->ws_bytes = want;
    }
    return SERHIP_OK;
}

// Calls on one stream are ordered by the stream.  A call on another stream than the previous user of the
// workspace first waits, on the device, for everything enqueued so far on that user's stream.  The event
// is recorded only then, so calls that stay on one stream pay nothing.  If the old stream cannot take the
// event any more (the caller destroyed it), the device is drained instead.
struct WorkspaceUse {
    serhip_ctx* ctx;
    hipStream_t stream;
    WorkspaceUse(serhip_ctx* c, hipStream_t s) : ctx(c), stream(s) {
        if (c->ws_marked && c->ws_stream != s) {
            if (hipEventRecord(c->ws_done, c->ws_stream) != hipSuccess ||
                hipStreamWaitEvent(s, c->ws_done, 0) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipDeviceSynchronize();
            }
        }
    }
    ~WorkspaceUse() {
        ctx->ws_marked = true;
        ctx->ws_stream = stream;
    }
    WorkspaceUse(const WorkspaceUse&) = delete;
    WorkspaceUse& operator=(const WorkspaceUse&) = delete;
};

int ensure_staging(serhip_ctx* ctx, size_t bytes) {
    const int slot = ctx->st_next;
    ctx->st_next = (slot + 1) % serhip_ctx::kStagingSlots;
    HIP_TRY(hipEventSynchronize(ctx->st_done[slot]));   // the upload that read this slot last (an unrecorded event is complete)
    if (bytes > ctx->st_bytes[slot]) {
        if (ctx->st_buf[slot]) HIP_TRY(hipHostFree(ctx->st_buf[slot]));
        ctx->st_buf[slot] = nullptr;
        ctx->st_bytes[slot] = 0;
        const size_t want = align_up(bytes * 2, 4096);
        HIP_TRY(hipHostMalloc(&ctx->st_buf[slot], want, hipHostMallocDefault));
        ctx->st_bytes[slot] = want;
    }
    ctx->staging = ctx->st_buf[slot];
    ctx->staging_done = ctx->st_done[slot];
    return SERHIP_OK;
}

struct ScratchLayout {
    size_t mags, mel, peaks, cnt, cnorm, tidx, scmax, mpart, mdbmax, mdbsum, mdbmin, total;
};
ScratchLayout scratch_layout(size_t frames, int peak_cap, size_t segs) {
    ScratchLayout l{};
    size_t o = 0;
    l.mags = o;
    // rows of kBinStride floats (generic path) or 16-frame tiles of kTileBytes (fast path; frames is a
    // multiple of 16 there)
    o = align_up(o + std::max(frames * kBinStride * sizeof(float), (frames + 15) / 16 * static_cast<size_t>(kTileBytes)), 256);
    l.mel = o;
    o = align_up(o + frames * kMels * sizeof(float), 256);
    l.peaks = o;
    o = align_up(o + frames * static_cast<size_t>(peak_cap) * sizeof(v2f), 256);
    l.cnt = o;
    o = align_up(o + frames * sizeof(int), 256);
    l.cnorm = o;
    o = align_up(o + frames * 16 * sizeof(float), 256);
    l.tidx = o;
    o = align_up(o + segs * sizeof(int), 256);
    l.scmax = o;
    o = align_up(o + segs * sizeof(unsigned), 256);
    l.mpart = o;
    o = align_up(o + (frames + 15) / 16 * kMels * sizeof(float), 256);
    l.mdbmax = o;
    o = align_up(o + (frames + 15) / 16 * 8 * sizeof(float), 256);
    l.mdbsum = o;
    o = align_up(o + (frames + 15) / 16 * kMels * sizeof(double), 256);
    l.mdbmin = o;
    o = align_up(o + (frames + 15) / 16 * kMels * sizeof(float), 256);
    l.total = o;
    return l;
}
Scratch scratch_at(char* base, const ScratchLayout& l, int peak_cap) {
    Scratch s{};
    s.mags = reinterpret_cast<float*>(base + l.mags);
    s.mel = reinterpret_cast<float*>(base + l.mel);
    s.peaks = reinterpret_cast<v2f*>(base + l.peaks);
    s.peak_cnt = reinterpret_cast<int*>(base + l.cnt);
    s.chroma_norm = reinterpret_cast<float*>(base + l.cnorm);
    s.tuning_idx = reinterpret_cast<int*>(base + l.tidx);
    s.seg_cmax = reinterpret_cast<unsigned*>(base + l.scmax);
    s.mel_part = reinterpret_cast<float*>(base + l.mpart);
    s.mel_dbmax = reinterpret_cast<float*>(base + l.mdbmax);
    s.mel_dbsum = reinterpret_cast<double*>(base + l.mdbsum);
    s.mel_dbmin = reinterpret_cast<float*>(base + l.mdbmin);
    s.peak_cap = peak_cap;
    return s;
}

constexpr int kGenericPeakCap = 520;  // > (1 + 2047/2) / 2 local maxima per frame

hipEvent_t take_event(serhip_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// brackets one kernel launch with events on its own stream when profiling is on
struct LaunchTimer {
    serhip_ctx* ctx;
    hipStream_t stream;
    hipEvent_t start = nullptr, stop = nullptr;
    int kind;
    LaunchTimer(serhip_ctx* c, int k, hipStream_t s, bool allowed = true) : ctx(c), stream(s), kind(k) {
        if (!ctx->profiling || !allowed) return;   // the allocation-free entry point creates no events
        start = take_event(ctx);
        stop = take_event(ctx);
        if (start && stop) (void)hipEventRecord(start, stream);
    }
    ~LaunchTimer() {
        if (!start || !stop) return;
        (void)hipEventRecord(stop, stream);
        ctx->timed.push_back({kind, start, stop});
    }
};

// ---------------------------------------------------------------------------------------
// planning of one serhip_fast_features call (host): which kernel each segment takes, where its frames
// sit in the scratch, the work units of the persistent FFT kernel
// ---------------------------------------------------------------------------------------
int plan_segments(serhip_ctx* ctx, const char* who, int64_t n_audio, const int64_t* seg_offset, const int64_t* seg_len,
                  int64_t n_seg, size_t chunk_cap, CallPlan& cp) {
    cp.descs.clear();
    cp.generic.clear();
    cp.chunks.clear();
    cp.units.clear();
    cp.max_chunk_frames = cp.gen_frames_max = cp.gen_mel_max = cp.max_chunk_segs = 0;
    Chunk cur{0, 0, 0, false};
    size_t cur_frames = 0;
    for (int64_t i = 0; i < n_seg; ++i) {
        const int64_t off = seg_offset[i], len = seg_len[i];
        if (len < 1 || off < 0 || off > n_audio - len)
            return fail(SERHIP_E_INVALID, std::string(who) + ": segment " + std::to_string(i) + " [" +
                                              std::to_string(off) + ", +" + std::to_string(len) +
                                              ") is empty or outside the audio buffer");
        if (len > 0x7fffffffLL - 4096) return fail(SERHIP_E_UNSUPPORTED, std::string(who) + ": segment too long");
        const int64_t np = std::max<int64_t>(len, 512);
        SegDesc d{};
        d.offset = off;
        d.out_row = i;
        d.len = static_cast<int>(len);
        if (np >= kNfft) {
            d.n_fft = kNfft;
            d.n_stft = d.n_mel = static_cast<int>(1 + np / kHop);
            const size_t fr = (static_cast<size_t>(d.n_stft) + 15) / 16 * 16;   // segments start on a 16-frame tile
            if (cur.count > 0 && (cur_frames + fr > chunk_cap || cur.count >= 65535)) {
                cp.chunks.push_back(cur);
                cp.max_chunk_frames = std::max(cp.max_chunk_frames, cur_frames);
                cur = Chunk{static_cast<int64_t>(cp.descs.size()), 0, 0, false};
                cur_frames = 0;
            }
            d.stft_base = d.mel_base = static_cast<int>(cur_frames);
            cur_frames += fr;
            cur.count += 1;
            cur.max_frames = std::max(cur.max_frames, d.n_stft);
            cp.descs.push_back(d);
        } else {
            const int n = static_cast<int>(np);
            d.n_fft = n;
            d.n_stft = static_cast<int>(1 + (np + 2 * (n / 2) - n) / (n / 4));
            d.n_mel = static_cast<int>(1 + (np + 2 * (n / 2) - n) / kHop);
            cp.generic.push_back(d);
        }
    }
    if (cur.count > 0) {
        cp.chunks.push_back(cur);
        cp.max_chunk_frames = std::max(cp.max_chunk_frames, cur_frames);
    }
    if (cp.max_chunk_frames > 0x7fffffffull / 2) return fail(SERHIP_E_UNSUPPORTED, std::string(who) + ": chunk too large");
    // generic segments: own chunks, own scratch region
    {
        Chunk g{static_cast<int64_t>(cp.descs.size()), 0, 0, true};
        size_t gs = 0, gm = 0;
        for (SegDesc& d : cp.generic) {
            if (g.count >= 65535 || gs + d.n_stft > 65536) {
                cp.chunks.push_back(g);
                cp.gen_frames_max = std::max(cp.gen_frames_max, gs);
                cp.gen_mel_max = std::max(cp.gen_mel_max, gm);
                g = Chunk{static_cast<int64_t>(cp.descs.size()), 0, 0, true};
                gs = gm = 0;
            }
            d.stft_base = static_cast<int>(gs);
            d.mel_base = static_cast<int>(gm);
            gs += d.n_stft;
            gm += d.n_mel;
            g.count += 1;
            g.max_frames = std::max(g.max_frames, d.n_stft + d.n_mel);
            cp.descs.push_back(d);
        }
        if (g.count > 0) {
            cp.chunks.push_back(g);
            cp.gen_frames_max = std::max(cp.gen_frames_max, gs);
            cp.gen_mel_max = std::max(cp.gen_mel_max, gm);
        }
    }

    // ---- work units of k_stft_frames: slices of the fast segments, chunk by chunk.  128 frames per unit
    // when there is work for every CU; a small call (one recording, a few windows) is cut finer, down to
    // 16 frames, so that it spreads over the chip instead of queueing on a few workgroups.
    for (Chunk& c : cp.chunks) {
        if (c.generic) continue;
        cp.max_chunk_segs = std::max(cp.max_chunk_segs, static_cast<size_t>(c.count));
        size_t frames = 0;
        for (int64_t i = 0; i < c.count; ++i) frames += static_cast<size_t>(cp.descs[static_cast<size_t>(c.first + i)].n_stft);
        const size_t share = (frames + ctx->cu_count - 1) / static_cast<size_t>(ctx->cu_count);
        c.unit_frames = static_cast<int>(std::min<size_t>(kUnitFramesMax, std::max<size_t>(16, (share + 15) / 16 * 16)));
        std::vector<WorkUnit>& plain = cp.plain;
        plain.clear();
        for (int64_t i = 0; i < c.count; ++i) {
            const int n_stft = cp.descs[static_cast<size_t>(c.first + i)].n_stft;
            for (int t0 = 0; t0 < n_stft; t0 += c.unit_frames) plain.push_back(WorkUnit{static_cast<int>(i), t0});
        }
        // one workgroup per CU (the kernel's LDS footprint allows no more), never more than units; every
        // unit goes, in order, to the workgroup with the fewest frames so far
        c.n_units = plain.size();
        c.wgs = static_cast<unsigned>(std::min<size_t>(c.n_units, static_cast<size_t>(ctx->cu_count)));
        std::vector<int>& owner = cp.owner;
        std::vector<int>& per_wg = cp.per_wg;
        owner.assign(plain.size(), 0);
        per_wg.assign(c.wgs + 1, 0);
        auto unit_len = [&](const WorkUnit& w) {
            return std::min(c.unit_frames, cp.descs[static_cast<size_t>(c.first + w.seg)].n_stft - w.t_begin);
        };
        bool uniform = true;
        for (size_t k = 1; k < plain.size() && uniform; ++k) uniform = unit_len(plain[k]) == unit_len(plain[0]);
        if (uniform) {   // equal units: least-loaded is round robin
            for (size_t k = 0; k < plain.size(); ++k) {
                owner[k] = static_cast<int>(k % c.wgs);
                per_wg[owner[k] + 1] += 1;
            }
        } else {
            typedef std::pair<long long, int> Load;   // frames so far, workgroup: a min-heap
            std::vector<Load>& q = cp.heap;
            q.clear();
            for (unsigned b = 0; b < c.wgs; ++b) q.push_back(Load{0, static_cast<int>(b)});   // already a heap
            for (size_t k = 0; k < plain.size(); ++k) {
                std::pop_heap(q.begin(), q.end(), std::greater<Load>());
                Load& l = q.back();
                l.first += unit_len(plain[k]);
                owner[k] = l.second;
                per_wg[l.second + 1] += 1;
                std::push_heap(q.begin(), q.end(), std::greater<Load>());
            }
        }
        for (unsigned b = 0; b < c.wgs; ++b) per_wg[b + 1] += per_wg[b];
        c.first_unit = cp.units.size();
        for (unsigned b = 0; b <= c.wgs; ++b) cp.units.push_back(WorkUnit{per_wg[b], 0});   // header
        const size_t list0 = cp.units.size();
        cp.units.resize(list0 + plain.size());
        std::vector<int>& fill = cp.fill;
        fill.assign(per_wg.begin(), per_wg.end() - 1);
        for (size_t k = 0; k < plain.size(); ++k) cp.units[list0 + static_cast<size_t>(fill[owner[k]]++)] = plain[k];
    }
    return SERHIP_OK;
}

// where the pieces of a planned call sit, relative to the base of its scratch:
// [descriptors | work units | fast scratch | generic scratch]
struct Placement {
    size_t seg_bytes, desc_bytes, head_bytes, total;
    ScratchLayout lf, lg;
};
Placement place_call(const CallPlan& cp, int peak_cap) {
    Placement p{};
    p.seg_bytes = align_up(cp.descs.size() * sizeof(SegDesc), 256);
    p.desc_bytes = p.seg_bytes + align_up(cp.units.size() * sizeof(WorkUnit), 256);
    p.head_bytes = p.seg_bytes + cp.units.size() * sizeof(WorkUnit);
    p.lf = scratch_layout(cp.max_chunk_frames, peak_cap, cp.max_chunk_segs);
    p.lg = scratch_layout(std::max(cp.gen_frames_max, cp.gen_mel_max), kGenericPeakCap, 0);
    p.total = p.desc_bytes + p.lf.total + p.lg.total;
    return p;
}

// the launches of a planned call whose descriptors and units are (being) uploaded to `base`
int launch_chunks(serhip_plan* plan, const CallPlan& cp, const Placement& pl, char* base, const float* audio_dev,
                  float* out_dev, float* tuning_dev, hipStream_t stream, bool timed) {
    serhip_ctx* ctx = plan->ctx;
    SegDesc* d_descs = reinterpret_cast<SegDesc*>(base);
    const WorkUnit* d_units = reinterpret_cast<const WorkUnit*>(base + pl.seg_bytes);
    const Scratch sf = scratch_at(base + pl.desc_bytes, pl.lf, plan->peak_cap);
    const Scratch sg = scratch_at(base + pl.desc_bytes + pl.lf.total, pl.lg, kGenericPeakCap);
    const bool want_chroma = (plan->flags & SERHIP_FEAT_CHROMA) != 0;
    const bool want_proj = (plan->flags & (SERHIP_FEAT_MFCC | SERHIP_FEAT_CHROMA | SERHIP_FEAT_MEL)) != 0;
    for (const Chunk& c : cp.chunks) {
        const SegDesc* dd = d_descs + c.first;
        if (!c.generic) {
            // the atomicMax route of the segment maximum starts from zero; with chroma on k_segment_finish<1> stores it
            if (!want_chroma) HIP_TRY(hipMemsetAsync(sf.seg_cmax, 0, static_cast<size_t>(c.count) * sizeof(unsigned), stream));
            {
                LaunchTimer lt(ctx, SERHIP_KERNEL_STFT_FRAMES, stream, timed);
                auto* k1 = want_proj ? (want_chroma ? k_stft_frames<true, true> : k_stft_frames<true, false>)
                                     : (want_chroma ? k_stft_frames<false, true> : k_stft_frames<false, false>);
                hipLaunchKernelGGL(k1, dim3(c.wgs), dim3(kWaves * 64), 0, stream, audio_dev, dd,
                                   d_units + c.first_unit, c.unit_frames, plan->tb, sf);
            }
#if SERHIP_K1_STAMPS
            {   // diagnostic build: print the per-phase cycle sums the FFT kernel left in the dB-row scratch
                (void)hipStreamSynchronize(stream);
                std::vector<unsigned> st(static_cast<size_t>(c.wgs) * kWaves * 16);
                (void)hipMemcpy(st.data(), sf.mel, st.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
                double sum[16] = {0};
                for (size_t w = 0; w < static_cast<size_t>(c.wgs) * kWaves; ++w)
                    for (int i = 0; i < 13; ++i) sum[i] += st[w * 16 + i];
                std::fprintf(stderr, "k1 stamps: frames/wave %.1f, cycles/frame total %.0f; phases:", sum[10] / (c.wgs * kWaves),
                             sum[11] / sum[10]);
                for (int i = 0; i <= 9; ++i) std::fprintf(stderr, " P%d %.0f", i, sum[i] / sum[10]);
                double lo = 1e30, hi = 0, t0 = 1e30, t1 = 0;
                for (size_t w = 0; w < static_cast<size_t>(c.wgs) * kWaves; ++w) {
                    lo = std::min(lo, static_cast<double>(st[w * 16 + 12]));
                    hi = std::max(hi, static_cast<double>(st[w * 16 + 12]));
                    t0 = std::min(t0, static_cast<double>(st[w * 16 + 13]));
                    t1 = std::max(t1, static_cast<double>(st[w * 16 + 13]) + st[w * 16 + 12]);
                }
                std::fprintf(stderr, "   clock %.3f GHz; wave lifetime min %.3f max %.3f ms; first start to last end %.3f ms\n",
                             sum[11] / sum[12] * 0.1, lo * 1e-5, hi * 1e-5, (t1 - t0) * 1e-5);
            }
#endif
            const dim3 segs_grid(static_cast<unsigned>(c.count));
            if (want_chroma || tuning_dev) {   // tuning estimate per segment (zeros when chroma is off)
                LaunchTimer lt(ctx, SERHIP_KERNEL_SEGMENT_FINISH, stream, timed);
                PlanTables tbl = plan->tb;
                if (want_chroma && c.count >= kTwMinSegments) {
                    // a large batch: one wave per segment for everything that fits a wave's staging; the rest is left
                    // to the workgroup kernel below, which skips the segments already done
                    hipLaunchKernelGGL(k_tuning_waves, dim3(static_cast<unsigned>((c.count + kTwWaves - 1) / kTwWaves)),
                                       dim3(kTwWaves * 64), 0, stream, dd, static_cast<int>(c.count), plan->tb, sf, tuning_dev);
                    tbl.flags |= kFlagTuningWavesRan;
                }
                hipLaunchKernelGGL(k_segment_finish<1>, segs_grid, dim3(256), 0, stream, dd, tbl, sf, out_dev,
                                   tuning_dev, plan->dim);
            }
            if (want_proj) {
                const int tiles = (c.max_frames + kProjTile - 1) / kProjTile;
                LaunchTimer lt(ctx, SERHIP_KERNEL_PROJECT_FRAMES, stream, timed);
                hipLaunchKernelGGL(k_project_frames, dim3((tiles + kProjWaves - 1) / kProjWaves, static_cast<unsigned>(c.count)),
                                   dim3(kProjWaves * 64), 0,
                                   stream, dd, plan->tb, sf, plan->proj_prog);
            }
            {
                LaunchTimer lt(ctx, SERHIP_KERNEL_SEGMENT_FINISH, stream, timed);
                hipLaunchKernelGGL(k_segment_finish<2>, segs_grid, dim3(256), 0, stream, dd, plan->tb, sf, out_dev,
                                   tuning_dev, plan->dim);
            }
        } else {
            const dim3 grid(static_cast<unsigned>(c.max_frames), static_cast<unsigned>(c.count));
            {
                LaunchTimer lt(ctx, SERHIP_KERNEL_DFT_FRAMES, stream, timed);
                hipLaunchKernelGGL(k_dft_frames, grid, dim3(256), 0, stream, audio_dev, dd, plan->tb, sg);
            }
            {
                LaunchTimer lt(ctx, SERHIP_KERNEL_SEGMENT_FINISH, stream, timed);
                hipLaunchKernelGGL(k_segment_finish<0>, dim3(static_cast<unsigned>(c.count)), dim3(256), 0, stream, dd,
                                   plan->tb, sg, out_dev, tuning_dev, plan->dim);
            }
        }
    }
    HIP_TRY(hipGetLastError());
    return SERHIP_OK;
}

// constant tables of the Whisper front end (16 kHz, n_fft 400), built once per n_mels
int ensure_whisper_tables(serhip_ctx* ctx, int n_mels) {
    const int slot = (n_mels == 80) ? 0 : 1;
    if (ctx->whisper_blob[slot]) return SERHIP_OK;
    std::vector<v2d> tw200(kWHalf), tw400(kWBins);
    for (int m = 0; m < kWHalf; ++m) {
        const double a = -2.0 * kPi * m / 200.0;
        tw200[m] = v2d{std::cos(a), std::sin(a)};
    }
    for (int k = 0; k < kWBins; ++k) {
        const double a = -2.0 * kPi * k / 400.0;
        tw400[k] = v2d{std::cos(a), std::sin(a)};
    }
    std::vector<double> win(kWNfft);
    for (int j = 0; j < kWNfft; ++j) win[j] = 0.5 - 0.5 * std::cos(2.0 * kPi * j / 400.0);
    const std::vector<float> dense = mel_filterbank(16000, kWNfft, n_mels);
    std::vector<float> mel_w;
    std::vector<int4> per_mel(n_mels, int4{0, 0, 0, 0});   // first bin, taps, offset into mel_w
    for (int m = 0; m < n_mels; ++m) {
        int first = -1, last = -1;
        for (int k = 0; k < kWBins; ++k)
            if (dense[static_cast<size_t>(m) * kWBins + k] != 0.0f) {
                if (first < 0) first = k;
                last = k;
            }
        if (first >= 0) {
            per_mel[m] = int4{first, last - first + 1, static_cast<int>(mel_w.size()), 0};
            for (int k = first; k <= last; ++k) mel_w.push_back(dense[static_cast<size_t>(m) * kWBins + k]);
        }
    }
    // two accumulator slots per lane (rng[lane], rng[64 + lane]): {first bin, taps, weight offset, mel or -1}.
    // 128 mels: lane l takes mels l and 64 + l.  80 mels: lane l takes mel l, and the 16 widest filters
    // (64..79, up to 30 taps) are cut in halves over the second slots of lanes 0..31, whose sums the even
    // lane of each pair adds up -- the tap loop runs 15 times instead of 30.
    std::vector<int4> rng(128, int4{0, 0, 0, -1});
    const bool pair_sum = n_mels == 80;
    for (int l = 0; l < 64; ++l) {
        if (l < n_mels) rng[l] = int4{per_mel[l].x, per_mel[l].y, per_mel[l].z, l};
        if (!pair_sum) {
            if (64 + l < n_mels) rng[64 + l] = int4{per_mel[64 + l].x, per_mel[64 + l].y, per_mel[64 + l].z, 64 + l};
        } else if (l < 2 * (n_mels - 64)) {
            const int m = 64 + (l >> 1);
            const int c = per_mel[m].y, c0 = (c + 1) / 2;
            rng[64 + l] = (l & 1) ? int4{per_mel[m].x + c0, c - c0, per_mel[m].z + c0, m}
                                  : int4{per_mel[m].x, c0, per_mel[m].z, m};
        }
    }
    if (static_cast<int>(mel_w.size()) > kWMelWCap) return fail(SERHIP_E_UNSUPPORTED, "whisper mel taps exceed capacity");
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o = align_up(o + bytes, 256);
        return at;
    };
    const size_t o200 = take(tw200.size() * sizeof(v2d)), o400 = take(tw400.size() * sizeof(v2d)),
                 owin = take(win.size() * sizeof(double)), omw = take(std::max<size_t>(mel_w.size(), 1) * sizeof(float)),
                 orng = take(rng.size() * sizeof(int4));
    std::vector<char> blob(o, 0);
    std::memcpy(blob.data() + o200, tw200.data(), tw200.size() * sizeof(v2d));
    std::memcpy(blob.data() + o400, tw400.data(), tw400.size() * sizeof(v2d));
    std::memcpy(blob.data() + owin, win.data(), win.size() * sizeof(double));
    if (!mel_w.empty()) std::memcpy(blob.data() + omw, mel_w.data(), mel_w.size() * sizeof(float));
    std::memcpy(blob.data() + orng, rng.data(), rng.size() * sizeof(int4));
    void* dev = nullptr;
    if (hipMalloc(&dev, o) != hipSuccess) return fail(SERHIP_E_NOMEM, "whisper table hipMalloc failed");
    if (hipMemcpy(dev, blob.data(), o, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(dev);
        return fail(SERHIP_E_HIP, "whisper table upload failed");
    }
    char* d = static_cast<char*>(dev);
    WhisperTables& wt = ctx->whisper_tb[slot];
    wt.tw200 = reinterpret_cast<const v2d*>(d + o200);
    wt.tw400 = reinterpret_cast<const v2d*>(d + o400);
    wt.win = reinterpret_cast<const double*>(d + owin);
    wt.mel_w = reinterpret_cast<const float*>(d + omw);
    wt.mel_rng = reinterpret_cast<const int4*>(d + orng);
    wt.mel_nnz = static_cast<int>(mel_w.size());
    wt.n_mels = n_mels;
    wt.mel_pair = pair_sum ? 1 : 0;
    ctx->whisper_blob[slot] = dev;
    return SERHIP_OK;
}

int whisper_logmel_launch(serhip_ctx* ctx, const float* audio_dev, int32_t n_clips, int64_t n_samples, int32_t n_mels,
                          float* out_dev, hipStream_t stream) {
    int rc = ensure_whisper_tables(ctx, n_mels);
    if (rc != SERHIP_OK) return rc;
    if (n_clips > 65535) return fail(SERHIP_E_UNSUPPORTED, "serhip_whisper_logmel: more than 65535 clips per call");
    rc = ensure_workspace(ctx, align_up(static_cast<size_t>(n_clips) * sizeof(unsigned), 256));
    if (rc != SERHIP_OK) return rc;
    WorkspaceUse use(ctx, stream);
    unsigned* clip_max = reinterpret_cast<unsigned*>(ctx->ws);
    HIP_TRY(hipMemsetAsync(clip_max, 0, static_cast<size_t>(n_clips) * sizeof(unsigned), stream));
    const int n_frames = static_cast<int>(n_samples / kWHop);
    const WhisperTables& wt = ctx->whisper_tb[n_mels == 80 ? 0 : 1];
    const dim3 grid((n_frames + kWTile - 1) / kWTile, static_cast<unsigned>(n_clips));
    hipLaunchKernelGGL(k_whisper_frames, grid, dim3(kWWaves * 64), 0, stream, audio_dev, static_cast<long long>(n_samples),
                       n_frames, wt, out_dev, clip_max);
    const long long per_clip = static_cast<long long>(n_mels) * n_frames;
    const unsigned gx = static_cast<unsigned>(std::min<long long>((per_clip + 255) / 256, 4096));
    hipLaunchKernelGGL(k_whisper_finalize, dim3(gx, static_cast<unsigned>(n_clips)), dim3(256), 0, stream, out_dev,
                       per_clip, clip_max);
    HIP_TRY(hipGetLastError());
    return SERHIP_OK;
}

}  // namespace

extern "C" {

int serhip_version(void) { return 100; }

const char* serhip_last_error(void) { return g_last_error.c_str(); }

int serhip_create(int device_id, serhip_ctx** out) {
    if (!out) return fail(SERHIP_E_INVALID, "serhip_create: out is NULL");
    *out = nullptr;
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (device_id < 0 || device_id >= n_dev)
        return fail(SERHIP_E_INVALID, "serhip_create: device " + std::to_string(device_id) + " of " +
                                          std::to_string(n_dev) + " does not exist");
    HIP_TRY(hipSetDevice(device_id));
    serhip_ctx* ctx = new serhip_ctx();
    ctx->device = device_id;
    {
        bool ok = hipEventCreateWithFlags(&ctx->ws_done, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < serhip_ctx::kStagingSlots; ++i)
            ok = hipEventCreateWithFlags(&ctx->st_done[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            for (hipEvent_t e : ctx->st_done)
                if (e) (void)hipEventDestroy(e);
            if (ctx->ws_done) (void)hipEventDestroy(ctx->ws_done);
            delete ctx;
            return fail(SERHIP_E_HIP, "serhip_create: hipEventCreate failed");
        }
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0)
            ctx->cu_count = cus;
        else
            (void)hipGetLastError();
    }
    // set-up of the allocation-free entry point, once per context with its device current: the device
    // address of the static scratch ON THIS GPU (every GPU holds its own copy at its own address)
    {
        void* sym = nullptr;
        hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(g_noalloc_scratch));
        if (e == hipSuccess) e = hipHostMalloc(&ctx->na_staging, kNoallocStagingBytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->na_staging_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->na_done, hipEventDisableTiming);
        if (e != hipSuccess) {
            const std::string msg = std::string("serhip_create: static scratch set-up failed: ") + hipGetErrorString(e);
            (void)hipGetLastError();
            serhip_destroy(ctx);
            return fail(SERHIP_E_HIP, msg);
        }
        ctx->na_scratch = sym;
        ctx->na_plan.reserve(kNoallocMaxSegments, kNoallocMaxFrameSlots + 16 * kNoallocMaxSegments,
                             static_cast<size_t>(ctx->cu_count));
    }
    if (const char* env = std::getenv("SERHIP_CHUNK_FRAMES")) {
        const long long v = std::atoll(env);
        if (v > 0) ctx->chunk_frames = static_cast<size_t>(v);
    }
    *out = ctx;
    return SERHIP_OK;
}

void serhip_destroy(serhip_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->ws) (void)hipFree(ctx->ws);
    for (void* b : ctx->whisper_blob)
        if (b) (void)hipFree(b);
    for (void* b : ctx->st_buf)
        if (b) (void)hipHostFree(b);
    if (ctx->na_staging) (void)hipHostFree(ctx->na_staging);
    if (ctx->na_staging_done) (void)hipEventDestroy(ctx->na_staging_done);
    if (ctx->na_done) (void)hipEventDestroy(ctx->na_done);
    for (hipEvent_t e : ctx->st_done)
        if (e) (void)hipEventDestroy(e);
    if (ctx->ws_done) (void)hipEventDestroy(ctx->ws_done);
    for (const serhip_ctx::Timed& t : ctx->timed) {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    delete ctx;
}

int serhip_plan_fast(serhip_ctx* ctx, int sample_rate, uint32_t feature_flags, serhip_plan** out) {
    if (!ctx || !out) return fail(SERHIP_E_INVALID, "serhip_plan_fast: NULL argument");
    *out = nullptr;
    if (sample_rate <= 0) return fail(SERHIP_E_INVALID, "serhip_plan_fast: sample_rate must be positive");
    if (feature_flags & ~15u) return fail(SERHIP_E_INVALID, "serhip_plan_fast: unknown feature flag bits");
    HIP_TRY(hipSetDevice(ctx->device));
    HostTables h;
    const bool with_chroma = (feature_flags & SERHIP_FEAT_CHROMA) != 0;
    build_host_tables(sample_rate, h, with_chroma);
    if (with_chroma && h.kmax >= kBins - 1)
        return fail(SERHIP_E_UNSUPPORTED,
                    "serhip_plan_fast: piptrack range reaches the Nyquist bin (sample_rate <= 8000) -- not implemented");

    // k_project_frames keeps two mel blocks live at a time: runs must advance monotonically and block
    // b + 2 must start after block b ends (true for the Slaney bank at any practical sample rate)
    if (feature_flags & (SERHIP_FEAT_MFCC | SERHIP_FEAT_MEL))
        for (int b = 0; b < 8; ++b) {
            const bool bad = h.mel_glo[b] > h.mel_ghi[b] || (b > 0 && (h.mel_glo[b] < h.mel_glo[b - 1] || h.mel_ghi[b] < h.mel_ghi[b - 1])) ||
                             (b > 1 && h.mel_glo[b] <= h.mel_ghi[b - 2]);
            if (bad)
                return fail(SERHIP_E_UNSUPPORTED,
                            "serhip_plan_fast: mel filterbank layout at this sample rate is not supported (filter block " +
                                std::to_string(b) + ")");
        }
    // the per-group program of k_project_frames (see kernels_project.hpp)
    int4 prog[kGroups];
    {
        int cur = 0;
        for (int G = 0; G < kGroups; ++G) {
            int4 e{0, 0, 0, 0};
            if (cur < 8 && G >= h.mel_glo[cur] && G <= h.mel_ghi[cur]) {
                e.x = h.mel_off[cur] + G - h.mel_glo[cur];
                e.z |= 1;
            }
            if (cur + 1 < 8 && G >= h.mel_glo[cur + 1] && G <= h.mel_ghi[cur + 1]) {
                e.y = h.mel_off[cur + 1] + G - h.mel_glo[cur + 1];
                e.z |= 2;
            }
            while (cur < 8 && h.mel_ghi[cur] <= G) {
                ++e.w;
                ++cur;
            }
            prog[G] = e;
        }
    }

    // one device blob, sub-allocated
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o = align_up(o + bytes, 256);
        return at;
    };
    const size_t o_twA = take(h.twA.size() * sizeof(v2f)), o_twB = take(h.twB.size() * sizeof(v2f)),
                 o_twU = take(h.twU.size() * sizeof(v2f)), o_win = take(h.win.size() * sizeof(v2f)),
                 o_melb = take(std::max<size_t>(h.mel_b.size(), 4) * sizeof(float)), o_blk = take(sizeof(prog)),
                 o_chroma = take(std::max<size_t>(h.chroma_b.size(), 4) * sizeof(float)),
                 o_dct = take(h.dct.size() * sizeof(float)), o_edges = take(h.edges.size() * sizeof(double)),
                 o_melf = take(h.mel_f.size() * sizeof(double));
    std::vector<char> blob(o, 0);
    auto put = [&](size_t at, const void* src, size_t bytes) {
        if (bytes) std::memcpy(blob.data() + at, src, bytes);
    };
    put(o_twA, h.twA.data(), h.twA.size() * sizeof(v2f));
    put(o_twB, h.twB.data(), h.twB.size() * sizeof(v2f));
    put(o_twU, h.twU.data(), h.twU.size() * sizeof(v2f));
    put(o_win, h.win.data(), h.win.size() * sizeof(v2f));
    put(o_melb, h.mel_b.data(), h.mel_b.size() * sizeof(float));
    put(o_blk, prog, sizeof(prog));
    put(o_chroma, h.chroma_b.data(), h.chroma_b.size() * sizeof(float));
    put(o_dct, h.dct.data(), h.dct.size() * sizeof(float));
    put(o_edges, h.edges.data(), h.edges.size() * sizeof(double));
    put(o_melf, h.mel_f.data(), h.mel_f.size() * sizeof(double));

    serhip_plan* plan = new serhip_plan();
    plan->ctx = ctx;
    plan->sample_rate = sample_rate;
    plan->flags = feature_flags;
    plan->dim = feature_dim_of(feature_flags);
    if (hipMalloc(&plan->dev_blob, o) != hipSuccess) {
        delete plan;
        return fail(SERHIP_E_NOMEM, "serhip_plan_fast: hipMalloc of the table blob failed");
    }
    if (hipMemcpy(plan->dev_blob, blob.data(), o, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(plan->dev_blob);
        delete plan;
        return fail(SERHIP_E_HIP, "serhip_plan_fast: table upload failed");
    }
    char* d = static_cast<char*>(plan->dev_blob);
    PlanTables& tb = plan->tb;
    tb.twA = reinterpret_cast<const v2f*>(d + o_twA);
    tb.twB = reinterpret_cast<const v2f*>(d + o_twB);
    tb.twU = reinterpret_cast<const v2f*>(d + o_twU);
    tb.win = reinterpret_cast<const v2f*>(d + o_win);
    tb.mel_b = reinterpret_cast<const float4*>(d + o_melb);
    tb.chroma_b = reinterpret_cast<const float4*>(d + o_chroma);
    tb.mel_prog = reinterpret_cast<const int4*>(d + o_blk);
    std::memcpy(plan->proj_prog.g, prog, sizeof(prog));
    tb.dct = reinterpret_cast<const float*>(d + o_dct);
    tb.edges = reinterpret_cast<const double*>(d + o_edges);
    tb.mel_f = reinterpret_cast<const double*>(d + o_melf);
    tb.kmin = h.kmin;
    tb.kmax = h.kmax;
    tb.sample_rate = sample_rate;
    tb.flags = feature_flags;
    {
        double wmax = 0.0;
        for (int m = 0; m < kMels; ++m) {
            double w = 0.0;
            for (int k = 0; k < kBins; ++k) w += static_cast<double>(h.mel_dense[static_cast<size_t>(m) * kBins + k]);
            wmax = std::max(wmax, w);
        }
        tb.mel_wsum_max = static_cast<float>(wmax * (1.0 + 1e-6));
    }
    plan->peak_cap = std::max(8, (h.kmax - h.kmin + 2) / 2 + 2);
    if (plan->peak_cap > static_cast<int>(kPeakCountMask)) {   // the count shares its word with the frame-max key
        serhip_plan_destroy(plan);
        return fail(SERHIP_E_UNSUPPORTED, "serhip_plan_fast: more peak candidates per frame than the count field holds");
    }
    *out = plan;
    return SERHIP_OK;
}

void serhip_plan_destroy(serhip_plan* plan) {
    if (!plan) return;
    (void)hipSetDevice(plan->ctx->device);
    (void)hipDeviceSynchronize();
    if (plan->dev_blob) (void)hipFree(plan->dev_blob);
    delete plan;
}

int serhip_plan_feature_dim(const serhip_plan* plan) { return plan ? plan->dim : SERHIP_E_INVALID; }

// ---------------------------------------------------------------------------------------
// RCCL fan-in of the feature shards (include/serhip.h: serhip_comm_*, serhip_allgather_rows)
// ---------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void* lib = nullptr;
    // prototypes spelled out (not decltype(&ncclXxx)): nothing here depends on the header's declarations
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*group_start)() = nullptr;
    ncclResult_t (*group_end)() = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
    std::string problem;
};
// a process that already holds an RCCL (torch's own copy) gets that one: dlopen by soname returns it.
// SERHIP_RCCL_LIBRARY names the library to load instead (a site-specific build; the host tests point it at
// a file that does not exist to drive the failure path).
const RcclApi* rccl_api() {
    static const RcclApi api = [] {
        RcclApi a;
        std::string tried;
        std::vector<std::string> names;
        if (const char* env = std::getenv("SERHIP_RCCL_LIBRARY"); env && *env)
            names.push_back(env);
        else
            names = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        std::string why;
        for (const std::string& name : names) {
            a.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
            const char* e = dlerror();   // ONE call: it returns the message and clears it
            why = e ? e : "unknown error";
            tried += (tried.empty() ? "" : ", ") + name;
        }
        if (!a.lib) {
            a.problem = "RCCL could not be loaded (tried " + tried + "): " + why;
            return a;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(a.lib, n);
            if (!p && a.problem.empty()) a.problem = std::string("RCCL symbol missing: ") + n;
            return p;
        };
        a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(sym("ncclGetUniqueId"));
        a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(sym("ncclCommInitRank"));
        a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(sym("ncclCommDestroy"));
        a.broadcast = reinterpret_cast<decltype(a.broadcast)>(sym("ncclBroadcast"));
        a.all_gather = reinterpret_cast<decltype(a.all_gather)>(sym("ncclAllGather"));
        a.group_start = reinterpret_cast<decltype(a.group_start)>(sym("ncclGroupStart"));
        a.group_end = reinterpret_cast<decltype(a.group_end)>(sym("ncclGroupEnd"));
        a.error_string = reinterpret_cast<decltype(a.error_string)>(sym("ncclGetErrorString"));
        return a;
    }();
    return &api;
}
int rccl_fail(const char* what, ncclResult_t r) {
    return fail(SERHIP_E_HIP, std::string(what) + ": " + rccl_api()->error_string(r));
}
}  // namespace

struct serhip_comm {
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, world = 1;
};

int serhip_comm_unique_id(unsigned char* id_out) {
    static_assert(sizeof(ncclUniqueId) == SERHIP_COMM_ID_BYTES, "SERHIP_COMM_ID_BYTES must be RCCL's id size");
    if (!id_out) return fail(SERHIP_E_INVALID, "serhip_comm_unique_id: NULL argument");
    const RcclApi& api = *rccl_api();
    if (!api.problem.empty()) return fail(SERHIP_E_UNSUPPORTED, "serhip_comm_unique_id: " + api.problem);
    ncclUniqueId id;
    const ncclResult_t r = api.get_unique_id(&id);
    if (r != ncclSuccess) return rccl_fail("serhip_comm_unique_id", r);
    std::memcpy(id_out, &id, sizeof(id));
    return SERHIP_OK;
}

int serhip_comm_create(serhip_ctx* ctx, const unsigned char* id, int rank, int world, serhip_comm** out) {
    if (!ctx || !id || !out) return fail(SERHIP_E_INVALID, "serhip_comm_create: NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(SERHIP_E_INVALID, "serhip_comm_create: rank outside [0, world)");
    const RcclApi& api = *rccl_api();
    if (!api.problem.empty()) return fail(SERHIP_E_UNSUPPORTED, "serhip_comm_create: " + api.problem);
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    auto* c = new serhip_comm;
    c->device = ctx->device;
    c->rank = rank;
    c->world = world;
    const ncclResult_t r = api.comm_init_rank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return rccl_fail("serhip_comm_create", r);
    }
    *out = c;
    return SERHIP_OK;
}

int serhip_allgather_rows(serhip_comm* comm, const float* shard_dev, const int64_t* rows_per_rank, int32_t dim,
                          float* out_dev, void* stream) {
    if (!comm || !rows_per_rank || !out_dev) return fail(SERHIP_E_INVALID, "serhip_allgather_rows: NULL argument");
    if (dim <= 0) return fail(SERHIP_E_INVALID, "serhip_allgather_rows: dim must be positive");
    for (int r = 0; r < comm->world; ++r)
        if (rows_per_rank[r] < 0) return fail(SERHIP_E_INVALID, "serhip_allgather_rows: negative row count");
    if (rows_per_rank[comm->rank] > 0 && !shard_dev) return fail(SERHIP_E_INVALID, "serhip_allgather_rows: NULL shard");
    const RcclApi& api = *rccl_api();
    if (!api.problem.empty()) return fail(SERHIP_E_UNSUPPORTED, "serhip_allgather_rows: " + api.problem);
    HIP_TRY(hipSetDevice(comm->device));
    bool equal = rows_per_rank[0] > 0;
    for (int r = 1; r < comm->world && equal; ++r) equal = rows_per_rank[r] == rows_per_rank[0];
    if (equal) {
        // the common case (shard_bounds pads nothing when world divides the row count; bench.py's weak scaling):
        // ONE ncclAllGather -- over xGMI every rank sends its shard to its 7 peers directly
        const ncclResult_t r = api.all_gather(shard_dev, out_dev, static_cast<size_t>(rows_per_rank[0]) * dim, ncclFloat32,
                                              comm->comm, static_cast<hipStream_t>(stream));
        if (r != ncclSuccess) return rccl_fail("serhip_allgather_rows (ncclAllGather)", r);
        return SERHIP_OK;
    }
    // uneven or empty shards: one grouped call of a broadcast per non-empty shard (no padding, no extra copy)
    ncclResult_t r = api.group_start();
    if (r != ncclSuccess) return rccl_fail("serhip_allgather_rows", r);
    int64_t row = 0;
    for (int root = 0; root < comm->world && r == ncclSuccess; ++root) {
        const int64_t n = rows_per_rank[root];
        if (n > 0)
            r = api.broadcast(root == comm->rank ? shard_dev : out_dev + row * dim /* ignored off the root */, out_dev + row * dim, static_cast<size_t>(n) * dim,
                              ncclFloat32, root, comm->comm, static_cast<hipStream_t>(stream));
        row += n;
    }
    const ncclResult_t e = api.group_end();
    if (r != ncclSuccess) return rccl_fail("serhip_allgather_rows", r);
    if (e != ncclSuccess) return rccl_fail("serhip_allgather_rows", e);
    return SERHIP_OK;
}

void serhip_comm_destroy(serhip_comm* comm) {
    if (!comm) return;
    if (comm->comm) (void)rccl_api()->comm_destroy(comm->comm);
    delete comm;
}

int64_t serhip_frame_count(int64_t n_samples) {
    if (n_samples < 1) return SERHIP_E_INVALID;
    const int64_t np = std::max<int64_t>(n_samples, 512);
    if (np >= kNfft) return 1 + np / kHop;
    const int64_t n = np, hop = n / 4;
    return 1 + (np + 2 * (n / 2) - n) / hop;
}

int serhip_fast_features(serhip_plan* plan, const float* audio_dev, int64_t n_audio, const int64_t* seg_offset,
                         const int64_t* seg_len, int64_t n_seg, float* out_dev, float* tuning_dev, void* stream_v) {
    if (!plan) return fail(SERHIP_E_INVALID, "serhip_fast_features: plan is NULL");
    if (n_seg < 0) return fail(SERHIP_E_INVALID, "serhip_fast_features: n_seg < 0");
    if (n_seg == 0) return SERHIP_OK;
    if (plan->dim == 0) return SERHIP_OK;
    if (!audio_dev || !seg_offset || !seg_len || !out_dev)
        return fail(SERHIP_E_INVALID, "serhip_fast_features: NULL buffer");
    serhip_ctx* ctx = plan->ctx;
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    HIP_TRY(hipSetDevice(ctx->device));

    CallPlan& cp = ctx->call_plan;
    const size_t chunk_cap = ctx->chunk_frames ? ctx->chunk_frames : (static_cast<size_t>(1) << 21);
    int rc = plan_segments(ctx, "serhip_fast_features", n_audio, seg_offset, seg_len, n_seg, chunk_cap, cp);
    if (rc != SERHIP_OK) return rc;
    const Placement pl = place_call(cp, plan->peak_cap);

    // ---- workspace: [descriptors | work units | fast scratch | generic scratch]
    rc = ensure_workspace(ctx, pl.total);
    if (rc != SERHIP_OK) return rc;
    WorkspaceUse use(ctx, stream);
    // descriptors and units travel in one copy: the staging buffer mirrors the head of the workspace
    const size_t desc_raw = cp.descs.size() * sizeof(SegDesc), unit_raw = cp.units.size() * sizeof(WorkUnit);
    char* ws = static_cast<char*>(ctx->ws);
    // the same plan as the last call, nothing else has used the workspace since, same stream: the head of
    // the workspace is still valid
    const bool head_hit = ctx->head_copy.size() == pl.head_bytes && ctx->head_call + 1 == ctx->ws_calls &&
                          ctx->head_alloc == ctx->ws_allocs && ctx->head_stream == stream &&
                          std::memcmp(ctx->head_copy.data(), cp.descs.data(), desc_raw) == 0 &&
                          (unit_raw == 0 || std::memcmp(ctx->head_copy.data() + pl.seg_bytes, cp.units.data(), unit_raw) == 0);
    if (!head_hit) {
        ctx->head_copy.assign(pl.head_bytes, 0);
        std::memcpy(ctx->head_copy.data(), cp.descs.data(), desc_raw);
        if (unit_raw) std::memcpy(ctx->head_copy.data() + pl.seg_bytes, cp.units.data(), unit_raw);
        rc = ensure_staging(ctx, pl.head_bytes);
        if (rc != SERHIP_OK) {
            ctx->head_copy.clear();
            return rc;
        }
        std::memcpy(ctx->staging, ctx->head_copy.data(), pl.head_bytes);
        hipError_t e = hipMemcpyAsync(ws, ctx->staging, pl.head_bytes, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipEventRecord(ctx->staging_done, stream);
        if (e != hipSuccess) {
            ctx->head_copy.clear();
            return fail(SERHIP_E_HIP, std::string("descriptor upload: ") + hipGetErrorString(e));
        }
        ctx->head_alloc = ctx->ws_allocs;
        ctx->head_stream = stream;
    }
    ctx->head_call = ctx->ws_calls;
    return launch_chunks(plan, cp, pl, ws, audio_dev, out_dev, tuning_dev, stream, true);
}

int serhip_fast_features_noalloc(serhip_plan* plan, const float* audio_dev, int64_t n_audio, const int64_t* seg_offset,
                                 const int64_t* seg_len, int64_t n_seg, float* out_dev, float* tuning_dev,
                                 void* stream_v) {
    if (!plan) return fail(SERHIP_E_INVALID, "serhip_fast_features_noalloc: plan is NULL");
    if (n_seg < 0) return fail(SERHIP_E_INVALID, "serhip_fast_features_noalloc: n_seg < 0");
    if (n_seg == 0) return SERHIP_OK;
    if (plan->dim == 0) return SERHIP_OK;
    if (!audio_dev || !seg_offset || !seg_len || !out_dev)
        return fail(SERHIP_E_INVALID, "serhip_fast_features_noalloc: NULL buffer");
    serhip_ctx* ctx = plan->ctx;
    if (!ctx->na_scratch || !ctx->na_staging)
        return fail(SERHIP_E_HIP, "serhip_fast_features_noalloc: the static scratch was not resolved at serhip_create");
    if (n_seg > kNoallocMaxSegments)
        return fail(SERHIP_E_UNSUPPORTED, "serhip_fast_features_noalloc: " + std::to_string(n_seg) +
                                              " segments exceed the limit of " + std::to_string(kNoallocMaxSegments) +
                                              " per call (use serhip_fast_features)");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    HIP_TRY(hipSetDevice(ctx->device));

    // the vectors of na_plan were reserved at serhip_create for the limits checked here: planning does not
    // touch the heap.  One chunk of each kind at most (chunk_cap = the frame-slot limit).
    CallPlan& cp = ctx->na_plan;
    size_t slots = 0;
    for (int64_t i = 0; i < n_seg && slots <= static_cast<size_t>(kNoallocMaxFrameSlots); ++i) {
        const int64_t len = seg_len[i];
        if (len < 1 || len > 0x7fffffffLL - 4096) continue;   // plan_segments reports it
        const size_t frames = static_cast<size_t>(serhip_frame_count(len));
        slots += len >= kNfft ? (frames + 15) / 16 * 16 : frames;
    }
    if (slots > static_cast<size_t>(kNoallocMaxFrameSlots))
        return fail(SERHIP_E_UNSUPPORTED,
                    "serhip_fast_features_noalloc: the segments need more than " + std::to_string(kNoallocMaxFrameSlots) +
                        " STFT frame slots (the static scratch holds that many; use serhip_fast_features)");
    int rc = plan_segments(ctx, "serhip_fast_features_noalloc", n_audio, seg_offset, seg_len, n_seg,
                           static_cast<size_t>(kNoallocMaxFrameSlots), cp);
    if (rc != SERHIP_OK) return rc;
    const Placement pl = place_call(cp, plan->peak_cap);
    if (pl.total > kNoallocScratchBytes || pl.head_bytes > kNoallocStagingBytes)
        return fail(SERHIP_E_UNSUPPORTED, "serhip_fast_features_noalloc: the call needs " + std::to_string(pl.total) +
                                              " bytes of scratch, the static array holds " +
                                              std::to_string(kNoallocScratchBytes) + " (use serhip_fast_features)");
    // calls through this context are ordered across streams like users of the hipMalloc'ed workspace
    if (ctx->na_marked && ctx->na_stream != stream) {
        if (hipEventRecord(ctx->na_done, ctx->na_stream) != hipSuccess ||
            hipStreamWaitEvent(stream, ctx->na_done, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
        }
    }
    ctx->na_marked = true;
    ctx->na_stream = stream;
    // the previous call's upload has left the pinned staging buffer before it is overwritten
    if (ctx->na_staging_used) HIP_TRY(hipEventSynchronize(ctx->na_staging_done));
    char* stage = static_cast<char*>(ctx->na_staging);
    std::memcpy(stage, cp.descs.data(), cp.descs.size() * sizeof(SegDesc));
    if (!cp.units.empty()) std::memcpy(stage + pl.seg_bytes, cp.units.data(), cp.units.size() * sizeof(WorkUnit));
    char* base = static_cast<char*>(ctx->na_scratch);   // the device address of g_noalloc_scratch on this GPU
    HIP_TRY(hipMemcpyAsync(base, stage, pl.head_bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(ctx->na_staging_done, stream));
    ctx->na_staging_used = true;
    return launch_chunks(plan, cp, pl, base, audio_dev, out_dev, tuning_dev, stream, false);
}

int serhip_pool_windows(serhip_ctx* ctx, const float* emb_dev, int32_t n_frames, int32_t dim, const int32_t* win_lo,
                        const int32_t* win_hi, int32_t n_win, int mode, double* out_dev, void* stream_v) {
    if (!ctx) return fail(SERHIP_E_INVALID, "serhip_pool_windows: ctx is NULL");
    if (n_win < 0 || n_frames < 0 || dim < 0) return fail(SERHIP_E_INVALID, "serhip_pool_windows: negative size");
    if (mode != 0 && mode != 1) return fail(SERHIP_E_INVALID, "serhip_pool_windows: mode must be 0 or 1");
    if (n_win == 0 || dim == 0) return SERHIP_OK;
    if (!emb_dev || !win_lo || !win_hi || !out_dev) return fail(SERHIP_E_INVALID, "serhip_pool_windows: NULL buffer");
    if (n_win > 65535) return fail(SERHIP_E_UNSUPPORTED, "serhip_pool_windows: more than 65535 windows per call");
    for (int i = 0; i < n_win; ++i)
        if (win_lo[i] < 0 || win_hi[i] <= win_lo[i] || win_hi[i] > n_frames)
            return fail(SERHIP_E_INVALID, "serhip_pool_windows: window " + std::to_string(i) + " is empty or out of range");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bytes = static_cast<size_t>(n_win) * sizeof(int32_t);
    int rc = ensure_workspace(ctx, 2 * align_up(bytes, 256));
    if (rc != SERHIP_OK) return rc;
    WorkspaceUse use(ctx, stream);
    rc = ensure_staging(ctx, 2 * bytes);
    if (rc != SERHIP_OK) return rc;
    std::memcpy(ctx->staging, win_lo, bytes);
    std::memcpy(static_cast<char*>(ctx->staging) + bytes, win_hi, bytes);
    int* d_lo = reinterpret_cast<int*>(ctx->ws);
    int* d_hi = reinterpret_cast<int*>(static_cast<char*>(ctx->ws) + align_up(bytes, 256));
    HIP_TRY(hipMemcpyAsync(d_lo, ctx->staging, bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(d_hi, static_cast<char*>(ctx->staging) + bytes, bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(ctx->staging_done, stream));
    const dim3 grid((dim + 255) / 256, static_cast<unsigned>(n_win));
    hipLaunchKernelGGL(k_pool_windows, grid, dim3(256), 0, stream, emb_dev, dim, d_lo, d_hi, n_win, mode, out_dev);
    HIP_TRY(hipGetLastError());
    return SERHIP_OK;
}

int serhip_whisper_logmel(serhip_ctx* ctx, const float* audio_dev, int32_t n_clips, int64_t n_samples, int32_t n_mels,
                          float* out_dev, void* stream_v) {
    if (!ctx) return fail(SERHIP_E_INVALID, "serhip_whisper_logmel: ctx is NULL");
    if (n_clips < 0) return fail(SERHIP_E_INVALID, "serhip_whisper_logmel: n_clips < 0");
    if (n_clips == 0) return SERHIP_OK;
    if (!audio_dev || !out_dev) return fail(SERHIP_E_INVALID, "serhip_whisper_logmel: NULL buffer");
    if (n_mels != 80 && n_mels != 128) return fail(SERHIP_E_INVALID, "serhip_whisper_logmel: n_mels must be 80 or 128");
    if (n_samples <= 200)
        return fail(SERHIP_E_INVALID, "serhip_whisper_logmel: n_samples must be > 200 (reflect padding)");
    if (n_samples / 160 > 0x7fffffffLL / 256) return fail(SERHIP_E_UNSUPPORTED, "serhip_whisper_logmel: clip too long");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    HIP_TRY(hipSetDevice(ctx->device));
    return whisper_logmel_launch(ctx, audio_dev, n_clips, n_samples, n_mels, out_dev, stream);
}

int serhip_prepare_audio(serhip_ctx* ctx, const void* raw_dev, int64_t raw_bytes, int sample_format,
                         const int64_t* clip_raw_offset, const int64_t* clip_frames, const int32_t* clip_channels,
                         const int64_t* clip_out_offset, int64_t n_clips, float* out_dev, int64_t out_samples,
                         void* stream_v) {
    if (!ctx) return fail(SERHIP_E_INVALID, "serhip_prepare_audio: ctx is NULL");
    if (n_clips < 0) return fail(SERHIP_E_INVALID, "serhip_prepare_audio: n_clips < 0");
    if (n_clips == 0) return SERHIP_OK;
    if (!raw_dev || !clip_raw_offset || !clip_frames || !clip_channels || !clip_out_offset || !out_dev)
        return fail(SERHIP_E_INVALID, "serhip_prepare_audio: NULL buffer");
    if (sample_format != SERHIP_SAMPLE_S16 && sample_format != SERHIP_SAMPLE_F32)
        return fail(SERHIP_E_INVALID, "serhip_prepare_audio: unknown sample format");
    const int64_t ssize = sample_format == SERHIP_SAMPLE_S16 ? 2 : 4;
    std::vector<PrepClip> clips(static_cast<size_t>(n_clips));
    int64_t max_frames = 0;
    for (int64_t i = 0; i < n_clips; ++i) {
        const int64_t off = clip_raw_offset[i], fr = clip_frames[i], ch = clip_channels[i], oo = clip_out_offset[i];
        if (fr < 1 || ch < 1 || ch > 64)
            return fail(SERHIP_E_INVALID, "serhip_prepare_audio: clip " + std::to_string(i) + " has no frames or a bad channel count");
        if (off < 0 || off % ssize != 0 || fr > (raw_bytes - off) / (ssize * ch) || oo < 0 || fr > out_samples - oo)
            return fail(SERHIP_E_INVALID, "serhip_prepare_audio: clip " + std::to_string(i) + " lies outside the buffers");
        clips[static_cast<size_t>(i)] = PrepClip{off, oo, fr, static_cast<int>(ch), 0};
        max_frames = std::max(max_frames, fr);
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t desc_bytes = align_up(clips.size() * sizeof(PrepClip), 256);
    const size_t peak_bytes = align_up(clips.size() * sizeof(unsigned), 256);
    int rc = ensure_workspace(ctx, desc_bytes + peak_bytes);
    if (rc != SERHIP_OK) return rc;
    WorkspaceUse use(ctx, stream);
    rc = ensure_staging(ctx, clips.size() * sizeof(PrepClip));
    if (rc != SERHIP_OK) return rc;
    std::memcpy(ctx->staging, clips.data(), clips.size() * sizeof(PrepClip));
    PrepClip* d_clips = reinterpret_cast<PrepClip*>(ctx->ws);
    unsigned* d_peak = reinterpret_cast<unsigned*>(static_cast<char*>(ctx->ws) + desc_bytes);
    HIP_TRY(hipMemcpyAsync(d_clips, ctx->staging, clips.size() * sizeof(PrepClip), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(ctx->staging_done, stream));
    HIP_TRY(hipMemsetAsync(d_peak, 0, clips.size() * sizeof(unsigned), stream));
    const unsigned char* raw = static_cast<const unsigned char*>(raw_dev);
    if (max_frames <= kPrepFusedMaxFrames && n_clips <= 0x7fffffffLL) {
        // clips short enough for one workgroup each: peak and normalisation in one launch, raw bytes re-read from L2
        const dim3 grid(static_cast<unsigned>(n_clips));
        if (sample_format == SERHIP_SAMPLE_S16)
            hipLaunchKernelGGL(k_prepare_clip<0>, grid, dim3(256), 0, stream, raw, d_clips, out_dev);
        else
            hipLaunchKernelGGL(k_prepare_clip<1>, grid, dim3(256), 0, stream, raw, d_clips, out_dev);
        HIP_TRY(hipGetLastError());
        return SERHIP_OK;
    }
    // enough workgroups per clip for ~4 frames per thread, bounded so that many short clips still
    // launch one modest grid
    const unsigned gx = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>((max_frames + 1023) / 1024, 1024)));
    for (int64_t first = 0; first < n_clips; first += 65535) {
        const unsigned gy = static_cast<unsigned>(std::min<int64_t>(65535, n_clips - first));
        if (sample_format == SERHIP_SAMPLE_S16)
            hipLaunchKernelGGL(k_prepare_convert<0>, dim3(gx, gy), dim3(256), 0, stream, raw, d_clips + first, out_dev,
                               d_peak + first);
        else
            hipLaunchKernelGGL(k_prepare_convert<1>, dim3(gx, gy), dim3(256), 0, stream, raw, d_clips + first, out_dev,
                               d_peak + first);
        hipLaunchKernelGGL(k_prepare_normalise, dim3(gx, gy), dim3(256), 0, stream, d_clips + first, out_dev,
                           d_peak + first);
    }
    HIP_TRY(hipGetLastError());
    return SERHIP_OK;
}

int serhip_mlp_predict(serhip_ctx* ctx, const void* x_dev, int x_is_f64, int64_t n_rows, int32_t n_features,
                       const double* mean_dev, const double* scale_dev, int32_t n_layers, const int32_t* layer_sizes,
                       const double* const* weights_dev, const double* const* bias_dev, int hidden_activation,
                       int out_activation, double* proba_dev, int32_t* label_dev, void* stream_v) {
    if (!ctx) return fail(SERHIP_E_INVALID, "serhip_mlp_predict: ctx is NULL");
    if (n_rows < 0) return fail(SERHIP_E_INVALID, "serhip_mlp_predict: n_rows < 0");
    if (n_rows == 0) return SERHIP_OK;
    if (!x_dev || !layer_sizes || !weights_dev || !bias_dev || !proba_dev)
        return fail(SERHIP_E_INVALID, "serhip_mlp_predict: NULL argument");
    if (n_layers < 1 || n_layers > 16) return fail(SERHIP_E_INVALID, "serhip_mlp_predict: n_layers must be 1..16");
    if (hidden_activation < 0 || hidden_activation > 3 || (out_activation != 0 && out_activation != 1))
        return fail(SERHIP_E_INVALID, "serhip_mlp_predict: unknown activation code");
    if (layer_sizes[0] != n_features) return fail(SERHIP_E_INVALID, "serhip_mlp_predict: layer_sizes[0] != n_features");
    int max_width = 0;
    for (int l = 0; l <= n_layers; ++l) {
        if (layer_sizes[l] < 1) return fail(SERHIP_E_INVALID, "serhip_mlp_predict: empty layer");
        max_width = std::max(max_width, layer_sizes[l]);
        if (l < n_layers && (!weights_dev[l] || !bias_dev[l]))
            return fail(SERHIP_E_INVALID, "serhip_mlp_predict: NULL layer parameters");
    }
    const int n_out = layer_sizes[n_layers];
    if (out_activation == 0 && n_out < 2) return fail(SERHIP_E_INVALID, "serhip_mlp_predict: softmax needs >= 2 outputs");
    if (out_activation == 1 && n_out != 1)
        return fail(SERHIP_E_UNSUPPORTED, "serhip_mlp_predict: multi-label logistic outputs are not supported");
    // rows of a layer's input are staged in LDS: 8 rows x width doubles
    if (static_cast<size_t>(max_width) * kMlpRows * sizeof(double) > 64 * 1024)
        return fail(SERHIP_E_UNSUPPORTED, "serhip_mlp_predict: layers wider than 1024 units are not supported");
    if (n_rows > (static_cast<int64_t>(1) << 31) * kMlpRows / 2)
        return fail(SERHIP_E_UNSUPPORTED, "serhip_mlp_predict: too many rows per call");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t buf_bytes = align_up(static_cast<size_t>(n_rows) * max_width * sizeof(double), 256);
    int rc = ensure_workspace(ctx, 2 * buf_bytes);
    if (rc != SERHIP_OK) return rc;
    WorkspaceUse use(ctx, stream);
    double* ping = reinterpret_cast<double*>(ctx->ws);
    double* pong = reinterpret_cast<double*>(static_cast<char*>(ctx->ws) + buf_bytes);
    const long long total = static_cast<long long>(n_rows) * n_features;
    const unsigned gs = static_cast<unsigned>(std::min<long long>((total + 255) / 256, 8192));
    if (x_is_f64)
        hipLaunchKernelGGL(k_mlp_scale<double>, dim3(gs), dim3(256), 0, stream, static_cast<const double*>(x_dev),
                           static_cast<long long>(n_rows), n_features, mean_dev, scale_dev, ping);
    else
        hipLaunchKernelGGL(k_mlp_scale<float>, dim3(gs), dim3(256), 0, stream, static_cast<const float*>(x_dev),
                           static_cast<long long>(n_rows), n_features, mean_dev, scale_dev, ping);
    const unsigned row_blocks = static_cast<unsigned>((n_rows + kMlpRows - 1) / kMlpRows);
    for (int l = 0; l < n_layers; ++l) {
        const int n_in = layer_sizes[l], n_o = layer_sizes[l + 1];
        const int act = (l + 1 < n_layers) ? hidden_activation : 0;   // the output activation runs in k_mlp_output
        hipLaunchKernelGGL(k_mlp_dense, dim3(row_blocks), dim3(256), static_cast<size_t>(n_in) * kMlpRows * sizeof(double),
                           stream, ping, static_cast<long long>(n_rows), n_in, n_o, weights_dev[l], bias_dev[l], act, pong);
        std::swap(ping, pong);
    }
    hipLaunchKernelGGL(k_mlp_output, dim3(static_cast<unsigned>((n_rows + 255) / 256)), dim3(256), 0, stream, ping,
                       static_cast<long long>(n_rows), n_out, out_activation, proba_dev, label_dev);
    HIP_TRY(hipGetLastError());
    return SERHIP_OK;
}

int serhip_tuning_from_peaks(serhip_plan* plan, const float* pitch_mag, const int32_t* frame_counts, int32_t n_frames,
                             float* tuning_out) {
    if (!plan || !frame_counts || !tuning_out) return fail(SERHIP_E_INVALID, "serhip_tuning_from_peaks: NULL argument");
    if (n_frames < 1 || n_frames > (1 << 20)) return fail(SERHIP_E_INVALID, "serhip_tuning_from_peaks: n_frames out of range");
    serhip_ctx* ctx = plan->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    int cap = 1;
    size_t total = 0;
    for (int t = 0; t < n_frames; ++t) {
        // the count shares its scratch word with the frame-max key: the kernel reads the low 10 bits only
        if (frame_counts[t] < 0 || frame_counts[t] > static_cast<int>(kPeakCountMask))
            return fail(SERHIP_E_INVALID, "serhip_tuning_from_peaks: frame " + std::to_string(t) + " has a peak count outside 0.." +
                                              std::to_string(kPeakCountMask));
        cap = std::max(cap, frame_counts[t]);
        total += static_cast<size_t>(frame_counts[t]);
    }
    // the host builds [n_frames][cap] rows: bound them (16 Mi entries = 128 MiB) instead of trusting the caller
    if (static_cast<size_t>(n_frames) * static_cast<size_t>(cap) > (static_cast<size_t>(1) << 24))
        return fail(SERHIP_E_UNSUPPORTED, "serhip_tuning_from_peaks: n_frames x largest count exceeds 2^24 peak slots");
    if (total > 0 && !pitch_mag) return fail(SERHIP_E_INVALID, "serhip_tuning_from_peaks: pitch_mag is NULL");
    // the scratch rows k_stft_frames would have written: [frame][cap] (pitch, mag) + the counts
    const size_t frames = (static_cast<size_t>(n_frames) + 15) / 16 * 16;
    const ScratchLayout lay = scratch_layout(frames, cap, 1);
    const size_t head = align_up(sizeof(SegDesc), 256);
    int rc = ensure_workspace(ctx, head + lay.total + 256);
    if (rc != SERHIP_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());   // a diagnostic call: plainly synchronous
    ctx->head_copy.clear();            // the head of the workspace is overwritten
    char* ws = static_cast<char*>(ctx->ws);
    const Scratch sc = scratch_at(ws + head, lay, cap);
    SegDesc d{};
    d.n_fft = kNfft;
    d.n_stft = d.n_mel = n_frames;
    std::vector<v2f> rows(static_cast<size_t>(n_frames) * cap, v2f{0.0f, 0.0f});
    size_t at = 0;
    for (int t = 0; t < n_frames; ++t)
        for (int j = 0; j < frame_counts[t]; ++j, ++at) rows[static_cast<size_t>(t) * cap + j] = v2f{pitch_mag[2 * at], pitch_mag[2 * at + 1]};
    HIP_TRY(hipMemcpy(ws, &d, sizeof(d), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(sc.peaks, rows.data(), rows.size() * sizeof(v2f), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(sc.peak_cnt, frame_counts, static_cast<size_t>(n_frames) * sizeof(int), hipMemcpyHostToDevice));
    float* d_tuning = reinterpret_cast<float*>(ws + head + lay.total);
    PlanTables tb = plan->tb;
    tb.flags |= SERHIP_FEAT_CHROMA;    // the estimate runs whatever the plan's flags
    // both forms of the estimate, so that the known-answer tests drive both: first the workgroup kernel alone, then
    // the wave kernel followed by the workgroup kernel for what it leaves; the answers must agree
    float* d_both = d_tuning + 1;
    hipLaunchKernelGGL(k_segment_finish<1>, dim3(1), dim3(256), 0, nullptr, reinterpret_cast<const SegDesc*>(ws), tb, sc,
                       d_tuning, d_tuning, 0);
    PlanTables tb2 = tb;
    tb2.flags |= kFlagTuningWavesRan;
    hipLaunchKernelGGL(k_tuning_waves, dim3(1), dim3(kTwWaves * 64), 0, nullptr, reinterpret_cast<const SegDesc*>(ws), 1, tb, sc,
                       d_both);
    hipLaunchKernelGGL(k_segment_finish<1>, dim3(1), dim3(256), 0, nullptr, reinterpret_cast<const SegDesc*>(ws), tb2, sc,
                       d_both, d_both, 0);
    HIP_TRY(hipGetLastError());
    float both[2] = {0.0f, 0.0f};
    HIP_TRY(hipMemcpy(both, d_tuning, sizeof(both), hipMemcpyDeviceToHost));
    if (both[0] != both[1])
        return fail(SERHIP_E_HIP, "serhip_tuning_from_peaks: k_segment_finish<1> and k_tuning_waves disagree (" +
                                      std::to_string(both[0]) + " vs " + std::to_string(both[1]) + ")");
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(tuning_out, d_tuning, sizeof(float), hipMemcpyDeviceToHost));
    return SERHIP_OK;
}

int serhip_profile_enable(serhip_ctx* ctx, int on) {
    if (!ctx) return fail(SERHIP_E_INVALID, "serhip_profile_enable: ctx is NULL");
    ctx->profiling = on != 0;
    return SERHIP_OK;
}

int serhip_profile_read(serhip_ctx* ctx, int kind, double* total_ms, int64_t* launches) {
    if (!ctx || !total_ms || !launches) return fail(SERHIP_E_INVALID, "serhip_profile_read: NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    double sum = 0.0;
    int64_t n = 0;
    std::vector<serhip_ctx::Timed> keep;
    for (const serhip_ctx::Timed& t : ctx->timed) {
        if (t.kind != kind) {
            keep.push_back(t);
            continue;
        }
        HIP_TRY(hipEventSynchronize(t.stop));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, t.start, t.stop));
        sum += ms;
        n += 1;
        ctx->event_pool.push_back(t.start);
        ctx->event_pool.push_back(t.stop);
    }
    ctx->timed.swap(keep);
    *total_ms = sum;
    *launches = n;
    return SERHIP_OK;
}

int serhip_read_wav_batch(const char* paths_blob, const int64_t* path_offset, int64_t n_files, const double* start_seconds,
                          const double* duration_seconds, const int64_t* dst_offset, const int64_t* dst_capacity, void* dst,
                          int64_t dst_bytes, int threads, serhip_wav_clip* clips) {
    if (n_files < 0) return fail(SERHIP_E_INVALID, "serhip_read_wav_batch: n_files < 0");
    if (n_files == 0) return SERHIP_OK;
    if (!paths_blob || !path_offset || !dst_offset || !dst_capacity || !dst || !clips)
        return fail(SERHIP_E_INVALID, "serhip_read_wav_batch: NULL argument");
    for (int64_t i = 0; i < n_files; ++i)
        if (dst_offset[i] < 0 || dst_capacity[i] < 0 || dst_offset[i] > dst_bytes - dst_capacity[i])
            return fail(SERHIP_E_INVALID, "serhip_read_wav_batch: region of file " + std::to_string(i) + " lies outside dst");
    unsigned char* slab = static_cast<unsigned char*>(dst);
    // runs: consecutive entries that name the same region and the same path share ONE read of the file (segments of
    // one long recording: a manifest with 1 800 utterances of an hour-long file reads 115 MB once, not 1 800 times)
    std::vector<int64_t> run_start;
    run_start.reserve(static_cast<size_t>(n_files) + 1);
    for (int64_t i = 0; i < n_files; ++i) {
        const bool shares = i > 0 && dst_offset[i] == dst_offset[i - 1] && dst_capacity[i] == dst_capacity[i - 1] &&
                            std::strcmp(paths_blob + path_offset[i], paths_blob + path_offset[i - 1]) == 0;
        if (!shares) run_start.push_back(i);
    }
    run_start.push_back(n_files);
    for (size_t r = 1; r + 1 < run_start.size(); ++r) {   // regions of DIFFERENT runs must not overlap
        const int64_t a = run_start[r - 1], b = run_start[r];
        if (dst_capacity[a] > 0 && dst_capacity[b] > 0 && dst_offset[b] < dst_offset[a] + dst_capacity[a] &&
            dst_offset[a] < dst_offset[b] + dst_capacity[b])
            return fail(SERHIP_E_INVALID, "serhip_read_wav_batch: regions of files " + std::to_string(a) + " and " +
                                              std::to_string(b) + " overlap");
    }
    const int64_t n_runs = static_cast<int64_t>(run_start.size()) - 1;
    // threads left over when the batch holds fewer files than the team has members read pieces of the large ones
    const int read_threads = static_cast<int>(std::max<int64_t>(1, threads / std::max<int64_t>(1, n_runs)));
    parallel_for(n_runs, threads, [&](int64_t r) {
        const int64_t lo = run_start[static_cast<size_t>(r)], hi = run_start[static_cast<size_t>(r) + 1];
        WaveInSlab wave;
        const int status = load_wave(paths_blob + path_offset[lo], slab, dst_offset[lo], dst_capacity[lo], &wave, read_threads);
        for (int64_t i = lo; i < hi; ++i) {
            serhip_wav_clip c{};
            c.status = status != SERHIP_WAV_OK ? status
                                               : select_clip(wave, start_seconds ? start_seconds[i] : NAN,
                                                             duration_seconds ? duration_seconds[i] : NAN, &c);
            clips[i] = c;
        }
    });
    return SERHIP_OK;
}

int serhip_stat_files(const char* paths_blob, const int64_t* path_offset, int64_t n_files, int threads, int64_t* sizes) {
    if (n_files < 0) return fail(SERHIP_E_INVALID, "serhip_stat_files: n_files < 0");
    if (n_files == 0) return SERHIP_OK;
    if (!paths_blob || !path_offset || !sizes) return fail(SERHIP_E_INVALID, "serhip_stat_files: NULL argument");
    parallel_for(n_files, threads, [&](int64_t i) {
        struct stat st;
        sizes[i] = (::stat(paths_blob + path_offset[i], &st) == 0 && S_ISREG(st.st_mode)) ? static_cast<int64_t>(st.st_size) : -1;
    });
    return SERHIP_OK;
}

int serhip_parallel_memcpy(void* dst, const void* src, int64_t bytes, int threads) {
    if (bytes < 0) return fail(SERHIP_E_INVALID, "serhip_parallel_memcpy: bytes < 0");
    if (bytes == 0) return SERHIP_OK;
    if (!dst || !src) return fail(SERHIP_E_INVALID, "serhip_parallel_memcpy: NULL argument");
    const int64_t piece = 4 << 20;   // 4 MiB pieces drawn from a counter
    const int64_t n = (bytes + piece - 1) / piece;
    parallel_for(n, threads, [&](int64_t i) {
        const int64_t at = i * piece;
        std::memcpy(static_cast<char*>(dst) + at, static_cast<const char*>(src) + at, static_cast<size_t>(std::min(piece, bytes - at)));
    });
    return SERHIP_OK;
}

int serhip_gather_f32(const uint64_t* src, const int64_t* src_len, const int64_t* dst_offset, int64_t n_arrays, float* dst,
                      int64_t dst_len, int threads, int32_t* all_finite) {
    if (n_arrays < 0 || dst_len < 0) return fail(SERHIP_E_INVALID, "serhip_gather_f32: negative count");
    if (!all_finite) return fail(SERHIP_E_INVALID, "serhip_gather_f32: all_finite is NULL");
    *all_finite = 1;
    if (n_arrays == 0) return SERHIP_OK;
    if (!src || !src_len || !dst_offset || !dst) return fail(SERHIP_E_INVALID, "serhip_gather_f32: NULL argument");
    // work items: pieces of at most 1 Mi floats, so that one hour-long array is shared by the team as 12 500 clips are
    constexpr int64_t kPiece = 1 << 20;
    struct Item {
        int64_t array, at, count;
    };
    std::vector<Item> items;
    items.reserve(static_cast<size_t>(n_arrays));
    for (int64_t i = 0; i < n_arrays; ++i) {
        if (src_len[i] < 0 || dst_offset[i] < 0 || dst_offset[i] > dst_len - src_len[i])
            return fail(SERHIP_E_INVALID, "serhip_gather_f32: region of array " + std::to_string(i) + " lies outside dst");
        if (src_len[i] > 0 && src[i] == 0) return fail(SERHIP_E_INVALID, "serhip_gather_f32: array " + std::to_string(i) + " is NULL");
        for (int64_t at = 0; at < src_len[i]; at += kPiece) items.push_back({i, at, std::min(kPiece, src_len[i] - at)});
    }
    std::atomic<uint32_t> bad{0};
    parallel_for(static_cast<int64_t>(items.size()), threads, [&](int64_t k) {
        const Item& it = items[static_cast<size_t>(k)];
        const float* from = reinterpret_cast<const float*>(static_cast<uintptr_t>(src[it.array])) + it.at;
        float* to = dst + dst_offset[it.array] + it.at;
        // copy and test block by block: the block just copied is still in this core's cache when it is tested
        constexpr int64_t kBlock = 8192;
        uint32_t seen = 0;
        for (int64_t b = 0; b < it.count; b += kBlock) {
            const int64_t m = std::min(kBlock, it.count - b);
            std::memcpy(to + b, from + b, static_cast<size_t>(m) * sizeof(float));
            uint32_t word;
            uint32_t acc = 0;
            for (int64_t j = 0; j < m; ++j) {   // exponent all ones = NaN or infinity
                std::memcpy(&word, from + b + j, sizeof(word));
                acc |= static_cast<uint32_t>((word & 0x7f800000u) == 0x7f800000u);
            }
            seen |= acc;
        }
        if (seen) bad.store(1, std::memory_order_relaxed);
    });
    if (bad.load()) *all_finite = 0;
    return SERHIP_OK;
}

int serhip_flac_probe(const void* bytes, int64_t n_bytes, int32_t* sample_rate, int32_t* channels, int32_t* bits_per_sample,
                      int64_t* total_frames) {
    if (!bytes || n_bytes < 0) return fail(SERHIP_E_INVALID, "serhip_flac_probe: NULL or negative-length input");
    flac::StreamInfo info;
    std::string err;
    if (!flac::parse_stream_info(static_cast<const uint8_t*>(bytes), static_cast<size_t>(n_bytes), &info, &err))
        return fail(SERHIP_E_INVALID, "serhip_flac_probe: " + err);
    if (sample_rate) *sample_rate = info.sample_rate;
    if (channels) *channels = info.channels;
    if (bits_per_sample) *bits_per_sample = info.bits;
    if (total_frames) *total_frames = info.total_frames;
    return SERHIP_OK;
}

int serhip_flac_decode(const void* bytes, int64_t n_bytes, int32_t* out, int64_t capacity_frames, int verify_md5,
                       int64_t* frames_decoded) {
    if (!bytes || n_bytes < 0 || !frames_decoded || (out && capacity_frames < 0))
        return fail(SERHIP_E_INVALID, "serhip_flac_decode: NULL or negative argument");
    flac::StreamInfo info;
    std::string err;
    const uint8_t* p = static_cast<const uint8_t*>(bytes);
    if (!flac::parse_stream_info(p, static_cast<size_t>(n_bytes), &info, &err))
        return fail(SERHIP_E_INVALID, "serhip_flac_decode: " + err);
    int64_t done = 0;
    try {
        if (!flac::decode(p, static_cast<size_t>(n_bytes), info, out, capacity_frames, verify_md5 != 0, &done, &err))
            return fail(SERHIP_E_INVALID, "serhip_flac_decode: " + err);
    } catch (const std::bad_alloc&) {
        return fail(SERHIP_E_NOMEM, "serhip_flac_decode: out of host memory");
    }
    *frames_decoded = done;
    return SERHIP_OK;
}

int64_t serhip_host_table(int sample_rate, int which, int arg, float* out, int64_t out_cap) {
    if (sample_rate <= 0) return fail(SERHIP_E_INVALID, "serhip_host_table: sample_rate must be positive");
    std::vector<float> v;
    switch (which) {
        case 0: {
            const std::vector<double> w = hann_periodic(kNfft);
            v.assign(w.begin(), w.end());
            break;
        }
        case 1:
            v = mel_filterbank(sample_rate, kNfft, kMels);
            break;
        case 2: {
            if (arg < 0 || arg >= kTunings) return fail(SERHIP_E_INVALID, "serhip_host_table: tuning index out of range");
            std::vector<float> bin_major(static_cast<size_t>(kBins) * kChroma, 0.0f);
            chroma_filterbank(sample_rate, kNfft, tuning_edge(arg), bin_major.data());
            v.resize(static_cast<size_t>(kChroma) * kBins);
            for (int c = 0; c < kChroma; ++c)
                for (int k = 0; k < kBins; ++k) v[static_cast<size_t>(c) * kBins + k] = bin_major[static_cast<size_t>(k) * kChroma + c];
            break;
        }
        case 3:
            v = dct_ortho(kMfcc, kMels);
            break;
        case 4: {
            int lo, hi;
            piptrack_bin_range(sample_rate, kNfft, &lo, &hi);
            v = {static_cast<float>(lo), static_cast<float>(hi)};
            break;
        }
        default:
            return fail(SERHIP_E_INVALID, "serhip_host_table: unknown table id");
    }
    const int64_t n = static_cast<int64_t>(v.size());
    if (!out) return n;
    if (out_cap < n) return fail(SERHIP_E_INVALID, "serhip_host_table: output buffer too small");
    std::memcpy(out, v.data(), v.size() * sizeof(float));
    return n;
}

}  // extern "C"
