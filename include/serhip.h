/*
 * serhip.h -- C ABI of libserhip.so, the MI355X (gfx950) implementation of the
 * jsugg/ser fast-profile acoustic feature path.
 *
 * The reference has no native ABI; its seam is a set of Python callables
 * (SURVEY.md section 8b).  Every entry point below names the reference
 * interface (file:line under /root/reference) whose arithmetic it replaces.
 * The Python package ser_amd binds these with ctypes (ser_amd/_native.py) and
 * re-exposes the reference's own signatures on top.
 *
 * Conventions
 *   - every function returns 0 on success or a negative SERHIP_E_* code;
 *     serhip_last_error() then holds a thread-local message.  No exceptions
 *     cross the boundary.
 *   - "dev" pointers are device (HBM) pointers on the context's GPU; "host"
 *     pointers are ordinary host memory.  The caller owns all buffers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *     Work is enqueued asynchronously; outputs are valid once the stream has
 *     been synchronised.  Host descriptor arrays may be reused as soon as the
 *     call returns.  Calls on one context share its device workspace: on one
 *     stream they are ordered by the stream; a call on another stream than the
 *     previous call's first waits on the device (hipStreamWaitEvent) for that
 *     call's kernels, so streams may be switched between calls without a
 *     host synchronisation.
 *   - a serhip_ctx is bound to one device and is not thread-safe; distinct
 *     contexts are independent.  Handles are opaque.
 */
#ifndef SERHIP_H
#define SERHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SERHIP_OK 0
#define SERHIP_E_INVALID (-1)   /* bad argument (NULL, range, shape)            */
#define SERHIP_E_HIP (-2)       /* a HIP runtime call failed                    */
#define SERHIP_E_NOMEM (-3)     /* workspace allocation failed                  */
#define SERHIP_E_UNSUPPORTED (-4) /* valid request outside the implemented path */

/* Feature-group bits; order of the output blocks is fixed: mfcc | chroma | mel |
 * contrast, as in ser/_internal/utils/dsp.py:106-151 and
 * ser/_internal/repr/handcrafted.py:45-59 (FeatureFlags, config/schema.py:219-227).
 * Tonnetz (6 dims) is outside the path. */
#define SERHIP_FEAT_MFCC 1u      /* 40 dims */
#define SERHIP_FEAT_CHROMA 2u    /* 12 dims */
#define SERHIP_FEAT_MEL 4u       /* 128 dims */
#define SERHIP_FEAT_CONTRAST 8u  /* 7 dims (identically 0: reference quirk) */

typedef struct serhip_ctx serhip_ctx;
typedef struct serhip_plan serhip_plan;

/* Library / ABI version (major*10000 + minor*100 + patch). */
int serhip_version(void);

/* Thread-local message of the last failing call on this thread ("" if none). */
const char* serhip_last_error(void);

/* Create / destroy a context bound to HIP device `device_id`.
 * No reference counterpart (the reference runs on the host). */
int serhip_create(int device_id, serhip_ctx** out);
void serhip_destroy(serhip_ctx* ctx);

/* Build the constant tables (periodic Hann window, FFT twiddles, Slaney mel
 * filterbank, the 101 chroma filterbanks one per possible tuning estimate,
 * ortho DCT-II) for one sample rate and upload them.
 * Replaces the per-call table construction inside librosa that
 * ser/_internal/utils/dsp.py:100-136 triggers (filters.mel, filters.chroma,
 * get_window, scipy.fft.dct), done once per (sample_rate, flags). */
int serhip_plan_fast(serhip_ctx* ctx, int sample_rate, uint32_t feature_flags, serhip_plan** out);
void serhip_plan_destroy(serhip_plan* plan);

/* Output width D of a plan: 40*mfcc + 12*chroma + 128*mel + 7*contrast
 * (HandcraftedBackend.feature_dim, repr/handcrafted.py:45-59). */
int serhip_plan_feature_dim(const serhip_plan* plan);

/* THE HOT PATH.  One feature vector per segment:
 *   out_dev[i, :] = extract_feature_from_signal(audio[seg_offset[i] : seg_offset[i]+seg_len[i]], sr)
 * i.e. ser/_internal/utils/dsp.py:67-151 (pad to >=512 samples, n_fft = min(N, 2048),
 * STFT, mel / log-mel+DCT MFCC / tuning-corrected chroma / contrast, time-mean), batched
 * over `n_seg` independent segments that may overlap in `audio_dev` (sliding windows of
 * HandcraftedBackend.encode_sequence, repr/handcrafted.py:85-97, need no copies).
 *   audio_dev   device, f32 mono samples, finite (the caller validates, dsp.py:94-95)
 *   n_audio     number of samples in audio_dev; every segment must lie inside it
 *   seg_offset  host, [n_seg] first sample of each segment
 *   seg_len     host, [n_seg] samples per segment, >= 1
 *   out_dev     device, [n_seg, D] row-major f32
 *   tuning_dev  device, [n_seg] f32 chroma tuning estimate per segment, or NULL
 * Segments with >= 2048 samples run the LDS-staged radix FFT kernels; shorter ones
 * (n_fft = N, any integer 512..2047) run a direct double-precision DFT kernel. */
int serhip_fast_features(serhip_plan* plan, const float* audio_dev, int64_t n_audio,
                         const int64_t* seg_offset, const int64_t* seg_len, int64_t n_seg,
                         float* out_dev, float* tuning_dev, void* stream);

/* serhip_fast_features for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists
 * for tight serving loops, real-time threads and harnesses that forbid hipMalloc.  It is NOT capturable:
 * do not call it on a stream that is under HIP graph capture -- every call uploads its descriptors from
 * one reused pinned staging buffer (a replayed graph would upload whatever the last call left there) and,
 * from the second call on, waits on the host for the previous upload's event.  Same
 * arguments, same kernels and bit-identical results as serhip_fast_features, but every byte of scratch
 * (descriptors, work units, magnitudes, peaks, partial sums) lives in a file-scope `static __device__`
 * array of SERHIP_NOALLOC_SCRATCH_BYTES that each GPU the library runs on holds from code-object load
 * until the process ends; its device address is resolved once per context in serhip_create
 * (hipGetSymbolAddress).  The call itself performs no hipMalloc / hipFree / hipHostMalloc / event
 * creation and no host heap allocation on the success path (the host plan's buffers were reserved in
 * serhip_create); per-kernel profiling (serhip_profile_*) does not record these launches.
 * Limits (SERHIP_E_UNSUPPORTED with a message otherwise): at most SERHIP_NOALLOC_MAX_SEGMENTS segments
 * and SERHIP_NOALLOC_MAX_FRAME_SLOTS STFT frame slots per call (a segment of >= 2048 samples takes
 * 1 + len/512 frames rounded up to a multiple of 16: a 3 s window at 16 kHz is 96 slots, so 64 such
 * windows fit), and the byte total of the planned scratch must fit the array.
 * All calls on one GPU share that GPU's array, which keeps the previous call's contents (what must start
 * at zero is cleared by the call): calls through ONE context are ordered by the library like users of
 * its workspace (same stream: stream order; another stream: a device-side event wait), but calls
 * through different contexts on the same GPU that overlap in time race. */
#define SERHIP_NOALLOC_SCRATCH_BYTES (48u << 20)
#define SERHIP_NOALLOC_MAX_SEGMENTS 256
#define SERHIP_NOALLOC_MAX_FRAME_SLOTS 6144
int serhip_fast_features_noalloc(serhip_plan* plan, const float* audio_dev, int64_t n_audio,
                                 const int64_t* seg_offset, const int64_t* seg_len, int64_t n_seg,
                                 float* out_dev, float* tuning_dev, void* stream);

/* Segment statistics pooling over contiguous frame ranges [win_lo[i], win_hi[i]):
 *   mode 0: out[i] = [mean | population std] in float64, width 2*D
 *           (mean_std_pool, ser/_internal/pool/stats_pool.py:15-43)
 *   mode 1: out[i] = mean accumulated in float32 then widened, width D
 *           (HandcraftedBackend.pool, repr/handcrafted.py:109-122)
 * The ranges are what overlap_frame_mask (repr/backend.py:81-111) selects for
 * non-decreasing frame times; the host computes them.
 *   emb_dev  device, [n_frames, dim] row-major f32
 *   win_lo/win_hi host, [n_win] int32, 0 <= lo < hi <= n_frames
 *   out_dev  device, [n_win, 2*dim] (mode 0) or [n_win, dim] (mode 1) float64 */
int serhip_pool_windows(serhip_ctx* ctx, const float* emb_dev, int32_t n_frames, int32_t dim,
                        const int32_t* win_lo, const int32_t* win_hi, int32_t n_win, int mode,
                        double* out_dev, void* stream);

/* Whisper-style log-mel front end of the transcript lane:
 *   whisper.audio.log_mel_spectrogram(audio, n_mels, padding=0)
 * (called at ser/_internal/transcript/backends/stable_whisper_mps_compat.py:192-201):
 * reflect-padded STFT n_fft 400 / hop 160 / periodic Hann, last frame dropped, |X|^2,
 * Slaney mel (16 kHz), log10(max(.,1e-10)), clamp to max-8 per clip, (x+4)/4.
 *   audio_dev device, [n_clips, n_samples] f32 (each clip processed independently)
 *   out_dev   device, [n_clips, n_mels, n_samples/160] f32 (integer division: torch.stft
 *             yields 1 + n_samples/160 columns and whisper drops the last)
 * n_mels is 80 or 128; n_samples must be > 200 (reflect padding of 200 per side). */
int serhip_whisper_logmel(serhip_ctx* ctx, const float* audio_dev, int32_t n_clips,
                          int64_t n_samples, int32_t n_mels, float* out_dev, void* stream);

/* Device-side audio preparation: replaces _prepare_audio_buffer / _to_mono / _normalize_audio
 * (ser/_internal/utils/audio_utils.py:28-60) and the sample conversion libsndfile does inside
 * librosa.load (audio_utils.py:104-109) for a batch of decoded-but-raw clips, so that the host
 * only parses file headers and copies payload bytes.
 *   raw_dev        device, the interleaved sample payloads of the clips (any layout; see offsets)
 *   sample_format  SERHIP_SAMPLE_S16 (little-endian PCM16, value / 32768) or SERHIP_SAMPLE_F32
 *   clip_raw_offset host, byte offset of each clip's first frame in raw_dev (multiple of the
 *                  sample size), clip_frames host, sample frames per clip (>= 1),
 *   clip_channels  host, channels per clip (1..64), clip_out_offset host, first output sample
 *   out_dev        device f32: per clip `frames` mono samples: non-finite -> 0, float32 channel
 *                  mean (numpy summation order), divided by the clip's max |x| (all-zero stays 0)
 * Results are bit-identical to the numpy reference code.  Every clip must lie inside
 * [0, raw_bytes) and [0, out_samples). */
#define SERHIP_SAMPLE_S16 0
#define SERHIP_SAMPLE_F32 1
int serhip_prepare_audio(serhip_ctx* ctx, const void* raw_dev, int64_t raw_bytes, int sample_format,
                         const int64_t* clip_raw_offset, const int64_t* clip_frames,
                         const int32_t* clip_channels, const int64_t* clip_out_offset, int64_t n_clips,
                         float* out_dev, int64_t out_samples, void* stream);

/* The fast profile's classifier on the device: forward pass of the sklearn
 * Pipeline(StandardScaler, MLPClassifier) built by ser/_internal/models/training_support.py:87-106,
 * i.e. model.predict / model.predict_proba as called at ser/_internal/models/fast_path.py:180-192
 * (SURVEY.md 8f item 3).  All arithmetic in float64, as sklearn does on the reference's matrix.
 *   x_dev        device, [n_rows, n_features] row-major, float32 (x_is_f64 = 0) or float64 (1)
 *   mean_dev / scale_dev  device f64 [n_features] (StandardScaler.mean_ / scale_) or NULL
 *   n_layers     weight matrices (hidden layers + 1); layer_sizes host [n_layers + 1] =
 *                {n_features, hidden..., n_out}; weights_dev / bias_dev host arrays of n_layers
 *                device pointers: coefs_[l] as [layer_sizes[l], layer_sizes[l+1]] row-major f64,
 *                intercepts_[l] as [layer_sizes[l+1]] f64
 *   hidden_activation  0 identity, 1 relu, 2 tanh, 3 logistic (MLPClassifier.activation)
 *   out_activation     0 softmax (n_out >= 2 classes), 1 logistic (n_out == 1: two classes)
 *   proba_dev    device f64 [n_rows, n_classes], n_classes = n_out (softmax) or 2 (logistic)
 *   label_dev    device int32 [n_rows] index into classes_ (arg-max, first on ties), or NULL */
int serhip_mlp_predict(serhip_ctx* ctx, const void* x_dev, int x_is_f64, int64_t n_rows, int32_t n_features,
                       const double* mean_dev, const double* scale_dev, int32_t n_layers,
                       const int32_t* layer_sizes, const double* const* weights_dev,
                       const double* const* bias_dev, int hidden_activation, int out_activation,
                       double* proba_dev, int32_t* label_dev, void* stream);

/* Fan-in of the per-GPU feature shards: one process per GPU, contiguous shards by segment, ONE collective at
 * the end.  Replaces the result collection of the reference's multiprocessing.Pool over files
 * (ser/_internal/data/data_loader.py:368-379, pool.imap_unordered -> list) for a multi-GPU host that does not go
 * through torch.distributed (ser_amd/distributed.py is the torch form of the same gather).  RCCL is loaded on
 * first use (dlopen of librccl.so.1; libserhip.so has no link-time dependency on it).
 *   serhip_comm_unique_id   rank 0 calls it and hands the SERHIP_COMM_ID_BYTES bytes to the other ranks by any
 *                           means the host has (a file, MPI, a socket)
 *   serhip_comm_create      collective over all `world` ranks: joins the communicator on the context's GPU
 *   serhip_allgather_rows   out_dev[sum(rows_per_rank), dim] = the shards [rows_per_rank[r], dim] of all ranks in
 *                           rank order, on every rank; rows_per_rank is a HOST array [world] (shards may differ in
 *                           size, and may be empty); shard_dev holds this rank's rows_per_rank[rank] rows.
 *                           Enqueued on `stream`: ONE ncclAllGather when all shards have the same non-zero row
 *                           count, otherwise one grouped call of a broadcast per non-empty shard. */
#define SERHIP_COMM_ID_BYTES 128
typedef struct serhip_comm serhip_comm;
int serhip_comm_unique_id(unsigned char* id_out);
int serhip_comm_create(serhip_ctx* ctx, const unsigned char* id, int rank, int world, serhip_comm** out);
int serhip_allgather_rows(serhip_comm* comm, const float* shard_dev, const int64_t* rows_per_rank, int32_t dim,
                          float* out_dev, void* stream);
void serhip_comm_destroy(serhip_comm* comm);

/* Per-kernel timing with HIP events recorded on the stream each kernel is launched on.
 * No reference counterpart (the reference's only timer is the phase stopwatch of
 * ser/_internal/runtime/phase_timing.py:29-78 around whole inferences); bench.py uses
 * it for the roofline line.  While enabled, every launch made by
 * serhip_fast_features is bracketed by an event pair; serhip_profile_read waits
 * for the recorded launches of one kind, returns their summed duration and
 * count, and forgets them. */
#define SERHIP_KERNEL_STFT_FRAMES 0     /* k_stft_frames    (framing+FFT+|X|+peaks)            */
#define SERHIP_KERNEL_SEGMENT_FINISH 1  /* k_segment_finish (tuning / MFCC + time means)        */
#define SERHIP_KERNEL_DFT_FRAMES 2      /* k_dft_frames     (segments < 2048 samples)           */
#define SERHIP_KERNEL_PROJECT_FRAMES 3  /* k_project_frames (mel + chroma projections on MFMA)  */
int serhip_profile_enable(serhip_ctx* ctx, int on);
int serhip_profile_read(serhip_ctx* ctx, int kind, double* total_ms, int64_t* launches);

/* Diagnostic hook: the tuning estimate alone -- librosa.estimate_tuning's median threshold over the
 * piptrack peaks followed by librosa.pitch_tuning's 100-bin histogram arg-max, reached from
 * ser/_internal/utils/dsp.py:115 (chroma_stft) -- computed by the same kernel the hot path uses
 * (k_segment_finish<1>) from peak lists supplied by the caller instead of by the FFT kernel.  Lets tests
 * drive the discrete decisions (even-count median, ties, empty sets, more peaks than fit in LDS) with
 * hand-derived answers.  Synchronous; all pointers are HOST pointers.
 *   pitch_mag     [sum(frame_counts)][2] float32 (pitch in Hz, magnitude), frame after frame
 *   frame_counts  [n_frames] peaks per STFT frame, each 0..1023 (the count field of the scratch word);
 *                 n_frames x max(count) is bounded at 2^24 peak slots
 *   tuning_out    one float32: the estimate in fractions of a semitone, in [-0.5, 0.5) */
int serhip_tuning_from_peaks(serhip_plan* plan, const float* pitch_mag, const int32_t* frame_counts,
                             int32_t n_frames, float* tuning_out);

/* ---- ingest: files and host buffers on their way to the GPU (host-only code: no GPU touched) ----
 *
 * serhip_read_wav_batch reads `n_files` RIFF/WAVE files with a team of threads straight into caller-owned host
 * memory `dst` (meant to be pinned, so that the slab goes to the GPU in one asynchronous copy) and parses the
 * headers in place; serhip_prepare_audio then converts / mixes down / normalises the payloads on the device.
 * Together they replace, for PCM16 and float32 WAVE files, the per-file librosa.load(path, sr=None, offset,
 * duration) of ser/_internal/utils/audio_utils.py:99-113 and the multiprocessing.Pool over files around it
 * (ser/_internal/data/data_loader.py:368-379: worker processes decoding while others extract).
 *   paths_blob / path_offset  the paths, NUL-terminated, back to back; path i starts at paths_blob + path_offset[i]
 *   start_seconds / duration_seconds  [n_files] or NULL; NaN = not given (whole file).  Frame range as in the
 *                 reference's segment reads: first = (int64)(start * rate), count = (int64)(duration * rate)
 *   dst_offset / dst_capacity  [n_files] the region of `dst` each WHOLE file is read into (header included);
 *                 regions should start 16-byte aligned; capacity >= the file's size.  Regions must not overlap, with
 *                 one exception that is the point of it: CONSECUTIVE entries that name the same path and the same
 *                 region are segments of one recording -- the file is read once and every entry selects its own
 *                 frames from it (a manifest with thousands of utterances of one long file costs one read)
 *   clips         [n_files] out: per file a status and, for SERHIP_WAV_OK, where its selected frames sit in `dst`
 *                 (raw_offset is a multiple of the sample size: what serhip_prepare_audio's clip_raw_offset takes)
 * Every status other than SERHIP_WAV_OK means "not handled here": the caller's own reader (ser_amd/audio_io.py)
 * then decodes the file on the host or raises the reference's error for it (FileNotFoundError, AudioIntegrityError
 * for Git-LFS pointers, AudioDecodeError, "contains no samples").  Returns 0, or SERHIP_E_INVALID for bad arguments. */
#define SERHIP_WAV_OK 0
#define SERHIP_WAV_HOST_DECODE 1   /* a WAVE file in an encoding the device path does not take (8/24/32-bit PCM, float64, ...) */
#define SERHIP_WAV_IO 2            /* open / fstat / read failed, or not a regular file */
#define SERHIP_WAV_LFS_POINTER 3   /* a Git-LFS pointer in place of audio bytes */
#define SERHIP_WAV_NOT_WAVE 4      /* not RIFF/WAVE, or no usable fmt / data chunk */
#define SERHIP_WAV_NOSPACE 5       /* the file is larger than its region */
#define SERHIP_WAV_EMPTY 6         /* the selection holds no frames */
typedef struct serhip_wav_clip {
    int64_t raw_offset;     /* byte offset in dst of the first selected frame */
    int64_t frames;         /* selected sample frames */
    int32_t sample_rate;
    int32_t channels;
    int32_t sample_format;  /* SERHIP_SAMPLE_S16 or SERHIP_SAMPLE_F32 */
    int32_t status;         /* SERHIP_WAV_* */
} serhip_wav_clip;
int serhip_read_wav_batch(const char* paths_blob, const int64_t* path_offset, int64_t n_files,
                          const double* start_seconds, const double* duration_seconds,
                          const int64_t* dst_offset, const int64_t* dst_capacity, void* dst, int64_t dst_bytes,
                          int threads, serhip_wav_clip* clips);

/* Sizes of many files with a team of threads (one stat each; -1 = missing or not a regular file): what the slab
 * planner of the loader needs before it can give every file its region.  Replaces nothing in the reference (its
 * workers open files one by one); exists so that the per-file cost of a 100k-file corpus is not Python's. */
int serhip_stat_files(const char* paths_blob, const int64_t* path_offset, int64_t n_files, int threads, int64_t* sizes);

/* FLAC decoding on the host (no GPU touched): the reference reads .flac through librosa.load -> soundfile, with an
 * explicit soundfile fallback (ser/_internal/utils/audio_utils.py:99-154); this library decodes the stream itself
 * (ser_amd/csrc/flac_decoder.hpp, RFC 9639: every subframe type, both Rice code books, the stereo modes, wasted bits,
 * variable block size, leading ID3v2 tags).  Lossless, so the result is the encoder's PCM bit for bit; every frame's
 * CRC-8 / CRC-16 is checked and, with verify_md5 != 0, the MD5 signature of STREAMINFO.  A stream that fails a check
 * is an error (SERHIP_E_INVALID with a message), never a partial result.
 *   serhip_flac_probe   STREAMINFO: sample rate, channels (1..8), bits per sample (4..32), frames (0 = unknown)
 *   serhip_flac_decode  out = interleaved int32 samples [capacity_frames][channels] (sign-extended, NOT scaled), or
 *                       NULL to count the frames of a stream whose STREAMINFO does not say; *frames_decoded on return */
int serhip_flac_probe(const void* bytes, int64_t n_bytes, int32_t* sample_rate, int32_t* channels, int32_t* bits_per_sample,
                      int64_t* total_frames);
int serhip_flac_decode(const void* bytes, int64_t n_bytes, int32_t* out, int64_t capacity_frames, int verify_md5,
                       int64_t* frames_decoded);

/* memcpy of one large host buffer by a team of threads: pageable caller memory (a numpy array) -> pinned staging
 * at more than one core's copy bandwidth, so that staging keeps up with the H2D DMA behind it (the reference has
 * no counterpart: its arrays never leave the host). */
int serhip_parallel_memcpy(void* dst, const void* src, int64_t bytes, int threads);

/* The same team gathering `n_arrays` float32 arrays (array i: `src_len[i]` floats at `src[i]`, a host address passed
 * as an integer) into one buffer -- array i lands at `dst + dst_offset[i]` floats; regions must lie inside
 * [0, dst_len) and are the caller's to keep apart -- and checking, in the same pass over the bytes, that every sample
 * is finite: `*all_finite` = 1, or 0 when a NaN or an infinity was copied (the copy is complete either way).
 * Replaces, for a batch of clips held as separate numpy arrays, the per-clip `np.isfinite(...).all()` of the
 * reference's input validation (ser/_internal/utils/dsp.py:93-95) and the `np.concatenate` a batched call would
 * otherwise need: neither pass over the samples remains on the host. */
int serhip_gather_f32(const uint64_t* src, const int64_t* src_len, const int64_t* dst_offset, int64_t n_arrays, float* dst,
                      int64_t dst_len, int threads, int32_t* all_finite);

/* ---- host-only helpers (no GPU touched; used by the CPU test-suite) ---- */

/* Copy one host-built constant table of the fast plan into `out` (f32).
 *   which: 0 window[2048]   1 mel filterbank [128,1025]   2 chroma filterbank for
 *          tuning index `arg` (0..100) [12,1025]   3 DCT [40,128]
 *          4 piptrack bin range {kmin,kmax} as 2 floats
 * Returns the element count written (or needed if out is NULL), <0 on error. */
int64_t serhip_host_table(int sample_rate, int which, int arg, float* out, int64_t out_cap);

/* Frame bookkeeping of the STFT used by every kernel: number of STFT columns
 * for a segment of n_samples (after the >=512 padding) -- librosa.stft center=True,
 * hop 512 for n_fft 2048.  Exposed so hosts can size buffers / tests can pin it. */
int64_t serhip_frame_count(int64_t n_samples);

#ifdef __cplusplus
}
#endif
#endif /* SERHIP_H */
