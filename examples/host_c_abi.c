/* A host written in plain C over the C ABI (include/serhip.h): no Python, no torch.
 *
 * What a maintainer of a compiled host would write to put the fast-profile features of jsugg/ser on an MI355X:
 * read WAVE files into one host buffer with the library's reader team, copy the raw bytes to the GPU, prepare
 * (mix-down, scrub, peak-normalise: ser/_internal/utils/audio_utils.py:28-60) and featurise
 * (ser/_internal/utils/dsp.py:67-151) every whole file in one batch, print one line per file.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/host_c_abi.c \
 *       -o examples/host_c_abi ser_amd/libserhip.so -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../ser_amd' -Wl,-rpath,/opt/rocm/lib
 *   examples/host_c_abi a.wav b.wav ...         (PCM16 or float32 WAVE; one sample rate per run)
 *
 * Output: "<path> <frames> <rate> <D> v0 v1 ... v(D-1)" with %.9g values (float32 round-trips), or
 * "<path> SKIPPED status=<n>" for a file the device path does not take (include/serhip.h: SERHIP_WAV_*).
 * __graft_entry__.build() compiles it; tests/test_gpu_parity.py runs it against the Python host's rows. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "serhip.h"

#define CHECK_HIP(call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                    \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)
#define CHECK_SERHIP(call)                                                                \
    do {                                                                                  \
        int rc_ = (call);                                                                 \
        if (rc_ < 0) {                                                                    \
            fprintf(stderr, "%s: error %d: %s\n", #call, rc_, serhip_last_error());       \
            return 3;                                                                     \
        }                                                                                 \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s file.wav [file.wav ...]\n", argv[0]);
        return 1;
    }
    const int64_t n = argc - 1;
    /* the path table the reader takes: NUL-terminated paths back to back, and where each starts */
    size_t blob_len = 0;
    for (int64_t i = 0; i < n; ++i) blob_len += strlen(argv[1 + i]) + 1;
    char* blob = (char*)malloc(blob_len);
    int64_t* path_at = (int64_t*)malloc(sizeof(int64_t) * n);
    int64_t* sizes = (int64_t*)malloc(sizeof(int64_t) * n);
    int64_t* region = (int64_t*)malloc(sizeof(int64_t) * n);
    int64_t* room = (int64_t*)malloc(sizeof(int64_t) * n);
    serhip_wav_clip* clips = (serhip_wav_clip*)calloc((size_t)n, sizeof(serhip_wav_clip));
    size_t at = 0;
    for (int64_t i = 0; i < n; ++i) {
        path_at[i] = (int64_t)at;
        strcpy(blob + at, argv[1 + i]);
        at += strlen(argv[1 + i]) + 1;
    }
    CHECK_SERHIP(serhip_stat_files(blob, path_at, n, 4, sizes));
    int64_t total = 0;
    for (int64_t i = 0; i < n; ++i) {
        region[i] = total;
        room[i] = ((sizes[i] > 0 ? sizes[i] : 0) + 15) / 16 * 16;
        total += room[i];
    }
    if (total == 0) total = 16;
    unsigned char* slab = NULL;   /* pinned: the H2D copy below is a plain DMA */
    CHECK_HIP(hipHostMalloc((void**)&slab, (size_t)total, hipHostMallocDefault));
    CHECK_SERHIP(serhip_read_wav_batch(blob, path_at, n, NULL, NULL, region, room, slab, total, 4, clips));

    /* the device path takes one (sample rate, sample format) per batch: the first readable file sets both */
    int rate = 0, format = -1;
    int64_t n_ok = 0, total_frames = 0;
    int64_t* raw_offset = (int64_t*)malloc(sizeof(int64_t) * n);
    int64_t* frames = (int64_t*)malloc(sizeof(int64_t) * n);
    int32_t* channels = (int32_t*)malloc(sizeof(int32_t) * n);
    int64_t* out_offset = (int64_t*)malloc(sizeof(int64_t) * n);
    int64_t* which = (int64_t*)malloc(sizeof(int64_t) * n);
    for (int64_t i = 0; i < n; ++i) {
        if (clips[i].status != SERHIP_WAV_OK) continue;
        if (rate == 0) {
            rate = clips[i].sample_rate;
            format = clips[i].sample_format;
        }
        if (clips[i].sample_rate != rate || clips[i].sample_format != format) {
            clips[i].status = 100;   /* another batch's business */
            continue;
        }
        raw_offset[n_ok] = clips[i].raw_offset;
        frames[n_ok] = clips[i].frames;
        channels[n_ok] = clips[i].channels;
        out_offset[n_ok] = total_frames;
        which[n_ok] = i;
        total_frames += clips[i].frames;
        ++n_ok;
    }
    int dim = 0;
    float* host_out = NULL;
    if (n_ok > 0) {
        serhip_ctx* ctx = NULL;
        serhip_plan* plan = NULL;
        CHECK_SERHIP(serhip_create(0, &ctx));
        CHECK_SERHIP(serhip_plan_fast(ctx, rate, SERHIP_FEAT_MFCC | SERHIP_FEAT_CHROMA | SERHIP_FEAT_MEL | SERHIP_FEAT_CONTRAST, &plan));
        dim = serhip_plan_feature_dim(plan);
        unsigned char* raw_dev = NULL;
        float *audio_dev = NULL, *out_dev = NULL;
        CHECK_HIP(hipMalloc((void**)&raw_dev, (size_t)total));
        CHECK_HIP(hipMalloc((void**)&audio_dev, sizeof(float) * (size_t)total_frames));
        CHECK_HIP(hipMalloc((void**)&out_dev, sizeof(float) * (size_t)(n_ok * dim)));
        hipStream_t stream;
        CHECK_HIP(hipStreamCreate(&stream));
        CHECK_HIP(hipMemcpyAsync(raw_dev, slab, (size_t)total, hipMemcpyHostToDevice, stream));
        CHECK_SERHIP(serhip_prepare_audio(ctx, raw_dev, total, format, raw_offset, frames, channels, out_offset, n_ok,
                                          audio_dev, total_frames, stream));
        /* whole files: segment i is clip i of the prepared buffer */
        CHECK_SERHIP(serhip_fast_features(plan, audio_dev, total_frames, out_offset, frames, n_ok, out_dev, NULL, stream));
        host_out = (float*)malloc(sizeof(float) * (size_t)(n_ok * dim));
        CHECK_HIP(hipMemcpyAsync(host_out, out_dev, sizeof(float) * (size_t)(n_ok * dim), hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_HIP(hipStreamDestroy(stream));
        CHECK_HIP(hipFree(out_dev));
        CHECK_HIP(hipFree(audio_dev));
        CHECK_HIP(hipFree(raw_dev));
        serhip_plan_destroy(plan);
        serhip_destroy(ctx);
    }
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (k < n_ok && which[k] == i) {
            printf("%s %lld %d %d", argv[1 + i], (long long)frames[k], rate, dim);
            for (int d = 0; d < dim; ++d) printf(" %.9g", host_out[k * dim + d]);
            printf("\n");
            ++k;
        } else {
            printf("%s SKIPPED status=%d\n", argv[1 + i], clips[i].status);
        }
    }
    CHECK_HIP(hipHostFree(slab));
    free(host_out); free(which); free(out_offset); free(channels); free(frames); free(raw_offset);
    free(clips); free(room); free(region); free(sizes); free(path_at); free(blob);
    return 0;
}
