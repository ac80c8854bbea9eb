// Host side of the ingest path: many RIFF/WAVE files read by a team of threads straight into one slab of
// (pinned) host memory, headers parsed in place, so that the slab can travel to the GPU in ONE copy and
// serhip_prepare_audio can convert / mix down / normalise the payloads there.
//
// Replaces, for PCM16 and float32 WAVE files, the per-file `librosa.load(path, sr=None, offset, duration)` of
// ser/_internal/utils/audio_utils.py:99-113 and the multiprocessing.Pool fan-out around it
// (ser/_internal/data/data_loader.py:368-379).  Everything this reader does not take -- other encodings, other
// containers, Git-LFS pointers, missing files, empty selections -- is reported per file with a status and left to
// the Python reader (ser_amd/audio_io.py), which produces the reference's retries and error messages.  No GPU
// code in this header.
#pragma once

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/serhip.h"

namespace serhip {

inline uint32_t rd_u32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24); }
inline uint16_t rd_u16(const unsigned char* p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }

// A team of worker threads that lives as long as the process: the loader calls into this header once per slab
// (about once a millisecond), and creating and joining sixteen threads each time costs a good part of that.
// Workers sleep on a condition variable between calls.  One call at a time owns the team (a second caller that
// finds it busy runs its loop on ad-hoc threads instead); a forked child starts a fresh team (threads do not
// survive fork).  Never destroyed: at process exit the sleeping workers simply go away with the process.
class Team {
  public:
    static constexpr int kMaxWorkers = 63;

    template <typename F>
    bool try_run(int64_t n, int threads, F& fn) {   // false: busy, the caller must do the work some other way
        std::unique_lock<std::mutex> owner(call_, std::try_to_lock);
        if (!owner.owns_lock()) return false;
        const int helpers = std::min(threads - 1, kMaxWorkers);
        {
            std::lock_guard<std::mutex> lk(m_);
            while (static_cast<int>(workers_.size()) < helpers) {
                const int index = static_cast<int>(workers_.size());
                workers_.emplace_back([this, index] { work(index); });
                workers_.back().detach();
            }
            next_.store(0, std::memory_order_relaxed);
            n_ = n;
            body_ = [&fn](int64_t i) { fn(i); };
            helpers_ = helpers;
            active_ = helpers;
            ++generation_;
        }
        wake_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return active_ == 0; });
        body_ = nullptr;
        return true;
    }

  private:
    void drain() {
        for (;;) {
            const int64_t i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_) return;
            body_(i);
        }
    }
    void work(int index) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                wake_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                if (index >= helpers_) continue;   // this call wants fewer threads
            }
            drain();
            std::lock_guard<std::mutex> lk(m_);
            if (--active_ == 0) done_.notify_one();
        }
    }
    std::mutex call_, m_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> workers_;
    std::function<void(int64_t)> body_;
    std::atomic<int64_t> next_{0};
    int64_t n_ = 0;
    int helpers_ = 0, active_ = 0;
    uint64_t generation_ = 0;
};

inline Team* process_team() {
    static std::mutex guard;
    static Team* team = nullptr;
    static pid_t owner = 0;
    std::lock_guard<std::mutex> lk(guard);
    const pid_t me = ::getpid();
    if (!team || owner != me) {   // first use, or a forked child: the parent's team (threads, locks) is left alone
        team = new Team();
        owner = me;
    }
    return team;
}

// run fn(i) for i in [0, n) on `threads` threads (work drawn from a counter: files differ in size)
template <typename F>
void parallel_for(int64_t n, int threads, F fn) {
    threads = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(threads, n)));
    if (threads == 1) {
        for (int64_t i = 0; i < n; ++i) fn(i);
        return;
    }
    if (process_team()->try_run(n, threads, fn)) return;
    std::atomic<int64_t> next{0};
    auto body = [&]() {
        for (;;) {
            const int64_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n) return;
            fn(i);
        }
    };
    std::vector<std::thread> team;
    team.reserve(static_cast<size_t>(threads - 1));
    for (int t = 1; t < threads; ++t) team.emplace_back(body);
    body();
    for (std::thread& t : team) t.join();
}

// int(float(seconds) * rate) of the Python reader (audio_io.read_raw_clip): truncation toward zero of a double product
inline int64_t frames_of_seconds(double seconds, int64_t rate) {
    const double frames = seconds * static_cast<double>(rate);
    if (!(frames < 4.0e18)) return static_cast<int64_t>(4.0e18);   // "past any file"; the conversion of a larger double is undefined
    return frames <= 0.0 ? 0 : static_cast<int64_t>(frames);
}

// What one WAVE file in the slab looks like once its header has been parsed.
struct WaveInSlab {
    int64_t body_at = 0;      // byte offset IN THE SLAB of the first frame of the data chunk (a multiple of the sample size)
    int64_t n_frames = 0;
    int64_t rate = 0;
    int channels = 0, block_align = 0, format = 0, sample_size = 0;
};

// One file -> its region [region, region + capacity) of the slab, header parsed.  Returns the status; fills `wave` on
// SERHIP_WAV_OK.
inline int load_wave(const char* path, unsigned char* slab, int64_t region, int64_t capacity, WaveInSlab* wave,
                     int read_threads = 1) {
    static const char kLfs[] = "version https://git-lfs.github.com/spec/v1";
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return SERHIP_WAV_IO;
    struct stat st;
    if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        ::close(fd);
        return SERHIP_WAV_IO;
    }
    if (st.st_size > capacity) {   // the file grew after the caller sized its region
        ::close(fd);
        return SERHIP_WAV_NOSPACE;
    }
    unsigned char* data = slab + region;
    int64_t got = 0;
    constexpr int64_t kPiece = 8 << 20;
    if (read_threads > 1 && st.st_size >= 4 * kPiece) {
        // a long recording in a batch of few files: its pieces are read by several threads (pread), not by one
        const int64_t pieces = (st.st_size + kPiece - 1) / kPiece;
        std::atomic<bool> failed{false};
        std::atomic<int64_t> total{0};
        parallel_for(pieces, read_threads, [&](int64_t p) {
            int64_t at = p * kPiece;
            const int64_t end = std::min<int64_t>(st.st_size, at + kPiece);
            while (at < end) {
                const ssize_t r = ::pread(fd, data + at, static_cast<size_t>(end - at), static_cast<off_t>(at));
                if (r < 0) {
                    failed.store(true);
                    return;
                }
                if (r == 0) return;   // the file shrank: what was read is what there is
                at += r;
                total.fetch_add(r);
            }
        });
        if (failed.load()) {
            ::close(fd);
            return SERHIP_WAV_IO;
        }
        got = total.load();           // (short only if the file shrank under the reader: the chunk walk below cuts there)
    } else {
        while (got < st.st_size) {
            const ssize_t r = ::read(fd, data + got, static_cast<size_t>(st.st_size - got));
            if (r < 0) {
                ::close(fd);
                return SERHIP_WAV_IO;
            }
            if (r == 0) break;
            got += r;
        }
    }
    ::close(fd);
    const int64_t len = got;
    if (len >= static_cast<int64_t>(sizeof(kLfs) - 1) && std::memcmp(data, kLfs, sizeof(kLfs) - 1) == 0) return SERHIP_WAV_LFS_POINTER;
    if (len < 12 || std::memcmp(data, "RIFF", 4) != 0 || std::memcmp(data + 8, "WAVE", 4) != 0) return SERHIP_WAV_NOT_WAVE;
    // chunk walk, same rules as the Python parser: the LAST fmt / data chunk wins, sizes padded to even, a data
    // chunk that runs past the end of the file is cut there
    int64_t pos = 12, fmt_at = -1, fmt_len = 0, body_at = -1, body_len = 0;
    while (pos + 8 <= len) {
        const int64_t size = rd_u32(data + pos + 4);
        if (std::memcmp(data + pos, "fmt ", 4) == 0) {
            fmt_at = pos + 8;
            fmt_len = std::min<int64_t>(size, len - pos - 8);
        } else if (std::memcmp(data + pos, "data", 4) == 0) {
            body_at = pos + 8;
            body_len = std::min<int64_t>(size, len - pos - 8);
        }
        pos += 8 + size + (size & 1);
    }
    if (fmt_at < 0 || body_at < 0 || fmt_len < 16) return SERHIP_WAV_NOT_WAVE;
    const unsigned char* fmt = data + fmt_at;
    int code = rd_u16(fmt);
    const int channels = rd_u16(fmt + 2);
    const int64_t rate = rd_u32(fmt + 4);
    const int block_align = rd_u16(fmt + 12), bits = rd_u16(fmt + 14);
    if (code == 0xFFFE && fmt_len >= 26) code = rd_u16(fmt + 24);   // WAVE_FORMAT_EXTENSIBLE: sub-format GUID
    if (channels == 0 || rate == 0 || block_align == 0) return SERHIP_WAV_NOT_WAVE;
    int format, ssize;
    if (code == 1 && bits == 16 && block_align == 2 * channels) {
        format = SERHIP_SAMPLE_S16;
        ssize = 2;
    } else if (code == 3 && bits == 32 && block_align == 4 * channels) {
        format = SERHIP_SAMPLE_F32;
        ssize = 4;
    } else {
        return SERHIP_WAV_HOST_DECODE;   // 8/24/32-bit PCM, float64, compressed: the host decoder's business
    }
    if (channels > 64 || rate > 0x7fffffff) return SERHIP_WAV_HOST_DECODE;
    const int64_t n_frames = body_len / block_align;
    int64_t at = region + body_at;
    if (at % ssize != 0) {
        // chunks are 2-byte aligned, float32 samples want 4: slide the whole payload down (the header in front of it
        // is not needed any more; every selection from this file then starts on a sample boundary)
        const int64_t aligned = at / ssize * ssize;
        std::memmove(slab + aligned, slab + at, static_cast<size_t>(n_frames * block_align));
        at = aligned;
    }
    wave->body_at = at;
    wave->n_frames = n_frames;
    wave->rate = rate;
    wave->channels = channels;
    wave->block_align = block_align;
    wave->format = format;
    wave->sample_size = ssize;
    return SERHIP_WAV_OK;
}

// The frames one job selects from a loaded file (NaN = not given), as the reference's segment reads do:
// first = (int64)(start * rate), count = (int64)(duration * rate), cut at the end of the file.
inline int select_clip(const WaveInSlab& wave, double start_seconds, double duration_seconds, serhip_wav_clip* clip) {
    int64_t first = std::isnan(start_seconds) ? 0 : frames_of_seconds(start_seconds, wave.rate);
    int64_t last = wave.n_frames;
    if (!std::isnan(duration_seconds)) {
        const int64_t count = frames_of_seconds(duration_seconds, wave.rate);
        last = std::min(wave.n_frames, first > wave.n_frames ? wave.n_frames : first + count);
    }
    first = std::min(first, wave.n_frames);
    const int64_t frames = std::max<int64_t>(0, last - first);
    if (frames == 0) return SERHIP_WAV_EMPTY;
    clip->raw_offset = wave.body_at + first * wave.block_align;
    clip->frames = frames;
    clip->sample_rate = static_cast<int32_t>(wave.rate);
    clip->channels = wave.channels;
    clip->sample_format = wave.format;
    return SERHIP_WAV_OK;
}

}  // namespace serhip
