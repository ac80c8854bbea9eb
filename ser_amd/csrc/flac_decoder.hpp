// Host-only FLAC stream decoder (RFC 9639) behind serhip_flac_probe / serhip_flac_decode (include/serhip.h).
//
// The reference reads every container through librosa.load -> soundfile (libsndfile), with an explicit soundfile
// fallback (ser/_internal/utils/audio_utils.py:99-154); this package has neither.  FLAC is lossless, so "parity" for
// this decoder is bit-exactness of the PCM, and the stream carries its own checks: a CRC-8 over every frame header, a
// CRC-16 over every frame, and the MD5 of the decoded PCM in STREAMINFO.  All three are verified here; a stream that
// fails one is refused, never half-decoded.
//
// Supported: 1..8 channels, 4..32 bits per sample, fixed and variable block size, every subframe type (constant,
// verbatim, fixed order 0..4, LPC order 1..32), both Rice code book sizes with escape partitions, wasted bits, the
// three stereo decorrelation modes, leading ID3v2 tags, any metadata blocks (skipped).  Not supported: Ogg-encapsulated
// FLAC, resynchronisation after a damaged frame (a damaged file is an error).
#pragma once

#include <array>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace serhip {
namespace flac {

struct StreamInfo {
    int sample_rate = 0;
    int channels = 0;
    int bits = 0;
    int min_block = 0, max_block = 0;
    int64_t total_frames = 0;   // samples per channel; 0 = unknown
    uint8_t md5[16] = {};
    bool has_md5 = false;
    size_t audio_at = 0;        // byte offset of the first frame
};

// ---- MD5 (RFC 1321), incremental
class Md5 {
public:
    Md5() { reset(); }
    void reset() {
        a_ = 0x67452301u; b_ = 0xefcdab89u; c_ = 0x98badcfeu; d_ = 0x10325476u;
        len_ = 0; fill_ = 0;
    }
    void update(const uint8_t* p, size_t n) {
        len_ += n;
        if (fill_) {
            const size_t take = n < 64 - fill_ ? n : 64 - fill_;
            std::memcpy(buf_ + fill_, p, take);
            fill_ += take; p += take; n -= take;
            if (fill_ < 64) return;
            block(buf_);
            fill_ = 0;
        }
        for (; n >= 64; p += 64, n -= 64) block(p);
        if (n) { std::memcpy(buf_, p, n); fill_ = n; }
    }
    void finish(uint8_t out[16]) {
        const uint64_t bits = len_ * 8;
        const uint8_t pad = 0x80;
        update(&pad, 1);
        const uint8_t zero = 0;
        while (fill_ != 56) update(&zero, 1);
        uint8_t tail[8];
        for (int i = 0; i < 8; ++i) tail[i] = static_cast<uint8_t>(bits >> (8 * i));
        update(tail, 8);
        const uint32_t w[4] = {a_, b_, c_, d_};
        for (int i = 0; i < 16; ++i) out[i] = static_cast<uint8_t>(w[i >> 2] >> (8 * (i & 3)));
    }

private:
    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0xd76aa478u, 0xe8c7b756u, 0x242070dbu, 0xc1bdceeeu, 0xf57c0fafu, 0x4787c62au, 0xa8304613u, 0xfd469501u,
            0x698098d8u, 0x8b44f7afu, 0xffff5bb1u, 0x895cd7beu, 0x6b901122u, 0xfd987193u, 0xa679438eu, 0x49b40821u,
            0xf61e2562u, 0xc040b340u, 0x265e5a51u, 0xe9b6c7aau, 0xd62f105du, 0x02441453u, 0xd8a1e681u, 0xe7d3fbc8u,
            0x21e1cde6u, 0xc33707d6u, 0xf4d50d87u, 0x455a14edu, 0xa9e3e905u, 0xfcefa3f8u, 0x676f02d9u, 0x8d2a4c8au,
            0xfffa3942u, 0x8771f681u, 0x6d9d6122u, 0xfde5380cu, 0xa4beea44u, 0x4bdecfa9u, 0xf6bb4b60u, 0xbebfbc70u,
            0x289b7ec6u, 0xeaa127fau, 0xd4ef3085u, 0x04881d05u, 0xd9d4d039u, 0xe6db99e5u, 0x1fa27cf8u, 0xc4ac5665u,
            0xf4292244u, 0x432aff97u, 0xab9423a7u, 0xfc93a039u, 0x655b59c3u, 0x8f0ccc92u, 0xffeff47du, 0x85845dd1u,
            0x6fa87e4fu, 0xfe2ce6e0u, 0xa3014314u, 0x4e0811a1u, 0xf7537e82u, 0xbd3af235u, 0x2ad7d2bbu, 0xeb86d391u};
        static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                                  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                                  4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
        uint32_t m[16];
        for (int i = 0; i < 16; ++i)
            m[i] = static_cast<uint32_t>(p[4 * i]) | static_cast<uint32_t>(p[4 * i + 1]) << 8 |
                   static_cast<uint32_t>(p[4 * i + 2]) << 16 | static_cast<uint32_t>(p[4 * i + 3]) << 24;
        uint32_t a = a_, b = b_, c = c_, d = d_;
        for (int i = 0; i < 64; ++i) {
            uint32_t f;
            int g;
            if (i < 16) { f = (b & c) | (~b & d); g = i; }
            else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
            else { f = c ^ (b | ~d); g = (7 * i) & 15; }
            const uint32_t t = d;
            d = c;
            c = b;
            b = b + rol(a + f + K[i] + m[g], S[i]);
            a = t;
        }
        a_ += a; b_ += b; c_ += c; d_ += d;
    }
    uint32_t a_, b_, c_, d_;
    uint64_t len_;
    size_t fill_;
    uint8_t buf_[64];
};

// ---- CRCs of the frame layer: CRC-8 (x^8 + x^2 + x + 1) and CRC-16 (x^16 + x^15 + x^2 + 1), both initialised with 0
inline uint8_t crc8(const uint8_t* p, size_t n) {
    // (a function-local static initialised by a lambda: thread-safe, the decode team calls this concurrently)
    static const std::array<uint8_t, 256> table = [] {
        std::array<uint8_t, 256> t{};
        for (int i = 0; i < 256; ++i) {
            uint8_t c = static_cast<uint8_t>(i);
            for (int k = 0; k < 8; ++k) c = static_cast<uint8_t>((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1));
            t[i] = c;
        }
        return t;
    }();
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) c = table[c ^ p[i]];
    return c;
}

inline uint16_t crc16(const uint8_t* p, size_t n) {
    static const std::array<uint16_t, 256> table = [] {
        std::array<uint16_t, 256> t{};
        for (int i = 0; i < 256; ++i) {
            uint16_t c = static_cast<uint16_t>(i << 8);
            for (int k = 0; k < 8; ++k) c = static_cast<uint16_t>((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1));
            t[i] = c;
        }
        return t;
    }();
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) c = static_cast<uint16_t>((c << 8) ^ table[(c >> 8) ^ p[i]]);
    return c;
}

// ---- MSB-first bit reader over a byte range
class BitReader {
public:
    BitReader(const uint8_t* p, size_t n, size_t at) : p_(p), n_(n), pos_(at) {}
    bool failed() const { return failed_; }
    // byte offset of the next unread bit (exact when the reader is byte-aligned)
    size_t byte_position() const { return pos_ - static_cast<size_t>(bits_ >> 3); }
    bool aligned() const { return (bits_ & 7) == 0; }
    void align() {
        const int drop = bits_ & 7;
        acc_ <<= drop;
        bits_ -= drop;
    }
    uint32_t read(int k) {   // 0 <= k <= 32
        if (k == 0) return 0;
        refill();
        if (bits_ < k) { failed_ = true; return 0; }
        const uint32_t v = static_cast<uint32_t>(acc_ >> (64 - k));
        acc_ <<= k;
        bits_ -= k;
        return v;
    }
    int64_t read_signed(int k) {   // 1 <= k <= 33
        if (k <= 32) {
            const uint32_t v = read(k);
            const uint32_t sign = 1u << (k - 1);
            return static_cast<int64_t>(v ^ sign) - static_cast<int64_t>(sign);
        }
        const uint64_t hi = read(k - 32);
        const uint64_t lo = read(32);
        const uint64_t v = (hi << 32) | lo;
        const uint64_t sign = 1ull << (k - 1);
        return static_cast<int64_t>(v ^ sign) - static_cast<int64_t>(sign);
    }
    uint32_t read_unary() {   // number of 0 bits before the next 1 bit (which is consumed)
        uint32_t count = 0;
        for (;;) {
            refill();
            if (bits_ == 0) { failed_ = true; return 0; }
            if (acc_ == 0) {          // every buffered bit is 0
                count += static_cast<uint32_t>(bits_);
                bits_ = 0;
                if (count > (1u << 28)) { failed_ = true; return 0; }
                continue;
            }
            const int z = __builtin_clzll(acc_);   // < bits_: the bits below the buffered ones are 0
            count += static_cast<uint32_t>(z);
            acc_ <<= z;
            acc_ <<= 1;
            bits_ -= z + 1;
            return count;
        }
    }

private:
    void refill() {
        while (bits_ <= 56 && pos_ < n_) {
            acc_ |= static_cast<uint64_t>(p_[pos_++]) << (56 - bits_);
            bits_ += 8;
        }
    }
    const uint8_t* p_;
    size_t n_, pos_;
    uint64_t acc_ = 0;
    int bits_ = 0;
    bool failed_ = false;
};

inline bool parse_stream_info(const uint8_t* p, size_t n, StreamInfo* info, std::string* err) {
    size_t at = 0;
    if (n >= 10 && std::memcmp(p, "ID3", 3) == 0) {   // ID3v2 tag in front of the stream: 4 sync-safe size bytes
        const size_t size = (static_cast<size_t>(p[6] & 0x7f) << 21) | (static_cast<size_t>(p[7] & 0x7f) << 14) |
                            (static_cast<size_t>(p[8] & 0x7f) << 7) | static_cast<size_t>(p[9] & 0x7f);
        at = 10 + size + ((p[5] & 0x10) ? 10 : 0);
    }
    if (at + 4 > n || std::memcmp(p + at, "fLaC", 4) != 0) { *err = "not a FLAC stream (no fLaC marker)"; return false; }
    at += 4;
    bool seen = false;
    for (;;) {
        if (at + 4 > n) { *err = "FLAC metadata runs past the end of the file"; return false; }
        const bool last = (p[at] & 0x80) != 0;
        const int type = p[at] & 0x7f;
        const size_t len = (static_cast<size_t>(p[at + 1]) << 16) | (static_cast<size_t>(p[at + 2]) << 8) | p[at + 3];
        at += 4;
        if (at + len > n) { *err = "FLAC metadata block runs past the end of the file"; return false; }
        if (type == 0) {
            if (len < 34) { *err = "FLAC STREAMINFO block too short"; return false; }
            const uint8_t* s = p + at;
            info->min_block = (s[0] << 8) | s[1];
            info->max_block = (s[2] << 8) | s[3];
            info->sample_rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
            info->channels = ((s[12] >> 1) & 7) + 1;
            info->bits = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
            info->total_frames = (static_cast<int64_t>(s[13] & 0x0f) << 32) | (static_cast<int64_t>(s[14]) << 24) |
                                 (static_cast<int64_t>(s[15]) << 16) | (static_cast<int64_t>(s[16]) << 8) | s[17];
            std::memcpy(info->md5, s + 18, 16);
            info->has_md5 = false;
            for (int i = 0; i < 16; ++i) info->has_md5 |= info->md5[i] != 0;
            seen = true;
        } else if (type == 127) {
            *err = "invalid FLAC metadata block type";
            return false;
        }
        at += len;
        if (last) break;
    }
    if (!seen) { *err = "FLAC stream without STREAMINFO"; return false; }
    if (info->sample_rate <= 0) { *err = "FLAC STREAMINFO has sample rate 0"; return false; }
    if (info->bits < 4 || info->bits > 32) { *err = "FLAC bits per sample outside 4..32"; return false; }
    info->audio_at = at;
    return true;
}

namespace detail {

inline bool decode_residual(BitReader& br, int block, int order, int64_t* out, std::string* err) {
    const int method = static_cast<int>(br.read(2));
    if (method > 1) { *err = "reserved FLAC residual coding method"; return false; }
    const int param_bits = method == 0 ? 4 : 5;
    const uint32_t escape = method == 0 ? 15u : 31u;
    const int porder = static_cast<int>(br.read(4));
    const int parts = 1 << porder;
    if ((block & (parts - 1)) != 0 || (block >> porder) < order) {
        *err = "FLAC residual partitioning does not fit the block";
        return false;
    }
    int at = order;
    for (int part = 0; part < parts; ++part) {
        const int count = (block >> porder) - (part == 0 ? order : 0);
        const uint32_t param = br.read(param_bits);
        if (param == escape) {
            const int raw = static_cast<int>(br.read(5));
            for (int i = 0; i < count; ++i) out[at++] = raw ? br.read_signed(raw) : 0;
        } else {
            for (int i = 0; i < count; ++i) {
                const uint32_t q = br.read_unary();
                const uint64_t u = (static_cast<uint64_t>(q) << param) | br.read(static_cast<int>(param));
                out[at++] = static_cast<int64_t>(u >> 1) ^ -static_cast<int64_t>(u & 1);
            }
        }
        if (br.failed()) { *err = "FLAC frame ends inside its residual"; return false; }
    }
    return true;
}

inline bool decode_subframe(BitReader& br, int block, int bps, int64_t* out, std::string* err) {
    if (br.read(1) != 0) { *err = "FLAC subframe padding bit is set"; return false; }
    const int type = static_cast<int>(br.read(6));
    int wasted = 0;
    if (br.read(1)) wasted = static_cast<int>(br.read_unary()) + 1;
    if (br.failed()) { *err = "FLAC frame ends inside a subframe header"; return false; }
    if (wasted >= bps) { *err = "FLAC subframe wastes every bit"; return false; }
    bps -= wasted;
    if (type == 0) {                       // constant
        const int64_t v = br.read_signed(bps);
        for (int i = 0; i < block; ++i) out[i] = v;
    } else if (type == 1) {                // verbatim
        for (int i = 0; i < block; ++i) out[i] = br.read_signed(bps);
    } else if (type >= 8 && type <= 12) {  // fixed predictor, order 0..4
        const int order = type - 8;
        if (order > block) { *err = "FLAC fixed predictor order exceeds the block"; return false; }
        for (int i = 0; i < order; ++i) out[i] = br.read_signed(bps);
        if (!decode_residual(br, block, order, out, err)) return false;
        // (unsigned arithmetic: a damaged stream may drive these sums past 64 bits, which must wrap, not be undefined;
        // such a frame fails its CRC afterwards)
        uint64_t* u = reinterpret_cast<uint64_t*>(out);
        switch (order) {
            case 1: for (int i = 1; i < block; ++i) u[i] += u[i - 1]; break;
            case 2: for (int i = 2; i < block; ++i) u[i] += 2 * u[i - 1] - u[i - 2]; break;
            case 3: for (int i = 3; i < block; ++i) u[i] += 3 * u[i - 1] - 3 * u[i - 2] + u[i - 3]; break;
            case 4: for (int i = 4; i < block; ++i) u[i] += 4 * u[i - 1] - 6 * u[i - 2] + 4 * u[i - 3] - u[i - 4]; break;
            default: break;
        }
    } else if (type >= 32) {               // LPC, order 1..32
        const int order = type - 31;
        if (order > block) { *err = "FLAC LPC order exceeds the block"; return false; }
        for (int i = 0; i < order; ++i) out[i] = br.read_signed(bps);
        const int precision = static_cast<int>(br.read(4)) + 1;
        if (precision == 16) { *err = "invalid FLAC LPC coefficient precision"; return false; }
        const int shift = static_cast<int>(br.read_signed(5));
        if (shift < 0) { *err = "negative FLAC LPC shift"; return false; }
        int64_t coef[32];
        for (int j = 0; j < order; ++j) coef[j] = br.read_signed(precision);
        if (!decode_residual(br, block, order, out, err)) return false;
        for (int i = order; i < block; ++i) {
            uint64_t sum = 0;
            for (int j = 0; j < order; ++j) sum += static_cast<uint64_t>(coef[j]) * static_cast<uint64_t>(out[i - 1 - j]);
            out[i] = static_cast<int64_t>(static_cast<uint64_t>(out[i]) + static_cast<uint64_t>(static_cast<int64_t>(sum) >> shift));
        }
    } else {
        *err = "reserved FLAC subframe type";
        return false;
    }
    if (br.failed()) { *err = "FLAC frame ends inside a subframe"; return false; }
    if (wasted)
        for (int i = 0; i < block; ++i) out[i] = static_cast<int64_t>(static_cast<uint64_t>(out[i]) << wasted);
    return true;
}

}  // namespace detail

// Decode the whole stream.  `out` = NULL counts frames only.  Interleaved int32 samples, as many as `capacity_frames`
// allows (more is an error).  Returns false with *err set; *frames_out is valid on success.
inline bool decode(const uint8_t* p, size_t n, const StreamInfo& info, int32_t* out, int64_t capacity_frames,
                   bool verify_md5, int64_t* frames_out, std::string* err) {
    static const int kBlockSizes[16] = {0, 192, 576, 1152, 2304, 4608, -8, -16, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
    static const int kRates[12] = {0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};
    static const int kBits[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    const int channels = info.channels;
    std::vector<int64_t> work(static_cast<size_t>(channels) * 65536);
    std::vector<uint8_t> pcm;
    Md5 md5;
    const bool hash = verify_md5 && info.has_md5;
    const int sample_bytes = (info.bits + 7) / 8;
    int64_t done = 0;
    size_t at = info.audio_at;
    while (at < n) {
        if (info.total_frames > 0 && done >= info.total_frames) break;   // trailing bytes (tags) are not audio
        if (at + 2 > n || p[at] != 0xFF || (p[at + 1] & 0xFE) != 0xF8) {
            // bytes that are no frame: tolerated only at the very end of a stream of unknown length (ID3v1, padding)
            if (info.total_frames == 0 && done > 0) break;
            *err = "FLAC frame sync lost at byte " + std::to_string(at);
            return false;
        }
        BitReader br(p, n, at + 2);
        const int bs_code = static_cast<int>(br.read(4));
        const int sr_code = static_cast<int>(br.read(4));
        const int ch_code = static_cast<int>(br.read(4));
        const int bits_code = static_cast<int>(br.read(3));
        if (br.read(1) != 0) { *err = "reserved bit set in a FLAC frame header"; return false; }
        // frame or sample number, coded like UTF-8 (up to 7 bytes): not needed for a sequential decode
        {
            const uint32_t first = br.read(8);
            int extra = 0;
            if (first >= 0xFE) extra = 6;
            else if (first >= 0xFC) extra = 5;
            else if (first >= 0xF8) extra = 4;
            else if (first >= 0xF0) extra = 3;
            else if (first >= 0xE0) extra = 2;
            else if (first >= 0xC0) extra = 1;
            else if (first >= 0x80) { *err = "invalid coded number in a FLAC frame header"; return false; }
            for (int i = 0; i < extra; ++i)
                if ((br.read(8) & 0xC0) != 0x80) { *err = "invalid coded number in a FLAC frame header"; return false; }
        }
        int block = kBlockSizes[bs_code];
        if (block == 0) { *err = "reserved FLAC block size code"; return false; }
        if (block == -8) block = static_cast<int>(br.read(8)) + 1;
        else if (block == -16) block = static_cast<int>(br.read(16)) + 1;
        if (sr_code == 12) br.read(8);
        else if (sr_code == 13 || sr_code == 14) br.read(16);
        else if (sr_code == 15) { *err = "invalid FLAC sample rate code"; return false; }
        else if (sr_code != 0 && kRates[sr_code] != info.sample_rate) { *err = "FLAC sample rate changes mid-stream"; return false; }
        if (br.failed() || !br.aligned()) { *err = "FLAC frame header truncated"; return false; }
        const size_t header_end = br.byte_position();
        const uint32_t want8 = br.read(8);
        if (br.failed() || crc8(p + at, header_end - at) != want8) { *err = "FLAC frame header CRC mismatch at byte " + std::to_string(at); return false; }
        int bps = kBits[bits_code];
        if (bps < 0) { *err = "reserved FLAC sample size code"; return false; }
        if (bps == 0) bps = info.bits;
        if (bps != info.bits) { *err = "FLAC bits per sample change mid-stream"; return false; }
        int n_ch;
        if (ch_code < 8) n_ch = ch_code + 1;
        else if (ch_code <= 10) n_ch = 2;
        else { *err = "reserved FLAC channel assignment"; return false; }
        if (n_ch != channels) { *err = "FLAC channel count changes mid-stream"; return false; }
        if (block > 65535 + 1 || block < 1) { *err = "invalid FLAC block size"; return false; }
        for (int c = 0; c < n_ch; ++c) {
            const bool side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
            if (!detail::decode_subframe(br, block, bps + (side ? 1 : 0), work.data() + static_cast<size_t>(c) * 65536, err)) return false;
        }
        br.align();
        const size_t body_end = br.byte_position();
        const uint32_t want16 = br.read(16);
        if (br.failed() || crc16(p + at, body_end - at) != want16) { *err = "FLAC frame CRC mismatch at byte " + std::to_string(at); return false; }
        at = body_end + 2;
        int64_t* c0 = work.data();
        int64_t* c1 = work.data() + 65536;
        if (ch_code == 8) {            // left, side
            for (int i = 0; i < block; ++i) c1[i] = static_cast<int64_t>(static_cast<uint64_t>(c0[i]) - static_cast<uint64_t>(c1[i]));
        } else if (ch_code == 9) {     // side, right
            for (int i = 0; i < block; ++i) c0[i] = static_cast<int64_t>(static_cast<uint64_t>(c0[i]) + static_cast<uint64_t>(c1[i]));
        } else if (ch_code == 10) {    // mid, side
            for (int i = 0; i < block; ++i) {
                const int64_t side = c1[i];
                const int64_t mid = static_cast<int64_t>((static_cast<uint64_t>(c0[i]) << 1) | static_cast<uint64_t>(side & 1));
                c0[i] = static_cast<int64_t>(static_cast<uint64_t>(mid) + static_cast<uint64_t>(side)) >> 1;
                c1[i] = static_cast<int64_t>(static_cast<uint64_t>(mid) - static_cast<uint64_t>(side)) >> 1;
            }
        }
        int take = block;
        if (info.total_frames > 0 && done + take > info.total_frames) { *err = "FLAC stream holds more samples than STREAMINFO says"; return false; }
        if (out != nullptr) {
            if (done + take > capacity_frames) { *err = "FLAC stream is longer than the output buffer"; return false; }
            int32_t* dst = out + done * channels;
            for (int i = 0; i < take; ++i)
                for (int c = 0; c < channels; ++c) dst[static_cast<size_t>(i) * channels + c] = static_cast<int32_t>(work[static_cast<size_t>(c) * 65536 + i]);
        }
        if (hash) {
            pcm.resize(static_cast<size_t>(take) * channels * sample_bytes);
            size_t k = 0;
            for (int i = 0; i < take; ++i)
                for (int c = 0; c < channels; ++c) {
                    const uint32_t v = static_cast<uint32_t>(static_cast<int32_t>(work[static_cast<size_t>(c) * 65536 + i]));
                    for (int b = 0; b < sample_bytes; ++b) pcm[k++] = static_cast<uint8_t>(v >> (8 * b));
                }
            md5.update(pcm.data(), pcm.size());
        }
        done += take;
    }
    if (info.total_frames > 0 && done != info.total_frames) { *err = "FLAC stream ends before the sample count of STREAMINFO"; return false; }
    if (hash) {
        uint8_t got[16];
        md5.finish(got);
        if (std::memcmp(got, info.md5, 16) != 0) { *err = "FLAC MD5 signature of the decoded audio does not match STREAMINFO"; return false; }
    }
    *frames_out = done;
    return true;
}

}  // namespace flac
}  // namespace serhip
