// Mutation fuzzer for the native WAVE reader (ser_amd/csrc/wav_reader.hpp: load_wave + select_clip), host code only:
//   g++ -std=c++17 -O1 -g -pthread -fsanitize=address,undefined -fno-sanitize-recover=all -I. scripts/fuzz_wav.cpp -o /tmp/fuzz_wav
//   /tmp/fuzz_wav /dev/shm/fuzz_wav_scratch.wav a.wav b.wav ...     (seed files: PCM16 / float32 / extensible WAVE)
// Every mutant (bit flips, random bytes, truncations, random runs, 0x00/0xFF runs, mostly aimed at the first 96 bytes
// where the chunk headers sit) is written to the scratch path and read into a heap slab of exactly the region's size:
// the reader must take it or refuse it without a sanitizer report, and a clip it reports must lie inside the region
// (every byte of it is read here, so an out-of-bounds descriptor is a report too).  Round 3: 400 000 mutants of four
// seeds (PCM16 mono / stereo, float32 stereo behind a 6-byte LIST chunk, WAVE_FORMAT_EXTENSIBLE), 0 reports.
#include "ser_amd/csrc/wav_reader.hpp"
#include <cstdio>
#include <fstream>
#include <iterator>
using namespace serhip;
static uint64_t rng = 88172645463325252ull;
static uint64_t next() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const char* scratch = argv[1];
    long taken = 0, refused = 0;
    unsigned long sink = 0;
    for (int a = 2; a < argc; ++a) {
        std::ifstream f(argv[a], std::ios::binary);
        std::vector<uint8_t> base((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        for (int it = 0; it < 100000; ++it) {
            std::vector<uint8_t> d = base;
            const int kind = next() % 5;
            const int n_mut = 1 + next() % 4;
            for (int m = 0; m < n_mut; ++m) {
                const size_t span = (next() % 4) ? std::min<size_t>(96, d.size()) : d.size();
                const size_t at = next() % span;
                if (kind == 0) d[at] ^= 1u << (next() % 8);
                else if (kind == 1) d[at] = static_cast<uint8_t>(next());
                else if (kind == 2) { d.resize(at + 1); break; }
                else if (kind == 3) { for (size_t k = at; k < d.size() && k < at + 16; ++k) d[k] = static_cast<uint8_t>(next()); }
                else { for (size_t k = at; k < d.size() && k < at + 8; ++k) d[k] = (next() & 1) ? 0xFF : 0x00; }
            }
            { std::ofstream o(scratch, std::ios::binary | std::ios::trunc); o.write(reinterpret_cast<const char*>(d.data()), static_cast<std::streamsize>(d.size())); }
            const int64_t room = (static_cast<int64_t>(d.size()) + 15) / 16 * 16;
            const int64_t region = 16 * static_cast<int64_t>(next() % 3);             // the file's region starts a little into the slab
            std::vector<unsigned char> slab(static_cast<size_t>(region + room));      // exact: one byte past it is a report
            WaveInSlab wave;
            const int status = load_wave(scratch, slab.data(), region, room, &wave, 1 + static_cast<int>(next() % 2));
            if (status != SERHIP_WAV_OK) { ++refused; continue; }
            for (int sel = 0; sel < 4; ++sel) {
                const double start = sel == 0 ? NAN : static_cast<double>(next() % 4000) / 1000.0 * (sel == 3 ? 1e15 : 1.0);
                const double dur = (sel & 1) ? NAN : static_cast<double>(next() % 3000) / 1000.0;
                serhip_wav_clip clip{};
                if (select_clip(wave, start, dur, &clip) != SERHIP_WAV_OK) continue;
                const int64_t bytes = clip.frames * clip.channels * (clip.sample_format == SERHIP_SAMPLE_S16 ? 2 : 4);
                if (clip.raw_offset < region || clip.raw_offset + bytes > region + room) { std::printf("clip outside its region\n"); return 1; }
                for (int64_t b = 0; b < bytes; ++b) sink += slab[static_cast<size_t>(clip.raw_offset + b)];
            }
            ++taken;
        }
    }
    std::printf("taken %ld refused %ld (%lu)\n", taken, refused, sink & 1);
    return 0;
}
