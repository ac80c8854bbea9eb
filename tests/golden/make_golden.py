#!/usr/bin/env python3
"""Generate tests/golden/fast_path_golden.npz from the CPU oracle (run in the build container).

The reference itself cannot produce vectors here (needs Python >= 3.12 and librosa),
and its tests hold none for the librosa arithmetic, so these vectors pin the ORACLE's current
behaviour (a regression guard) and give the GPU tests fixed expected values; they are not outputs
of the reference.  Inputs are regenerated from seeds by ser_amd.synth.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import ser_fast_path as ofp  # noqa: E402
from oracle.whisper_logmel import log_mel_spectrogram  # noqa: E402
from ser_amd.synth import harmonic_clips, ravdess_sine_clip, sample_wav_like  # noqa: E402

SR = 16000


def rows(audio, offs, lens):
    out, tun = [], []
    for o, n in zip(offs, lens):
        v, d = ofp.extract_feature_from_signal(audio[o : o + n], SR, return_debug=True)
        out.append(v)
        tun.append(d["tuning"])
    return np.asarray(out), np.asarray(tun)


def main():
    g = {}
    clips = harmonic_clips(1234, 6, 48000)
    g["clips_features"], g["clips_tuning"] = rows(clips.reshape(-1), [i * 48000 for i in range(6)], [48000] * 6)
    a = sample_wav_like()
    emb, starts, ends = ofp.encode_sequence(a, SR)
    g["sample_embeddings"], g["sample_starts"], g["sample_ends"] = emb, starts, ends
    rav = [ofp.prepare_audio_buffer(ravdess_sine_clip(ac, e)) for ac in (1, 2) for e in range(1, 9)]
    g["ravdess_features"] = np.asarray([ofp.extract_vector(c, SR) for c in rav])
    rng = np.random.default_rng(5)
    x = rng.standard_normal(40000).astype(np.float32)
    x /= np.abs(x).max()
    offs = np.asarray([0, 1, 4001, 100, 7, 0, 10, 33, 500, 77, 3, 9])
    lens = np.asarray([2048, 2049, 5937, 20000, 29993, 100, 512, 1023, 1500, 2047, 1, 513])
    g["ragged_offsets"], g["ragged_lengths"] = offs, lens
    g["ragged_features"], g["ragged_tuning"] = rows(x, offs, lens)
    wclips = harmonic_clips(99, 2, 16000 * 2)
    g["whisper_logmel80"] = np.stack([log_mel_spectrogram(c, 80) for c in wclips])
    np.savez_compressed(Path(__file__).with_name("fast_path_golden.npz"), **g)
    print({k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
