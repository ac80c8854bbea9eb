#!/usr/bin/env python3
"""The two pipelined ingest paths and nothing else, for a rocprofv3 kernel + memory-copy trace:

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/ingest_trace -- \
        python3 scripts/ingest_trace.py
    python3 scripts/overlap_report.py gpurun_out/ingest_trace > profiles/<tag>_ingest_overlap.json

Part 1: 11 520 PCM16 WAV files on tmpfs through featurize_audio_jobs_device(pipelined=True), three times.
Part 2: 12 500 float32 clips in pageable memory through engine.features (staged chunks), three times.
"""
import os
import shutil
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ser_amd import AudioReadConfig, FeatureFlags  # noqa: E402
from ser_amd.data_loader import featurize_audio_jobs_device  # noqa: E402
from ser_amd.engine import get_engine  # noqa: E402
from ser_amd.synth import harmonic_clips_torch, ravdess_sine_clip, write_wav_pcm16  # noqa: E402

FLAGS = FeatureFlags(tonnetz=False)


def main():
    eng = get_engine()
    root = Path(tempfile.mkdtemp(prefix="serhip_trace_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None))
    try:
        files = []
        for actor in range(1, 25):
            d = root / f"Actor_{actor:02d}"
            d.mkdir()
            for emo in range(1, 9):
                pcm = np.rint(ravdess_sine_clip(actor, emo) * 32768).astype(np.int16)
                for rep in range(60):
                    p = d / f"03-01-{emo:02d}-01-01-{rep:02d}-{actor:02d}.wav"
                    write_wav_pcm16(p, pcm, 16000)
                    files.append(str(p))
        jobs = [(f, None, None) for f in files]
        for i in range(4):
            t0 = time.perf_counter()
            m, errs = featurize_audio_jobs_device(jobs, feature_flags=FLAGS, audio_read_config=AudioReadConfig(), workers=16, pipelined=True)
            torch.cuda.synchronize()
            print(f"files pass {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms", file=sys.stderr)
            assert not any(errs)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    n, length = 12500, 48000
    audio = harmonic_clips_torch(1234, n, length, eng.device).reshape(-1).cpu().numpy()
    offs = np.arange(n, dtype=np.int64) * length
    lens = np.full(n, length, dtype=np.int64)
    for i in range(4):
        t0 = time.perf_counter()
        out = eng.features(audio, offs, lens, 16000, FLAGS)
        torch.cuda.synchronize()
        print(f"host pass {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms", file=sys.stderr)
    assert bool(torch.isfinite(out).all())


if __name__ == "__main__":
    main()
