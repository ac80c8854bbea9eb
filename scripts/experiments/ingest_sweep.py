#!/usr/bin/env python3
"""Pipelined file ingest with different reader-thread counts and slab sizes (one dataset, one process)."""
import os, shutil, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from ser_amd import AudioReadConfig, FeatureFlags, ingest
from ser_amd.data_loader import featurize_audio_jobs_device
from ser_amd.synth import ravdess_sine_clip, write_wav_pcm16
FLAGS = FeatureFlags(tonnetz=False)
root = Path(tempfile.mkdtemp(prefix="serhip_sweep_", dir="/dev/shm"))
try:
    files = []
    for actor in range(1, 25):
        d = root / f"Actor_{actor:02d}"; d.mkdir()
        for emo in range(1, 9):
            pcm = np.rint(ravdess_sine_clip(actor, emo) * 32768).astype(np.int16)
            for rep in range(60):
                p = d / f"03-01-{emo:02d}-01-01-{rep:02d}-{actor:02d}.wav"; write_wav_pcm16(p, pcm, 16000); files.append(str(p))
    jobs = [(f, None, None) for f in files]
    cfg = AudioReadConfig()
    def run(workers):
        st = {}
        t0 = time.perf_counter()
        m, e = featurize_audio_jobs_device(jobs, feature_flags=FLAGS, audio_read_config=cfg, workers=workers, pipelined=True, stats=st)
        t1 = time.perf_counter()
        h = m.to(torch.float64).cpu().numpy()
        t2 = time.perf_counter()
        return t2 - t0, t1 - t0, st
    run(16)
    for si in (5e-3, 1e-4, 2e-5, 5e-6):
      os.environ['SERHIP_SWITCH_INTERVAL'] = str(si); print('switch interval', si)
      for slab in (16 << 20,):
        os.environ["SERHIP_SLAB_BYTES"] = str(slab)
        for workers in (12, 15, 16):
            best = min((run(workers) for _ in range(4)), key=lambda r: r[0])
            st = best[2]
            print(f"slab {slab>>20:3d} MiB workers {workers:2d}: total {best[0]*1e3:6.2f} ms (to device matrix {best[1]*1e3:6.2f}) slabs {st['slabs']:3d} "
                  f"native {st['reader_native_s']*1e3:6.2f} wait_pin {st['reader_wait_pinned_s']*1e3:5.2f} main_wait {st['main_wait_reader_s']*1e3:6.2f} main_issue {st['main_issue_s']*1e3:6.2f} init {st['init_s']*1e3:5.2f} inside {st['total_s']*1e3:6.2f}", flush=True)
    # reader-only at the same thread counts
    sizes = ingest.stat_files(files, 16); room = (sizes + 15) // 16 * 16
    pin = torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True)
    for workers in (8, 12, 15, 16, 24):
        slabs = ingest.plan_slabs(sizes, 32 << 20)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            blob, at = ingest.encode_paths(files)
            ingest.stat_files(None, workers, table=(blob, at))
            for lo, hi in slabs:
                offs = np.concatenate(([0], np.cumsum(room[lo:hi])[:-1]))
                ingest.read_wav_batch(None, pin.data_ptr(), pin.numel(), offs, room[lo:hi], threads=workers, table=(blob, at[lo:hi]))
            best = min(best, time.perf_counter() - t0)
        print(f"reader only workers {workers}: {best*1e3:.2f} ms")
finally:
    shutil.rmtree(root, ignore_errors=True)
