"""End-to-end latency of the file-level inference entry points on a long and a short recording:
extract_feature_frames(file) [+ temporal_pooling_windows + mean_std_pool].  Usage: python scripts/experiments/file_inference_latency.py"""
import os, sys, time, tempfile, shutil, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts"))
from pathlib import Path
import numpy as np, torch
from bench_configs import harmonic_clips_torch, SR, FLAGS
from ser_amd import extract_feature_frames, mean_std_pool, temporal_pooling_windows, HandcraftedBackend
from ser_amd.feature_extractor import FeatureSettings
from ser_amd.engine import get_engine
from ser_amd.synth import write_wav_pcm16
eng = get_engine()
root = Path(tempfile.mkdtemp(prefix="serhip_lat_", dir="/dev/shm"))
try:
    settings = FeatureSettings(feature_flags=FLAGS)
    for seconds in (3600, 60, 4):
        audio = harmonic_clips_torch(4321, 1, seconds * SR, eng.device).reshape(-1).cpu().numpy()
        path = root / f"rec{seconds}.wav"
        write_wav_pcm16(path, np.rint(audio * 32767).astype(np.int16), SR)
        extract_feature_frames(str(path), settings=settings)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); frames = extract_feature_frames(str(path), settings=settings); best = min(best, time.perf_counter() - t0)
        print(f"{seconds} s file: extract_feature_frames {best*1e3:.1f} ms for {len(frames)} frames", flush=True)
        if seconds == 3600:
            pr = cProfile.Profile(); pr.enable(); extract_feature_frames(str(path), settings=settings); pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
finally:
    shutil.rmtree(root, ignore_errors=True)
