"""Kernel times of the bench workload (12 500 clips x 3 s) without bench.py's result checks: for timing
ablated builds of the library whose outputs are garbage (SERHIP_LIBRARY=...)."""
import sys
import os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from ser_amd import FeatureFlags
from ser_amd.engine import get_engine
from ser_amd.synth import harmonic_clips_torch
eng = get_engine()
n = 12500
audio = harmonic_clips_torch(1234, n, 48000, eng.device).reshape(-1)
offs = np.arange(n, dtype=np.int64) * 48000; lens = np.full(n, 48000)
fl = FeatureFlags(tonnetz=False) if not os.environ.get("EXP_CHROMA_ONLY") else FeatureFlags(mfcc=False, mel=False, contrast=False, tonnetz=False)
if os.environ.get("EXP_NO_CHROMA"):   # k_stft_frames<true, false>: no peak scan, control flow independent of the data
    fl = FeatureFlags(chroma=False, tonnetz=False)
for _ in range(2): eng.features(audio, offs, lens, 16000, fl)
torch.cuda.synchronize()
eng._ctx.profile_enable(True)
for _ in range(10): eng.features(audio, offs, lens, 16000, fl)
torch.cuda.synchronize()
eng._ctx.profile_enable(False)
k1, _ = eng._ctx.profile_read(0); k2, _ = eng._ctx.profile_read(1); k3, _ = eng._ctx.profile_read(3)
print(f"stft {k1/10:.3f}  fin {k2/10:.3f}  proj {k3/10:.3f} ms", flush=True)
