import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from ser_amd import FeatureFlags
from ser_amd.engine import get_engine
from ser_amd.synth import harmonic_clips_torch
eng = get_engine()
n = 12500
audio = harmonic_clips_torch(1, n, 48000, eng.device).reshape(-1)
offs = np.arange(n, dtype=np.int64) * 48000; lens = np.full(n, 48000)
fl = FeatureFlags(tonnetz=False)
for _ in range(2): eng.features(audio, offs, lens, 16000, fl)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): eng.features(audio, offs, lens, 16000, fl)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 5 * 1e3
eng._ctx.profile_enable(True)
for _ in range(5): eng.features(audio, offs, lens, 16000, fl)
torch.cuda.synchronize()
eng._ctx.profile_enable(False)
k1, n1 = eng._ctx.profile_read(0); k2, n2 = eng._ctx.profile_read(1); k3, n3 = eng._ctx.profile_read(3)
print(f"pipe={os.environ.get('SERHIP_PIPELINE','default')} chunk={os.environ.get('SERHIP_CHUNK_FRAMES','default'):>8s} wall {wall:.3f} ms/step | K1 {k1/5:.3f} ({n1//5} launches) Kfin {k2/5:.3f} Kproj {k3/5:.3f} sum {(k1+k2+k3)/5:.3f}", flush=True)
