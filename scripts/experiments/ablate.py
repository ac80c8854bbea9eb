import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from ser_amd import FeatureFlags
from ser_amd.engine import get_engine
from ser_amd.synth import harmonic_clips_torch
eng = get_engine()
n = 5000
audio = harmonic_clips_torch(1, n, 48000, eng.device).reshape(-1)
offs = np.arange(n, dtype=np.int64) * 48000; lens = np.full(n, 48000)
for name, fl in (("all", FeatureFlags(tonnetz=False)), ("mfcc+mel", FeatureFlags(True, False, True, False, False)),
                 ("chroma", FeatureFlags(False, True, False, False, False)), ("contrast-only", FeatureFlags(False, False, False, True, False))):
    for _ in range(2): eng.features(audio, offs, lens, 16000, fl)
    torch.cuda.synchronize()
    eng._ctx.profile_enable(True)
    for _ in range(5): eng.features(audio, offs, lens, 16000, fl)
    torch.cuda.synchronize()
    eng._ctx.profile_enable(False)
    k1, n1 = eng._ctx.profile_read(0); k2, n2 = eng._ctx.profile_read(1); k3, n3 = eng._ctx.profile_read(3)
    print(f"{name:14s} K1 {k1/5:.3f} ms/step ({n1} launches)  Kfin {k2/5:.3f} ms/step  Kproj {k3/5:.3f} ms/step  total {(k1+k2+k3)/5:.3f}")
