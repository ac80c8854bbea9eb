"""Kernel times on inputs with many spectral peaks (white noise, speech-like mixtures): the tuning
kernel's peaks no longer fit its LDS staging."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from ser_amd import FeatureFlags
from ser_amd.engine import get_engine
from ser_amd.synth import harmonic_clips_torch
eng = get_engine()
n = 5000
g = torch.Generator(device=eng.device); g.manual_seed(7)
harm = harmonic_clips_torch(1, n, 48000, eng.device)
noise = torch.randn((n, 48000), generator=g, device=eng.device)
offs = np.arange(n, dtype=np.int64) * 48000; lens = np.full(n, 48000)
fl = FeatureFlags(tonnetz=False)
for name, x in (("harmonic (bench)", harm), ("harmonic + 0.3 noise", harm + 0.3 * noise), ("white noise", noise)):
    a = (x / x.abs().amax(dim=1, keepdim=True)).reshape(-1).contiguous()
    for _ in range(2): eng.features(a, offs, lens, 16000, fl)
    torch.cuda.synchronize()
    eng._ctx.profile_enable(True)
    for _ in range(3): out = eng.features(a, offs, lens, 16000, fl)
    torch.cuda.synchronize()
    eng._ctx.profile_enable(False)
    k1, _ = eng._ctx.profile_read(0); k2, _ = eng._ctx.profile_read(1); k3, _ = eng._ctx.profile_read(3)
    print(f"{name:22s} K1 {k1/3:.3f}  Kfin {k2/3:.3f}  Kproj {k3/3:.3f}  total {(k1+k2+k3)/3:.3f} ms per {n} clips", flush=True)
