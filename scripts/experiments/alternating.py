"""k_stft_frames on a batch whose clip lengths alternate (3 s, 1.5 s, 3 s, ...): a period that divides the grid."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from ser_amd import FeatureFlags
from ser_amd.engine import get_engine
from ser_amd.synth import harmonic_clips_torch
eng = get_engine()
n = 8000
audio = harmonic_clips_torch(3, n, 48000, eng.device).reshape(-1)
offs = np.arange(n, dtype=np.int64) * 48000
lens = np.where(np.arange(n) % 2 == 0, 48000, 24000)
fl = FeatureFlags(tonnetz=False)
for _ in range(2): eng.features(audio, offs, lens, 16000, fl)
torch.cuda.synchronize()
eng._ctx.profile_enable(True)
for _ in range(5): out = eng.features(audio, offs, lens, 16000, fl)
torch.cuda.synchronize()
eng._ctx.profile_enable(False)
k1, _ = eng._ctx.profile_read(0); k2, _ = eng._ctx.profile_read(1); k3, _ = eng._ctx.profile_read(3)
print(f"alternating 3 s / 1.5 s x {n}: K1 {k1/5:.3f}  Kfin {k2/5:.3f}  Kproj {k3/5:.3f} ms", flush=True)
