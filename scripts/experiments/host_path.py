import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from ser_amd import FeatureFlags, HandcraftedBackend
from ser_amd.engine import get_engine
from ser_amd import dsp
eng = get_engine()
x = np.random.default_rng(0).standard_normal(57_600_000).astype(np.float32); x /= np.abs(x).max()
def t(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("validate_signal      %.1f ms" % t(lambda: dsp.validate_signal(x, 16000)))
print("to_device (pageable) %.1f ms" % t(lambda: eng.to_device(x)))
b = HandcraftedBackend(feature_flags=FeatureFlags(tonnetz=False))
print("encode_sequence      %.1f ms" % t(lambda: b.encode_sequence(x, 16000)))
clips = [x[i * 48000:(i + 1) * 48000] for i in range(1200)]
print("extract_vectors 1200 clips (230 MB)  %.1f ms" % t(lambda: b.extract_vectors(clips, 16000)))
xp = torch.from_numpy(x).pin_memory()
print("to_device (pinned)   %.1f ms" % t(lambda: xp.to(eng.device, non_blocking=True)))
