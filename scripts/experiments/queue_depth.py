"""Per-call time of small asynchronous calls as a function of how many are enqueued without a synchronisation."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from ser_amd import FeatureFlags
from ser_amd.synth import sample_wav_like
from ser_amd.engine import get_engine
eng = get_engine()
dev = eng.to_device(sample_wav_like())
offs = np.asarray([0, 16000, 32000, 48000, 64000]); lens = np.asarray([48000, 48000, 37937, 21937, 5937])
fl = FeatureFlags(tonnetz=False)
f = lambda: eng.features(dev, offs, lens, 16000, fl)
for _ in range(20): f()
torch.cuda.synchronize()
for reps in (10, 50, 100, 200, 400, 800):
    t0 = time.perf_counter()
    for _ in range(reps): f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"reps {reps:4d}: enqueue {1e6*(t1-t0)/reps:6.1f} us/call, total {1e6*(t2-t0)/reps:6.1f} us/call", flush=True)
