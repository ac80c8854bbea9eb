"""Host-side cost of one serhip_fast_features call whose plan cannot be replayed (ragged segment lengths that change
from call to call): the time until the call returns (planning + descriptor upload + four launches) against the time
the kernels take.  Usage: python scripts/experiments/host_plan_cost.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from ser_amd import FeatureFlags
from ser_amd.engine import get_engine
from ser_amd.synth import harmonic_clips_torch
eng = get_engine(); fl = FeatureFlags(tonnetz=False)
n = 12500
audio = harmonic_clips_torch(1234, n, 48000, eng.device).reshape(-1)
rng = np.random.default_rng(0)
for name, make in (("uniform 3 s", lambda: np.full(n, 48000)), ("ragged 1.5-3 s", lambda: rng.integers(24000, 48001, n)), ("ragged 0.2-3 s", lambda: rng.integers(3200, 48001, n))):
    offs = np.arange(n, dtype=np.int64) * 48000
    for _ in range(3): eng.features(audio, offs, make(), 16000, fl)
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(10):
        lens = make()
        t0 = time.perf_counter(); out = eng.features(audio, offs, lens, 16000, fl); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append(t1 - t0); total.append(t2 - t0)
    print(f"{name}: call returns after {1e3*np.median(host):.2f} ms, results ready after {1e3*np.median(total):.2f} ms", flush=True)
