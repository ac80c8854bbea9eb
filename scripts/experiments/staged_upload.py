import sys, time, numpy as np
sys.path.insert(0, '.')
from ser_amd.engine import FeatureEngine, get_engine
import torch
eng = get_engine()
x = np.random.default_rng(0).standard_normal((46080, 187))
for name, thr in (("plain", 1 << 40), ("staged", 8 << 20)):
    FeatureEngine.STAGED_TO_HOST_BYTES = thr
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter(); d = eng.to_device_staged(x); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(name, round(best * 1e3, 2), "ms")
