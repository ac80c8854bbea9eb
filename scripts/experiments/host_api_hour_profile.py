"""Where the 14 ms of config 4 through the host API go (encode_sequence on a host numpy hour + windows + pooling):
cProfile of the third call, top entries by cumulative time.  [gpu]"""
import cProfile
import io
import pstats
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

from ser_amd import FeatureFlags, HandcraftedBackend   # noqa: E402
from ser_amd.engine import get_engine   # noqa: E402
from ser_amd.pool import mean_std_pool, temporal_pooling_windows   # noqa: E402
from ser_amd.synth import harmonic_clips_torch   # noqa: E402

SR = 16000
eng = get_engine()
hour = harmonic_clips_torch(4321, 3600, SR, eng.device).reshape(-1)
hour = hour / hour.abs().max()
host_hour = hour.cpu().numpy()
del hour
backend = HandcraftedBackend(feature_flags=FeatureFlags(tonnetz=False))


def once():
    t0 = time.perf_counter()
    enc = backend.encode_sequence(host_hour, SR)
    t1 = time.perf_counter()
    wins = temporal_pooling_windows(enc, window_size_seconds=10.0, window_stride_seconds=5.0)
    t2 = time.perf_counter()
    pooled = mean_std_pool(enc, wins)
    t3 = time.perf_counter()
    return [(b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2, t3))], pooled


for _ in range(3):
    print("encode / windows / pool ms:", [round(x, 2) for x in once()[0]])
prof = cProfile.Profile()
prof.enable()
once()
prof.disable()
out = io.StringIO()
pstats.Stats(prof, stream=out).sort_stats("cumulative").print_stats(35)
print(out.getvalue())
