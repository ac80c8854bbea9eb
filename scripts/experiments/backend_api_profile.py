"""Where the time of HandcraftedBackend.extract_vectors on 12 500 separate arrays goes (cProfile of one call) and
repeated timings of encode_sequence on a host hour.  [gpu]"""
import cProfile
import io
import pstats
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

from ser_amd import FeatureFlags, HandcraftedBackend   # noqa: E402
from ser_amd.engine import get_engine   # noqa: E402
from ser_amd.synth import harmonic_clips_torch   # noqa: E402

SR = 16000
eng = get_engine()
n, length = 12500, 48000
audio = harmonic_clips_torch(1234, n, length, eng.device).reshape(-1).cpu().numpy()
clips = [audio[i * length:(i + 1) * length] for i in range(n)]
backend = HandcraftedBackend(feature_flags=FeatureFlags(tonnetz=False))
for _ in range(4):
    t0 = time.perf_counter()
    backend.extract_vectors(clips, SR)
    print("extract_vectors ms", round((time.perf_counter() - t0) * 1e3, 2))
prof = cProfile.Profile()
prof.enable()
backend.extract_vectors(clips, SR)
prof.disable()
out = io.StringIO()
pstats.Stats(prof, stream=out).sort_stats("cumulative").print_stats(30)
print(out.getvalue())
hour = audio[: 3600 * SR]
for _ in range(8):
    t0 = time.perf_counter()
    backend.encode_sequence(hour, SR)
    print("encode_sequence(hour) ms", round((time.perf_counter() - t0) * 1e3, 2))
