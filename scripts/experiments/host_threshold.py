"""At which buffer size does the staged host path (ingest.features_from_host) beat the plain pageable copy?
encode_sequence on host recordings of 30 s .. 10 min with FeatureEngine.STREAM_FROM_HOST_BYTES = never / always.  [gpu]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

from ser_amd import FeatureFlags, HandcraftedBackend   # noqa: E402
from ser_amd.engine import FeatureEngine, get_engine   # noqa: E402
from ser_amd.synth import harmonic_clips_torch   # noqa: E402

SR = 16000
eng = get_engine()
backend = HandcraftedBackend(feature_flags=FeatureFlags(tonnetz=False))
for seconds in (30, 60, 120, 300, 600, 1200):
    audio = harmonic_clips_torch(7, 1, seconds * SR, eng.device).reshape(-1).cpu().numpy()
    line = [f"{seconds:5d} s ({audio.nbytes / 2**20:6.1f} MiB)"]
    for name, threshold in (("plain", 1 << 40), ("staged", 1)):
        FeatureEngine.STREAM_FROM_HOST_BYTES = threshold
        backend.encode_sequence(audio, SR)
        best = 1e9
        for _ in range(6):
            t0 = time.perf_counter()
            backend.encode_sequence(audio, SR)
            best = min(best, time.perf_counter() - t0)
        line.append(f"{name} {best * 1e3:6.2f} ms")
    print("  ".join(line), flush=True)
