"""End-to-end host latency of the inference-shaped calls on one short recording (config 1):
HandcraftedBackend.encode_sequence and extract_vector on a host numpy buffer (H2D, kernels, D2H, Python)."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from ser_amd import FeatureFlags, HandcraftedBackend
from ser_amd.synth import sample_wav_like
from ser_amd.engine import get_engine
eng = get_engine()
audio = sample_wav_like()
b = HandcraftedBackend(feature_flags=FeatureFlags(tonnetz=False))
def t(f, n=200):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("encode_sequence (5 windows)   %.0f us" % t(lambda: b.encode_sequence(audio, 16000)))
print("extract_vector (whole file)   %.0f us" % t(lambda: b.extract_vector(audio, 16000)))
dev = eng.to_device(audio)
offs = [0, 16000, 32000, 48000, 64000]; lens = [48000, 48000, 37937, 21937, 5937]
fl = FeatureFlags(tonnetz=False)
print("engine.features on device buf %.0f us" % t(lambda: eng.features(dev, offs, lens, 16000, fl)))
print("  + .cpu()                    %.0f us" % t(lambda: eng.features(dev, offs, lens, 16000, fl).cpu()))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): b.encode_sequence(audio, 16000)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
