"""Differential stress of the two tuning kernels: random clip batches (harmonics at random levels of noise, silence,
near-silence, random lengths from 2 048 samples) through ONE large call (k_tuning_waves + fallback) and through calls of
512 clips (k_segment_finish<1> alone); tuning estimates and feature bits must be identical.
Usage: python scripts/experiments/tuning_kernels_stress.py [rounds]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from ser_amd import FeatureFlags
from ser_amd.engine import get_engine
from ser_amd.synth import harmonic_clips_torch
eng = get_engine(); fl = FeatureFlags(tonnetz=False)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, length = 4096, 48000
bad = 0
hist = {}
for r in range(rounds):
    rng = np.random.default_rng(1000 + r)
    gen = torch.Generator(device=eng.device).manual_seed(2000 + r)
    audio = harmonic_clips_torch(3000 + r, n, length, eng.device)
    level = torch.from_numpy(rng.choice([0.0, 0.0, 1e-3, 1e-2, 0.05, 0.3, 1.0], n).astype(np.float32)).to(eng.device)
    audio += level[:, None] * torch.randn((n, length), generator=gen, device=eng.device)
    audio[torch.from_numpy(rng.choice(n, 40, replace=False)).to(eng.device)] = 0.0
    audio[torch.from_numpy(rng.choice(n, 40, replace=False)).to(eng.device)] *= 1e-20
    audio = audio.reshape(-1).contiguous()
    lens = rng.integers(2048, length + 1, n)
    offs = np.arange(n, dtype=np.int64) * length
    big, big_t = eng.features(audio, offs, lens, 16000, fl, return_tuning=True)
    parts = [eng.features(audio, offs[lo:lo + 512], lens[lo:lo + 512], 16000, fl, return_tuning=True) for lo in range(0, n, 512)]
    small, small_t = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
    same = bool(torch.equal(big_t, small_t)) and bool(torch.equal(big, small))
    bad += not same
    vals, counts = torch.unique(big_t, return_counts=True)
    print(f"round {r}: identical={same}  distinct tunings {len(vals)}  finite={bool(torch.isfinite(big).all())}", flush=True)
print("FAILED rounds:", bad)
sys.exit(1 if bad else 0)
