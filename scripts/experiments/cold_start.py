"""Cold-start account of a one-shot process (the reference's CLI runs one inference per process): what each step
costs the first time.  Usage: python scripts/experiments/cold_start.py"""
import os, sys, time
t00 = time.perf_counter()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
t0 = time.perf_counter(); import torch; t_torch = time.perf_counter() - t0
t0 = time.perf_counter(); import ser_amd; from ser_amd import FeatureFlags, HandcraftedBackend; t_pkg = time.perf_counter() - t0
t0 = time.perf_counter(); torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize(); t_hip = time.perf_counter() - t0
from ser_amd.engine import get_engine
t0 = time.perf_counter(); eng = get_engine(); t_eng = time.perf_counter() - t0
flags = FeatureFlags(tonnetz=False)
t0 = time.perf_counter(); plan = eng._ctx.plan(16000, flags.native_bits()); t_plan = time.perf_counter() - t0
from ser_amd.synth import sample_wav_like
audio = sample_wav_like(seed=11, n_samples=69937)
be = HandcraftedBackend(feature_flags=flags)
t0 = time.perf_counter(); enc = be.encode_sequence(audio, 16000); t_first = time.perf_counter() - t0
t0 = time.perf_counter(); enc = be.encode_sequence(audio, 16000); t_second = time.perf_counter() - t0
t0 = time.perf_counter(); plan48 = eng._ctx.plan(48000, flags.native_bits()); t_plan48 = time.perf_counter() - t0
print(f"import torch {t_torch*1e3:.0f} ms | import ser_amd {t_pkg*1e3:.0f} | HIP init {t_hip*1e3:.0f} | engine (ctx) {t_eng*1e3:.0f} | "
      f"plan 16 kHz {t_plan*1e3:.0f} | first encode_sequence {t_first*1e3:.1f} | second {t_second*1e3:.2f} | plan 48 kHz {t_plan48*1e3:.0f} | "
      f"process so far {1e3*(time.perf_counter()-t00):.0f} ms")
