#!/usr/bin/env python3
"""Parity of the HIP path against the CPU oracle on a large sample of config-3 clips.

The parity tests use a few dozen clips per case so that the oracle finishes in seconds; this script
runs the same comparison on thousands (the oracle on every host core first, in forked workers, before
this process touches the GPU) and reports what the tests can only sample: how many tuning estimates
agree exactly, the worst normalised error per feature block (<= 1 passes the BASELINE.md gate), and
what the disagreeing clips look like.  The oracle is used here as the checker only.

    python scripts/parity_at_scale.py --clips 2000 [--kind harmonic|noisy|windows] [--out report.json]

``--kind windows`` is config 4's shape: ONE recording of ``--clips`` seconds, featurised as overlapping
3 s windows at a 1 s stride (the last two windows are shorter), every window against the oracle.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

SR = 16000
N = 48000


def make_clip(seed: int, kind: str):
    import numpy as np

    from ser_amd.synth import harmonic_clip, peak_normalise

    rng = np.random.default_rng(seed)
    clip = harmonic_clip(rng, N, SR)
    if kind == "noisy":     # harmonics buried in noise: many peaks near the median threshold
        clip = peak_normalise((clip + 0.2 * rng.standard_normal(N)).astype(np.float32))
    return clip


def make_recording(seconds: int, seed: int):
    import numpy as np

    from ser_amd.synth import harmonic_clip, peak_normalise

    rng = np.random.default_rng(seed)
    return peak_normalise(np.concatenate([harmonic_clip(rng, SR, SR) for _ in range(seconds)]))


_RECORDING = None   # set before the pool forks: the workers inherit it


def _oracle_row(args):
    seed, kind = args
    from oracle import ser_fast_path as ofp

    if kind == "windows":
        clip = _RECORDING[seed * SR : seed * SR + N]
    else:
        clip = make_clip(seed, kind)
    v, dbg = ofp.extract_feature_from_signal(clip, SR, flags=ofp.Flags(), return_debug=True)
    return v, dbg.get("tuning", 0.0)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=2000)
    ap.add_argument("--kind", choices=("harmonic", "noisy", "windows"), default="harmonic")
    ap.add_argument("--workers", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--seed0", type=int, default=500000)
    ap.add_argument("--out", type=str, default="")
    args = ap.parse_args()

    import multiprocessing as mp

    import numpy as np

    os.environ.setdefault("OMP_NUM_THREADS", "1")
    global _RECORDING
    if args.kind == "windows":
        _RECORDING = make_recording(args.clips, args.seed0)
        seeds = list(range(args.clips))                      # window w starts at second w
    else:
        seeds = list(range(args.seed0, args.seed0 + args.clips))
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(args.workers) as pool:
        rows = pool.map(_oracle_row, [(s, args.kind) for s in seeds], chunksize=max(1, args.clips // (8 * args.workers)))
    t_cpu = time.perf_counter() - t0
    want = np.asarray([r[0] for r in rows], dtype=np.float64)
    want_tun = np.asarray([r[1] for r in rows], dtype=np.float64)

    from parity_utils import CHROMA, block_errors

    from ser_amd import FeatureFlags
    from ser_amd.engine import get_engine

    eng = get_engine()
    if args.kind == "windows":
        audio = _RECORDING
        offs = [w * SR for w in range(args.clips)]
        lens = [min(N, audio.size - o) for o in offs]
    else:
        audio = np.concatenate([make_clip(s, args.kind) for s in seeds])
        offs, lens = [i * N for i in range(args.clips)], [N] * args.clips
    out, tun = eng.features(audio, offs, lens, SR, FeatureFlags(tonnetz=False), return_tuning=True)
    got = out.cpu().numpy().astype(np.float64)
    got_tun = tun.cpu().numpy().astype(np.float64)

    same = np.abs(got_tun - want_tun) < 1e-6
    report = {
        "clips": args.clips,
        "kind": args.kind,
        "oracle_seconds": round(t_cpu, 1),
        "oracle_workers": args.workers,
        "tuning_equal": int(same.sum()),
        "tuning_differs": int((~same).sum()),
        "tuning_max_abs_diff": float(np.max(np.abs(got_tun - want_tun))),
        "worst_normalised_error_where_tuning_equal": block_errors(got[same], want[same]) if same.any() else None,
    }
    if (~same).any():
        # a different tuning selects another chroma filterbank; every other block must still pass
        e = block_errors(got[~same], want[~same])
        report["where_tuning_differs"] = {
            "seeds": [seeds[i] for i in np.nonzero(~same)[0][:20]],
            "gpu_vs_oracle_tuning": [[float(got_tun[i]), float(want_tun[i])] for i in np.nonzero(~same)[0][:20]],
            "worst_normalised_error_mfcc_mel": {k: e[k] for k in ("mfcc", "mel", "contrast")},
            "chroma_max_abs_diff": float(np.max(np.abs(got[~same][:, CHROMA] - want[~same][:, CHROMA]))),
        }
    text = json.dumps(report, indent=1)
    print(text, flush=True)
    if args.out:
        Path(args.out).write_text(text + "\n")


if __name__ == "__main__":
    main()
