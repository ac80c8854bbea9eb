"""Per-run clocks of the pipelined loader on the config-2-from-files corpus (11 520 x reps/60 files): where does a slow
run spend its time?  Usage: python scripts/experiments/ingest_runs.py [reps] [runs]
EXP_PROFILE=1: a cProfile of one more call (main thread only) behind the runs, top entries by own time."""
import os, sys, time, tempfile, shutil, gc
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts"))
from pathlib import Path
import numpy as np, torch
from bench_configs import ravdess_sine_clip, SR, FLAGS
from ser_amd import AudioReadConfig, ingest
from ser_amd.data_loader import featurize_audio_jobs_device
from ser_amd.engine import get_engine
from ser_amd.synth import write_wav_pcm16

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 240
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
eng = get_engine()
root = Path(tempfile.mkdtemp(prefix="serhip_runs_", dir="/dev/shm"))
try:
    files = []
    for actor in range(1, 25):
        d = root / f"Actor_{actor:02d}"; d.mkdir()
        for emo in range(1, 9):
            pcm = np.rint(ravdess_sine_clip(actor, emo) * 32768).astype(np.int16)
            for rep in range(reps):
                p = d / f"03-01-{emo:02d}-01-01-{rep:03d}-{actor:02d}.wav"; write_wav_pcm16(p, pcm, SR); files.append(str(p))
    jobs = [(f, None, None) for f in files]
    cfg = AudioReadConfig(); threads = ingest.host_threads()
    for r in range(runs):
        if os.environ.get("EXP_GC_OFF"): gc.disable()
        stats = {}
        t0 = time.perf_counter()
        m, errs = featurize_audio_jobs_device(jobs, feature_flags=FLAGS, audio_read_config=cfg, workers=threads, pipelined=True, stats=stats)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host = eng.to_host_float64(m)
        t3 = time.perf_counter()
        print(f"run {r}: call {1e3*(t1-t0):.1f} sync {1e3*(t2-t1):.1f} to_host {1e3*(t3-t2):.1f} | " +
              " ".join(f"{k[:-2]} {1e3*v:.1f}" for k, v in stats.items() if isinstance(v, float)), flush=True)
    if os.environ.get("EXP_PROFILE"):
        import cProfile, io, pstats
        prof = cProfile.Profile()
        prof.enable()
        featurize_audio_jobs_device(jobs, feature_flags=FLAGS, audio_read_config=cfg, workers=threads, pipelined=True, to_host=True)
        prof.disable()
        out = io.StringIO()
        pstats.Stats(prof, stream=out).sort_stats("tottime").print_stats(40)
        print(out.getvalue())
    if os.environ.get("EXP_LOAD_DATA"):   # the reference-shaped entry point on the same corpus, profiled
        import cProfile, io, pstats
        from ser_amd.data_loader import LoaderSettings, load_data
        settings = LoaderSettings(dataset_folder=root, feature_flags=FLAGS, max_workers=threads, num_cores=threads)
        for _ in range(2):
            t0 = time.perf_counter(); load_data(settings=settings); print(f"load_data {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
        prof = cProfile.Profile()
        prof.enable()
        load_data(settings=settings)
        prof.disable()
        out = io.StringIO()
        pstats.Stats(prof, stream=out).sort_stats("cumulative").print_stats(45)
        print(out.getvalue())
finally:
    shutil.rmtree(root, ignore_errors=True)
