#!/usr/bin/env python3
"""Throughput of the BASELINE.json parity configurations 1, 2, 4 and 5 on one MI355X
(config 3 is bench.py's headline line).  Inputs are synthetic and resident in HBM before the
timed region; each figure is audio-seconds of input per second of wall time, with the
fraction of the HBM roofline on SURVEY.md 8d's algorithmic bytes.  Writes one JSON document.

Round 3 adds the END-TO-END ingest lines, where inputs start on the host and the figure is what a real caller
gets: ``config2_from_files`` (>= 10 k PCM16 WAV files on tmpfs -> feature matrix on the host -> prepared-feature
cache) and ``config3_from_host`` (12 500 float32 clips in pageable numpy memory -> features), each with the
bounds measured in the same run (reader threads alone, pageable->pinned staging alone, pinned H2D alone, kernels
alone) and the pipelined figure as a fraction of the tightest one.

    python scripts/bench_configs.py > gpurun_out/configs.json
    python scripts/bench_configs.py --only config2_from_files,config3_from_host
    python scripts/bench_configs.py --only generic_short      # 70 000 x 700-sample segments (k_dft_frames path)
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ser_amd import FeatureFlags, HandcraftedBackend, mean_std_pool, temporal_pooling_windows  # noqa: E402
from ser_amd.engine import get_engine  # noqa: E402
from ser_amd.handcrafted import sliding_window_bounds  # noqa: E402
from ser_amd.synth import harmonic_clips_torch, ravdess_sine_clip, sample_wav_like  # noqa: E402

SR = 16000
HBM = 8.0e12
FLAGS = FeatureFlags(tonnetz=False)


INVOCATIONS = {"n": 0}   # calls of the timed closures since the last reset (profiles divide kernel totals by it)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    INVOCATIONS["n"] += reps + 1
    return (time.perf_counter() - t0) / reps, out


def h2d_rate(eng, nbytes=256 << 20, reps=5):
    """GB/s of an asynchronous copy from pinned host memory to the GPU (the PCIe bound of both ingest paths)."""
    pin = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    dev = torch.empty(nbytes, dtype=torch.uint8, device=eng.device)
    dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t0) / 1e9


def config1(eng, res):
    audio = eng.to_device(sample_wav_like())
    starts, ends = sliding_window_bounds(int(audio.numel()), SR, 3, 1)
    offs, lens = starts, ends - starts
    assert len(offs) == 5
    dt, out = timed(lambda: eng.features(audio, offs, lens, SR, FLAGS), 50)
    res["config1"] = {"what": "69 937-sample recording, 5 windows (3 s / 1 s), features on device", "ms": dt * 1e3,
                      "algorithmic_bytes": 4 * int(audio.numel()) + 5 * 4 * 187,
                      "audio_seconds_per_s": (audio.numel() / SR) / dt}


def config2_scaled(eng, res):
    base = np.stack([ravdess_sine_clip(a, e) for a in range(1, 25) for e in range(1, 9)])
    base = base / np.abs(base).max(axis=1, keepdims=True)
    clips = eng.to_device(np.tile(base, (60, 1)).reshape(-1))
    n = 24 * 8 * 60
    offs2 = np.arange(n, dtype=np.int64) * 24000
    lens2 = np.full(n, 24000, dtype=np.int64)
    dt, out = timed(lambda: eng.features(clips, offs2, lens2, SR, FLAGS), 5)
    sec = n * 1.5
    res["config2_scaled"] = {"what": f"{n} clips x 1.5 s (47 frames each)", "ms": dt * 1e3, "audio_seconds_per_s": sec / dt,
                             "algorithmic_bytes": n * (4 * 24000 + 4 * 187),
                             "hbm_roofline_frac": (sec / dt) * (4 * 24000 + 4 * 187) / 1.5 / HBM}


def config4(eng, res):
    hour = harmonic_clips_torch(4321, 3600, SR, eng.device).reshape(-1)
    hour = hour / hour.abs().max()
    starts, ends = sliding_window_bounds(int(hour.numel()), SR, 3, 1)
    offs4, lens4 = starts, ends - starts
    assert len(offs4) == 3600
    dt, out = timed(lambda: eng.features(hour, offs4, lens4, SR, FLAGS), 10)
    res["config4"] = {"what": "1 h recording, 3 600 overlapping windows by descriptor (no copies)", "ms": dt * 1e3,
                      "audio_seconds_per_s": 3600.0 / dt, "algorithmic_bytes": 3600 * (4 * SR + 4 * 187),
                      "hbm_roofline_frac": (3600.0 / dt) * (4 * SR + 4 * 187) / HBM}


def config4_host_api(eng, res):
    hour = harmonic_clips_torch(4321, 3600, SR, eng.device).reshape(-1)
    hour = hour / hour.abs().max()
    backend = HandcraftedBackend(feature_flags=FLAGS)
    host_hour = hour.cpu().numpy()
    del hour
    times = []
    for _ in range(8):   # the first pass pays one-time costs (pooling workspace, staging buffers, page faults of the host buffer)
        t0 = time.perf_counter()
        enc = backend.encode_sequence(host_hour, SR)
        wins = temporal_pooling_windows(enc, window_size_seconds=10.0, window_stride_seconds=5.0)
        pooled = mean_std_pool(enc, wins)
        times.append((time.perf_counter() - t0) * 1e3)
    res["config4_host_api"] = {"what": "same through HandcraftedBackend.encode_sequence (230 MB H2D, staged in chunks) + "
                                       "temporal_pooling_windows + mean_std_pool", "ms": min(times[1:]),
                               "median_ms": float(np.median(times[1:])), "first_call_ms": times[0], "pooled_shape": list(pooled.shape)}


def config5(eng, res):
    b = 64
    clips5 = harmonic_clips_torch(99, b, 30 * SR, eng.device)
    dt, out = timed(lambda: eng.whisper_logmel(clips5, n_mels=80), 10)
    sec = b * 30.0
    res["config5"] = {"what": f"{b} x 30 s clips -> [{b}, 80, 3000] log-mel", "ms": dt * 1e3,
                      "audio_seconds_per_s": sec / dt, "algorithmic_bytes": int(sec * 96000),
                      "hbm_roofline_frac": (sec / dt) * 96000 / HBM}


def generic_short(eng, res):
    """The generic path (segments < 2048 samples: k_dft_frames + k_segment_finish<0>): the 70 000 x 700-sample
    batch of tests/test_gpu_parity.py::test_more_segments_than_one_launch_holds, so that its cost per
    audio-second is known (a batch of sub-0.128 s clips runs entirely here)."""
    n, length = 70000, 700
    audio = harmonic_clips_torch(7, 700, 70000, eng.device).reshape(-1)
    offs = np.arange(n, dtype=np.int64) * length
    lens = np.full(n, length, dtype=np.int64)
    dt, out = timed(lambda: eng.features(audio, offs, lens, SR, FLAGS), 3)
    sec = n * length / SR
    res["generic_short"] = {"what": f"{n} segments x {length} samples (n_fft = {length}: direct f64 DFT path)", "ms": dt * 1e3,
                            "audio_seconds_per_s": sec / dt, "algorithmic_bytes": n * (4 * length + 4 * 187),
                            "hbm_roofline_frac": (sec / dt) * (4 * length + 4 * 187) / (length / SR) / HBM}


def classifier(eng, res):
    """StandardScaler + MLP(187 -> 300 -> 8) forward pass on a device-resident feature matrix (k_mlp_*): the
    reference's model.predict / predict_proba (fast_path.py:180-192) for one GPU's shard of config 3."""
    from ser_amd.classifier import GpuEmotionClassifier

    rng = np.random.default_rng(0)
    n = 12500
    clf = GpuEmotionClassifier(classes=[f"e{i}" for i in range(8)], coefs=[rng.standard_normal((187, 300)) * 0.1,
                                                                           rng.standard_normal((300, 8)) * 0.1],
                               intercepts=[rng.standard_normal(300), rng.standard_normal(8)], mean=rng.standard_normal(187),
                               scale=rng.uniform(0.5, 2.0, 187))
    x = torch.from_numpy(rng.standard_normal((n, 187)).astype(np.float32)).to(eng.device)
    dt, _ = timed(lambda: clf.predict_device(x), 10)
    flop = 2.0 * n * (187 * 300 + 300 * 8)
    res["classifier"] = {"what": f"{n} x 187 float32 features -> scaler + 300-unit relu MLP + softmax (float64 arithmetic)",
                         "ms": dt * 1e3, "rows_per_s": n / dt, "f64_gflops": flop / dt / 1e9,
                         "algorithmic_bytes": n * (187 * 4 + 8 * 8 + 4)}


def pooling(eng, res):
    """mean_std_pool over one hour of windows (k_pool_windows): 3 600 x 187 embeddings, 719 windows of 10 s / 5 s."""
    emb = torch.randn((3600, 187), device=eng.device)
    lo = np.arange(0, 3591, 5, dtype=np.int32)
    hi = np.minimum(lo + 12, 3600).astype(np.int32)
    dt, out = timed(lambda: eng.pool_windows(emb, lo, hi, mode=0), 50)
    res["pooling"] = {"what": f"[3600, 187] float32 -> {len(lo)} windows x [mean | std] float64", "ms": dt * 1e3,
                      "algorithmic_bytes": 3600 * 187 * 4 + len(lo) * 374 * 8}


def prepare(eng, res):
    """Device-side _prepare_audio_buffer + PCM16 conversion (k_prepare_convert / k_prepare_normalise): 12 500 stereo
    PCM16 clips of 3 s already in HBM -> mono float32, scrubbed, peak-normalised."""
    n, frames = 12500, 48000
    raw = torch.randint(-20000, 20000, (n * frames * 2,), dtype=torch.int16, device=eng.device).view(torch.uint8)
    off = np.arange(n, dtype=np.int64) * frames * 4
    fr = np.full(n, frames, dtype=np.int64)
    ch = np.full(n, 2, dtype=np.int32)
    dt, _ = timed(lambda: eng.prepare_audio(raw, "s16", off, fr, ch), 5)
    nbytes = n * frames * (4 + 4)     # 4 B of PCM16 stereo in, 4 B of float32 out per frame
    res["prepare"] = {"what": f"{n} stereo PCM16 clips x 3 s in HBM -> mono float32 peak-normalised", "ms": dt * 1e3,
                      "audio_seconds_per_s": n * 3.0 / dt, "algorithmic_bytes": nbytes,
                      "hbm_roofline_frac": nbytes / dt / HBM}


def config2_from_files(eng, res, reps=60, key="config2_from_files"):
    """>= 10 k WAVs on tmpfs -> features (pipelined ingest) -> prepared-feature cache; bounds measured beside it.
    ``reps`` = copies of each of the 192 (actor, emotion) clips: 60 -> 11 520 files (0.55 GB), 240 -> 46 080 files."""
    from ser_amd import AudioReadConfig, ingest
    from ser_amd.data_loader import featurize_audio_jobs_device
    from ser_amd.prepared_cache import write_prepared_feature_payload
    from ser_amd.synth import write_wav_pcm16

    root = Path(tempfile.mkdtemp(prefix="serhip_cfg2_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None))
    try:
        t0 = time.perf_counter()
        files, labels = [], []
        for actor in range(1, 25):
            d = root / f"Actor_{actor:02d}"
            d.mkdir()
            for emo in range(1, 9):
                pcm = np.rint(ravdess_sine_clip(actor, emo) * 32768).astype(np.int16)
                for rep in range(reps):
                    p = d / f"03-01-{emo:02d}-01-01-{rep:03d}-{actor:02d}.wav"
                    write_wav_pcm16(p, pcm, SR)
                    files.append(str(p))
                    labels.append(f"e{emo}")
        n = len(files)
        sec = n * 1.5
        gen_s = time.perf_counter() - t0
        jobs = [(f, None, None) for f in files]
        cfg = AudioReadConfig()
        threads = ingest.host_threads()
        out = {"what": f"{n} PCM16 WAV files x 1.5 s on tmpfs ({root.parent}) -> [n, 187] float64 on the host -> "
                       "prepared-training-fast.npz", "files": n, "audio_seconds": sec, "dataset_write_s": gen_s,
               "host_threads": threads}
        # ---- bound 1: the reader alone (stat + native read into pinned slabs, the pipeline's own slab plan; no GPU)
        sizes = ingest.stat_files(files, threads)
        total = int(sizes.sum())
        slab = max(min(ingest.DEFAULT_SLAB_BYTES, max(8 << 20, -(-total // 32))), 1 << 20)
        slabs = ingest.plan_slabs(sizes, slab)
        room = (sizes + 15) // 16 * 16
        # (into the pipeline's own ring of pinned buffers, in rotation: a reader that refilled ONE cache-resident buffer
        # would set a bound no pipeline can meet, since a buffer in flight to the GPU cannot be overwritten)
        largest = max(int(room[lo:hi].sum()) for lo, hi in slabs)
        pins = [torch.empty(largest, dtype=torch.uint8, pin_memory=True) for _ in range(ingest.PINNED_BUFFERS)]
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            blob, path_at = ingest.encode_paths(files)
            ingest.stat_files(None, threads, table=(blob, path_at))
            for k, (lo, hi) in enumerate(slabs):
                pin = pins[k % len(pins)]
                offs = np.concatenate(([0], np.cumsum(room[lo:hi])[:-1]))
                clips = ingest.read_wav_batch(None, pin.data_ptr(), pin.numel(), offs, room[lo:hi], threads=threads,
                                              table=(blob, path_at[lo:hi]))
                assert not clips["status"].any()
            best = min(best, time.perf_counter() - t0)
        out["reader_only"] = {"s": best, "audio_seconds_per_s": sec / best, "GBps": total / best / 1e9, "slabs": len(slabs)}
        del pins, pin
        # ---- bound 2: pinned H2D alone
        gbps = h2d_rate(eng)
        out["h2d_pinned"] = {"GBps": gbps, "audio_seconds_per_s": sec / (total / (gbps * 1e9))}
        # ---- bound 3: kernels alone on device-resident raw bytes (prepare + features), one batch
        raw = np.concatenate([np.fromfile(f, dtype=np.uint8, offset=44) for f in files[:192]])
        raw_dev = torch.from_numpy(np.tile(raw, reps)).to(eng.device)
        r_off = np.arange(n, dtype=np.int64) * 48000
        frames = np.full(n, 24000, dtype=np.int64)
        chans = np.ones(n, dtype=np.int32)

        def kernels():
            audio, o, fr = eng.prepare_audio(raw_dev, "s16", r_off, frames, chans)
            return eng.features(audio, o, fr, SR, FLAGS)
        dt, _ = timed(kernels, 5)
        out["kernels_only"] = {"s": dt, "audio_seconds_per_s": sec / dt}
        del raw_dev
        # ---- end to end: sequential (round 2) and pipelined, files -> host float64 matrix
        def run(pipelined):
            stats = {}
            t0 = time.perf_counter()
            # what featurize_audio_jobs / featurize_files call: float64 on the host, rows sent back slab by slab
            host, errs = featurize_audio_jobs_device(jobs, feature_flags=FLAGS, audio_read_config=cfg, workers=threads,
                                                     pipelined=pipelined, stats=stats, to_host=True)
            dt = time.perf_counter() - t0
            stats["call_s"] = dt
            assert not any(errs) and np.isfinite(host).all()
            return dt, host, stats
        run(True)     # staging buffers, plans, page cache
        seq = min(run(False)[0] for _ in range(2))
        pip_runs = [run(True) for _ in range(5)]
        pip = min(r[0] for r in pip_runs)
        host, stats = min(pip_runs, key=lambda r: r[0])[1:]
        bound = min(out["reader_only"]["audio_seconds_per_s"], out["h2d_pinned"]["audio_seconds_per_s"],
                    out["kernels_only"]["audio_seconds_per_s"])
        out["sequential"] = {"s": seq, "audio_seconds_per_s": sec / seq}
        out["pipelined"] = {"s": pip, "audio_seconds_per_s": sec / pip, "all_runs_s": [r[0] for r in pip_runs], "stats": stats}
        out["bound"] = {"audio_seconds_per_s": bound,
                        "which": min(("reader_only", "h2d_pinned", "kernels_only"), key=lambda k: out[k]["audio_seconds_per_s"])}
        out["pipelined_frac_of_bound"] = (sec / pip) / bound
        # ---- the reference-shaped entry point on top: glob + label parsing + features + train_test_split
        from ser_amd.data_loader import LoaderSettings, load_data

        settings = LoaderSettings(dataset_folder=root, feature_flags=FLAGS, max_workers=threads, num_cores=threads)
        best_ld = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            split = load_data(settings=settings)
            best_ld = min(best_ld, time.perf_counter() - t0)
        assert split is not None and split[0].shape[0] + split[1].shape[0] == n
        out["load_data_glob_to_split"] = {"s": best_ld, "audio_seconds_per_s": sec / best_ld,
                                          "what": "load_data(settings): sorted glob, RAVDESS label parsing, pipelined features, stratified split"}
        # ---- the cache the reference's trainer consumes (np.savez_compressed, float64): reported beside, not inside
        t0 = time.perf_counter()
        cut = n * 3 // 4
        path = write_prepared_feature_payload(root / "tmp" / "prepared-training-fast.npz", x_train=host[:cut], x_dev=host[:0],
                                              x_test=host[cut:], y_train=labels[:cut], y_dev=[], y_test=labels[cut:],
                                              metadata={"cache_namespace": "fast_features", "feature_dim": 187})
        out["cache_write"] = {"s": time.perf_counter() - t0, "bytes": path.stat().st_size}
        # the corpus repeats 192 clips, so its matrix deflates to nothing; a real corpus has n distinct rows of float32
        # values widened to float64 -- the same writer on such a matrix, and numpy's own on a sixteenth of it
        import io

        distinct = np.random.default_rng(0).standard_normal(host.shape).astype(np.float32).astype(np.float64)
        t0 = time.perf_counter()
        path = write_prepared_feature_payload(root / "tmp" / "prepared-training-fast.npz", x_train=distinct[:cut], x_dev=distinct[:0],
                                              x_test=distinct[cut:], y_train=labels[:cut], y_dev=[], y_test=labels[cut:],
                                              metadata={"cache_namespace": "fast_features", "feature_dim": 187})
        dt_ours = time.perf_counter() - t0
        part = distinct[: max(1, n // 16)]
        t0 = time.perf_counter()
        np.savez_compressed(io.BytesIO(), x_train=part)
        dt_numpy = (time.perf_counter() - t0) * (n / part.shape[0])
        out["cache_write_distinct_rows"] = {"s": dt_ours, "bytes": path.stat().st_size, "matrix_bytes": int(distinct.nbytes),
                                            "numpy_savez_compressed_s": dt_numpy,
                                            "numpy_note": "np.savez_compressed timed on a sixteenth of the rows, scaled by 16"}
        res[key] = out
    finally:
        shutil.rmtree(root, ignore_errors=True)


def config2_from_files_x4(eng, res):
    """The same at four times the corpus (46 080 files, 2.2 GB): the pipeline's fill and drain (stat pass, first read,
    last slab's kernels, matrix to the host) are a fixed cost that a 20 ms job shows and a training corpus does not."""
    config2_from_files(eng, res, reps=240, key="config2_from_files_x4")


def config4_from_file(eng, res):
    """Config 4 as a manifest: ONE hour-long PCM16 WAVE on tmpfs and 3 600 utterances cut from it by
    (start_seconds, duration_seconds) -- 3 s at a 1 s stride -- through the file loader: the recording is read once
    and copied once, every utterance is a clip descriptor into it (peak-normalised per utterance, like the
    reference's segment reads)."""
    from ser_amd import AudioReadConfig, ingest
    from ser_amd.data_loader import featurize_audio_jobs_device
    from ser_amd.synth import write_wav_pcm16

    root = Path(tempfile.mkdtemp(prefix="serhip_cfg4_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None))
    try:
        audio = harmonic_clips_torch(4321, 1, 3600 * SR, eng.device).reshape(-1).cpu().numpy()
        path = root / "hour.wav"
        write_wav_pcm16(path, np.rint(audio * 32767).astype(np.int16), SR)
        jobs = [(str(path), float(k), 3.0) for k in range(3600)]
        cfg = AudioReadConfig()
        threads = ingest.host_threads()

        def run():
            stats = {}
            t0 = time.perf_counter()
            host, errs = featurize_audio_jobs_device(jobs, feature_flags=FLAGS, audio_read_config=cfg, workers=threads,
                                                     pipelined=True, stats=stats, to_host=True)
            dt = time.perf_counter() - t0
            assert not any(errs) and np.isfinite(host).all() and host.shape == (3600, 187)
            return dt, stats
        run()
        dt, stats = min((run() for _ in range(5)), key=lambda r: r[0])
        res["config4_from_file"] = {"what": "one 1 h PCM16 WAVE on tmpfs (115 MB) -> 3 600 utterances (3 s at a 1 s stride) by "
                                            "(start_seconds, duration_seconds) -> [3600, 187] float64 on the host",
                                    "s": dt, "unique_audio_seconds_per_s": 3600.0 / dt, "utterance_seconds_per_s": 3600 * 3.0 / dt,
                                    "file_bytes_read": stats.get("file_bytes"), "slabs": stats.get("slabs"), "stats": stats}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def config3_from_host(eng, res):
    """One GPU's shard of config 3 (12 500 x 3 s float32) starting in pageable numpy memory."""
    from ser_amd import ingest

    n, length = 12500, 48000
    audio = harmonic_clips_torch(1234, n, length, eng.device).reshape(-1).cpu().numpy()
    offs = np.arange(n, dtype=np.int64) * length
    lens = np.full(n, length, dtype=np.int64)
    sec = n * 3.0
    threads = ingest.host_threads()
    out = {"what": f"{n} clips x 3 s float32 in pageable host memory ({audio.nbytes / 1e9:.2f} GB) -> [n, 187] on the device",
           "host_threads": threads}
    pin = torch.empty(256 << 20, dtype=torch.uint8, pin_memory=True)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for k in range(4):
            ingest.parallel_memcpy(pin.data_ptr(), audio.ctypes.data + k * pin.numel(), pin.numel(), threads)
        best = min(best, time.perf_counter() - t0)
    out["staging_memcpy"] = {"GBps": 4 * pin.numel() / best / 1e9, "audio_seconds_per_s": sec / (audio.nbytes / (4 * pin.numel() / best))}
    del pin
    gbps = h2d_rate(eng)
    out["h2d_pinned"] = {"GBps": gbps, "audio_seconds_per_s": sec / (audio.nbytes / (gbps * 1e9))}
    dev = torch.from_numpy(audio[: 2500 * length]).to(eng.device)
    dt, _ = timed(lambda: eng.features(dev, offs[:2500], lens[:2500], SR, FLAGS), 5)
    out["kernels_only"] = {"audio_seconds_per_s": 2500 * 3.0 / dt}
    del dev

    def old():
        return eng.features(torch.from_numpy(audio).to(eng.device), offs, lens, SR, FLAGS)
    dt_old, want = timed(old, 2)
    stats = {}
    ingest.features_from_host(eng, audio, offs, lens, SR, FLAGS, stats=stats)
    dt_new, got = timed(lambda: eng.features(audio, offs, lens, SR, FLAGS), 3)
    assert torch.equal(want, got)
    bound = min(out[k]["audio_seconds_per_s"] for k in ("staging_memcpy", "h2d_pinned", "kernels_only"))
    out["whole_buffer_copy_then_compute"] = {"s": dt_old, "audio_seconds_per_s": sec / dt_old}
    out["pipelined"] = {"s": dt_new, "audio_seconds_per_s": sec / dt_new, "stats": stats}
    out["bound"] = {"audio_seconds_per_s": bound,
                    "which": min(("staging_memcpy", "h2d_pinned", "kernels_only"), key=lambda k: out[k]["audio_seconds_per_s"])}
    out["pipelined_frac_of_bound"] = (sec / dt_new) / bound
    res["config3_from_host"] = out


def config3_backend_api(eng, res):
    """The same shard as a caller of the reference-shaped backend holds it: 12 500 separate numpy arrays through
    ``HandcraftedBackend.extract_vectors`` -> [n, 187] float64 on the host.  Sequential = validate every clip,
    concatenate, copy, compute (``SERHIP_INGEST=sequential``); staged = gathered and tested chunk by chunk by the
    staging threads while the chunk before travels and is featurised."""
    n, length = 12500, 48000
    audio = harmonic_clips_torch(1234, n, length, eng.device).reshape(-1).cpu().numpy()
    clips = [audio[i * length:(i + 1) * length] for i in range(n)]
    backend = HandcraftedBackend(feature_flags=FLAGS)
    sec = n * 3.0
    out = {"what": f"{n} numpy arrays x 3 s float32 ({audio.nbytes / 1e9:.2f} GB) -> HandcraftedBackend.extract_vectors -> [n, 187] float64 on the host"}

    def run():
        t0 = time.perf_counter()
        m = backend.extract_vectors(clips, SR)
        return time.perf_counter() - t0, m
    os.environ["SERHIP_INGEST"] = "sequential"
    try:
        run()
        dt_seq, want = min((run() for _ in range(2)), key=lambda r: r[0])
    finally:
        del os.environ["SERHIP_INGEST"]
    run()
    runs = [run() for _ in range(4)]
    dt_new, got = min(runs, key=lambda r: r[0])
    assert np.array_equal(want, got)
    gbps = h2d_rate(eng)
    bound = sec / (audio.nbytes / (gbps * 1e9))
    out["h2d_pinned"] = {"GBps": gbps, "audio_seconds_per_s": bound}
    out["sequential"] = {"s": dt_seq, "audio_seconds_per_s": sec / dt_seq}
    out["staged"] = {"s": dt_new, "audio_seconds_per_s": sec / dt_new, "all_runs_s": [r[0] for r in runs]}
    out["staged_frac_of_bound"] = (sec / dt_new) / bound
    res["config3_backend_api"] = out


CONFIGS = {"config1": config1, "config2_scaled": config2_scaled, "config4": config4, "config4_host_api": config4_host_api,
           "config5": config5,
           "generic_short": generic_short, "classifier": classifier, "pooling": pooling, "prepare": prepare,
           "config2_from_files": config2_from_files, "config2_from_files_x4": config2_from_files_x4, "config4_from_file": config4_from_file,
           "config3_from_host": config3_from_host, "config3_backend_api": config3_backend_api}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="comma-separated subset of: " + ", ".join(CONFIGS))
    ap.add_argument("--out", default="", help="write the JSON document here instead of stdout")
    args = ap.parse_args()
    names = [n for n in args.only.split(",") if n] or list(CONFIGS)
    eng = get_engine()
    res = {}
    for name in names:
        INVOCATIONS["n"] = 0
        CONFIGS[name](eng, res)
        if name in res:
            res[name]["timed_invocations"] = INVOCATIONS["n"]
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        print(f"[bench_configs] {name} done", file=sys.stderr, flush=True)
    if args.out:
        Path(args.out).write_text(json.dumps(res, indent=1))
    else:
        print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
