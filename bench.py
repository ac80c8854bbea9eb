#!/usr/bin/env python3
"""Headline benchmark: audio-seconds featurized per second on N MI355X (BASELINE.json).

Workload (config 3 of BASELINE.json): 100 000 synthetic 3-second clips at 16 kHz, sharded
by utterance, 12 500 clips per GPU (weak scaling: --gpus 8 is exactly config 3), 187-dim
fast-profile features per clip, then one RCCL all-gather of the feature matrix.  A "step"
is one pass of the hot path over the rank's batch (inputs already resident in HBM) plus,
for N > 1, the gather; the gather of step k is left in flight on RCCL's stream while the
kernels of step k + 1 run, and is waited for before the next gather and before the clock stops.

    python bench.py                     # N = 1, 200 steps after 5 warm-up steps
    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  ``roofline`` is for the dominant kernel (k_stft_frames):
algorithmic bytes (4*N + 4*D per segment) over its HIP-event duration.
``cpu_baseline`` is the CPU restatement of the reference (oracle/, numpy+scipy) timed on
this host's cores on a bounded sample of the same workload (N=1, rank 0 only).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

SAMPLE_RATE = 16000
CLIP_SECONDS = 3
CLIP_SAMPLES = SAMPLE_RATE * CLIP_SECONDS
FEATURE_DIM = 187
CLIPS_PER_GPU = 12500
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
ALGO_BYTES_PER_CLIP = 4 * CLIP_SAMPLES + 4 * FEATURE_DIM   # 192 748 B
FRAMES_PER_CLIP = 1 + CLIP_SAMPLES // 512                  # 94 STFT frames (hop 512, centred)
FFT_FLOP_PER_FRAME = 2.5 * 2048 * 11                       # 56 320: 2.5 N log2 N for a real 2048-point FFT
VALU_F32_PEAK_TFLOPS = 157.3                               # packed-f32 vector peak, MI355X_MICROARCH.md


def _cpu_worker(seed: int) -> int:
    import numpy as np

    from oracle import ser_fast_path as ofp
    from ser_amd.synth import harmonic_clip

    clip = harmonic_clip(np.random.default_rng(seed), CLIP_SAMPLES, SAMPLE_RATE)
    ofp.extract_feature_from_signal(clip, SAMPLE_RATE, flags=ofp.Flags())
    return 1


_BLAS_LIMIT = None


def _one_thread_per_process() -> None:
    """One BLAS/OpenMP thread in this process (kept for its lifetime): the box exports OMP_NUM_THREADS=16,
    and eight workers with sixteen threads each on sixteen cores ran at a third of the rate measured here."""
    global _BLAS_LIMIT
    import numpy  # noqa: F401  (the libraries must be loaded before they can be limited)
    import scipy.fft  # noqa: F401
    import scipy.linalg  # noqa: F401
    from threadpoolctl import threadpool_limits

    _BLAS_LIMIT = threadpool_limits(limits=1)


def cpu_baseline(n_clips: int, workers: int) -> dict:
    """Time the oracle (reference-faithful call structure) on `workers` host processes.

    Runs BEFORE this process touches the GPU: the pool forks plain numpy workers.
    """
    import multiprocessing as mp

    ctx = mp.get_context("fork")
    with ctx.Pool(workers, initializer=_one_thread_per_process) as pool:
        pool.map(_cpu_worker, range(workers))            # warm-up: imports, filterbank code paths
        t0 = time.perf_counter()
        chunk = max(1, n_clips // (4 * workers))         # reference chunking, data_loader.py:374-379
        done = sum(pool.imap_unordered(_cpu_worker, range(1000, 1000 + n_clips), chunksize=chunk))
        dt = time.perf_counter() - t0
    # third leg of SURVEY.md 8d: all the host cores this job may use (the box's share is 16 for one GPU;
    # os.cpu_count() there shows the whole machine)
    all_cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    all_value = None
    if all_cores != workers:
        with ctx.Pool(all_cores, initializer=_one_thread_per_process) as pool:
            pool.map(_cpu_worker, range(all_cores))
            t2 = time.perf_counter()
            done_all = sum(pool.imap_unordered(_cpu_worker, range(20000, 20000 + n_clips),
                                               chunksize=max(1, n_clips // (4 * all_cores))))
            all_value = done_all * CLIP_SECONDS / (time.perf_counter() - t2)
    # the same code on ONE core (SURVEY.md 8d asks for both): a twentieth of the sample, this process
    n_single = max(8, n_clips // 20)
    _one_thread_per_process()
    _cpu_worker(999)
    t1 = time.perf_counter()
    for seed in range(5000, 5000 + n_single):
        _cpu_worker(seed)
    dt_single = time.perf_counter() - t1
    return {
        "value": done * CLIP_SECONDS / dt,
        "unit": "audio-seconds/s",
        "cores": workers,
        "single_core_value": n_single * CLIP_SECONDS / dt_single,
        "all_cores": all_cores,
        "all_cores_value": all_value if all_value is not None else done * CLIP_SECONDS / dt,
        "threads_per_process": 1,
        "kind": "port",
        "sample": f"{done} of the config-3 clips (3 s @16 kHz), numpy/scipy restatement of the reference "
                  f"(oracle/), multiprocessing.Pool({workers}) like data_loader.py:374-379, {dt:.1f} s wall",
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 1 000 steps of ~3.3 ms = a timed region of ~3.3 s (long enough for an outside sampler of GPU activity
    # that looks every few seconds to see it, and for the mean to sit within the box-to-box spread rather than within
    # one step's jitter; 5 steps measured a clock ramp, 200 steps were still shorter than the CPU baseline's import)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--clips-per-gpu", type=int, default=CLIPS_PER_GPU)
    ap.add_argument("--cpu-clips", type=int, default=4000, help="clips in the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-workers", type=int, default=8)
    # rehearsal of the N > 1 code path on a one-GPU box: every rank on the same device, gloo for the
    # collectives (features are copied to the host for the gather).  Not a measurement.
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    cpu = None
    if world == 1 and args.cpu_clips > 0:
        workers = max(1, min(args.cpu_workers, os.cpu_count() or 1, 16))
        cpu = cpu_baseline(args.cpu_clips, workers)

    import torch
    import torch.distributed as dist

    from ser_amd import FeatureFlags
    from ser_amd.distributed import all_gather_rows_async
    from ser_amd.engine import get_engine
    from ser_amd.synth import harmonic_clips_torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    nccl_log = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not args.rehearse_on_one_gpu and "NCCL_DEBUG" not in os.environ and "NCCL_DEBUG_FILE" not in os.environ:
            # RCCL's own account of the transports it connected (read back by rank 0 for the JSON line)
            nccl_log = f"/tmp/serhip_rccl_{os.getpid()}.log"
            os.environ["NCCL_DEBUG"] = "INFO"
            os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,GRAPH,P2P,SHM,NET"
            os.environ["NCCL_DEBUG_FILE"] = nccl_log
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    n_local = int(args.clips_per_gpu)
    n_total = n_local * world
    flags = FeatureFlags(tonnetz=False)
    engine = get_engine(local_rank)
    # synthetic shard, generated on the device from a per-rank seed (config 3: seed 1234)
    audio = harmonic_clips_torch(1234 + rank, n_local, CLIP_SAMPLES, device).reshape(-1)
    offsets = [i * CLIP_SAMPLES for i in range(n_local)]
    lengths = [CLIP_SAMPLES] * n_local
    torch.cuda.synchronize()

    # The gather of step k is left in flight (RCCL runs it on its own stream) while the kernels of step
    # k + 1 are enqueued; it is waited for before the next gather is issued and before the clock stops, so
    # every step's matrix is complete inside the timed region.
    in_flight = []   # [(work, full, feats)] of the last step: keeps the buffers alive until waited

    def drain():
        while in_flight:
            work, _full, _feats = in_flight.pop()
            work.wait()

    def step():
        feats = engine.features(audio, offsets, lengths, SAMPLE_RATE, flags)
        if args.rehearse_on_one_gpu and world > 1:
            feats = feats.cpu()   # gloo gathers host tensors; same in-flight bookkeeping as the RCCL path
        drain()
        full, work = all_gather_rows_async(feats, n_total)
        if work is not None:
            in_flight.append((work, full, feats))
        return full

    def fence():
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    fence()
    engine._ctx.profile_enable(True)
    # one event per step on the stream the kernels run on (torch's current stream): per-step device times
    # without a host synchronisation inside the timed region
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        out = step()
        marks[k + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    engine._ctx.profile_enable(False)
    step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    k1_ms, k1_n = engine._ctx.profile_read(0)
    k2_ms, k2_n = engine._ctx.profile_read(1)
    k3_ms, k3_n = engine._ctx.profile_read(3)
    assert tuple(out.shape) == (n_total, FEATURE_DIM)
    assert bool(torch.isfinite(out).all())
    if world > 1:
        # every rank holds every rank's rows: the `world` shards come from different seeds, so their column
        # sums differ pairwise; this rank's own rows sit at its own place
        sums = out.reshape(world, n_local, FEATURE_DIM).double().sum(dim=1).cpu()
        for a in range(world):
            for b in range(a + 1, world):
                assert not torch.equal(sums[a], sums[b]), f"rank {rank}: shards {a} and {b} are identical"
        mine = engine.features(audio, offsets, lengths, SAMPLE_RATE, flags)
        assert torch.equal(out[rank * n_local : (rank + 1) * n_local].to(mine.device), mine), f"rank {rank}: own shard misplaced"

    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # per-rank account, so that the first multi-GPU run can be read from one line: every rank's kernel times per step
    # (HIP events on its launch stream), its wall time per step, and the all-gather alone (blocking, timed with events
    # after the measured loop, outside it)
    mine_ms = {"rank": rank, "stft_ms": k1_ms / max(1, args.steps), "finish_ms": k2_ms / max(1, args.steps),
               "project_ms": k3_ms / max(1, args.steps), "step_ms_median": step_ms[len(step_ms) // 2],
               "wall_ms_per_step": elapsed / args.steps * 1e3, "gather_alone_ms": None}
    if world > 1:
        from ser_amd.distributed import all_gather_rows

        feats_alone = engine.features(audio, offsets, lengths, SAMPLE_RATE, flags)
        if args.rehearse_on_one_gpu:
            feats_alone = feats_alone.cpu()
        all_gather_rows(feats_alone, n_total)   # warm
        torch.cuda.synchronize()
        dist.barrier()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_g = time.perf_counter()
        g0.record()
        for _ in range(3):
            all_gather_rows(feats_alone, n_total)
        g1.record()
        torch.cuda.synchronize()
        mine_ms["gather_alone_ms"] = max(g0.elapsed_time(g1), (time.perf_counter() - t_g) * 1e3) / 3
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_ms)
    else:
        per_rank = [mine_ms]

    comm = {"world": world, "backend": None, "transport": None, "per_rank": per_rank,
            "gather": "one all_gather_into_tensor of [clips_per_gpu, 187] float32 per rank and step (9.35 MB per rank at "
                      "config 3), left in flight under the next step's kernels"}
    if world > 1:
        comm["backend"] = "rccl (torch.distributed 'nccl')" if dist.get_backend() == "nccl" else dist.get_backend()
        if nccl_log and rank == 0:
            try:
                text = Path(nccl_log).read_text(errors="replace")
                counts = {key: text.count(tag) for key, tag in
                          (("p2p_ipc", "via P2P/IPC"), ("p2p_direct", "via P2P/direct"), ("p2p_cumem", "via P2P/CUMEM"),
                           ("shm", "via SHM"), ("net", "via NET"))}
                ranks_seen = [ln for ln in text.splitlines() if "nranks" in ln][:1]
                comm["transport"] = {"channel_connections": {k: v for k, v in counts.items() if v},
                                     "init": ranks_seen[0].split("NCCL INFO")[-1].strip() if ranks_seen else None}
            except OSError:
                comm["transport"] = "NCCL_DEBUG log not readable"
        elif rank == 0:
            comm["transport"] = "not probed (NCCL_DEBUG set by the caller)" if dist.get_backend() == "nccl" else "tcp (gloo)"

    if rank == 0:
        audio_seconds = n_total * CLIP_SECONDS * args.steps
        value = audio_seconds / elapsed
        # dominant kernel: algorithmic bytes of the rank's batch per step / its event time per step
        k1_s_per_step = (k1_ms / 1e3) / max(1, args.steps)
        achieved = n_local * ALGO_BYTES_PER_CLIP / k1_s_per_step / 1e9 if k1_s_per_step > 0 else 0.0
        traffic = frac_rocprof = rocprof_ms = profile_tag = None
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists():
            try:
                prof = json.loads(tfile.read_text())
                # PMC bytes of one profiled launch, scaled to this run's clips per launch
                traffic = prof["k_stft_frames_hbm_bytes_per_launch"] * (
                    n_local / max(1, k1_n / max(1, args.steps)) / prof["k_stft_frames_segments_per_launch_avg"])
                # the same fraction from the committed rocprofv3 kernel trace of this command (profiles/<tag>_kernel_stats.csv):
                # algorithmic bytes of a profiled launch / the kernel's average duration there / peak
                profile_tag = prof.get("tag")
                rocprof_ms = prof.get("k_stft_frames_rocprof_avg_us") / 1e3
                frac_rocprof = (prof["k_stft_frames_algorithmic_bytes_per_launch_avg"] / (rocprof_ms / 1e3) / 1e9) / HBM_PEAK_GBS
            except Exception:
                pass
        line = {
            "metric": "audio-seconds featurized/sec (whole node)",
            "value": value,
            "unit": "audio-seconds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "step_ms_min": step_ms[0],
            "step_ms_median": step_ms[len(step_ms) // 2],
            "step_ms_max": step_ms[-1],
            "comm": comm,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not args.rehearse_on_one_gpu else "synthetic (REHEARSAL: all ranks on one GPU, gloo)",
            "parity": "features checked against oracle/ (numpy/scipy restatement of librosa 0.11) by tests/ -m gpu; parity with "
                      "librosa itself is unpinned: the reference holds no numeric vector there and librosa is not installable here",
            "config": {
                "workload": "config 3: 100k x 3 s clips @16 kHz sharded by utterance, 12 500 clips per GPU, "
                            "187-dim fast-profile features (mfcc40|chroma12|mel128|contrast7) + all-gather",
                "descriptor_upload": "skipped after the first step (identical plan replayed: uniform clip batch)",
                "clips_per_gpu": n_local,
                "clip_samples": CLIP_SAMPLES,
                "feature_dim": FEATURE_DIM,
                "parallelism": f"dp{world} by utterance",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_stft_frames",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_source": "HIP events on the launch stream, this run",
                "frac_rocprof": frac_rocprof,
                "frac_rocprof_source": None if profile_tag is None else
                    f"profiles/{profile_tag}_kernel_stats.csv: k_stft_frames AverageNs = {rocprof_ms:.3f} ms per 12 500-clip launch "
                    "(rocprofv3 --kernel-trace --stats of this command, committed; kernels run 5-15 % longer under the "
                    "profiler than by events in the same process, DESIGN.md section 5)",
                "traffic": traffic,
                "traffic_source": f"profiles/traffic.json (tag {profile_tag}: rocprofv3 PMC passes of this command, TCC FETCH/WRITE_SIZE, "
                                  "guide's gfx950 corrections), scaled to this run's clips per launch; not re-measured live",
                "launches_per_step": k1_n / max(1, args.steps),
                "avg_launch_ms": k1_ms / max(1, k1_n),
                "finish_kernels_ms_per_step": k2_ms / max(1, args.steps),
                "project_kernel_ms_per_step": k3_ms / max(1, args.steps),
                "whole_path_frac": (value / world) * (ALGO_BYTES_PER_CLIP / CLIP_SECONDS) / 1e9 / HBM_PEAK_GBS,
                # what actually binds this kernel (SURVEY.md 8d asks to say so): vector-ALU instruction issue.
                # Useful butterfly arithmetic of a 2048-point real FFT, 2.5 N log2 N = 56 320 flop per frame,
                # against the packed-f32 vector peak of the guide.
                "binding": "valu",
                "valu_achieved_tflops": n_local * FRAMES_PER_CLIP * FFT_FLOP_PER_FRAME / k1_s_per_step / 1e12
                                        if k1_s_per_step > 0 else 0.0,
                "valu_peak_tflops": VALU_F32_PEAK_TFLOPS,
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
