"""The N > 1 path on CPU: contiguous sharding by utterance and the single all-gather that
assembles the feature matrix, with the gloo backend and world_size 2 (uneven shards included)."""

import os
import socket

import numpy as np
import pytest

from ser_amd.distributed import shard_bounds, shard_range


def test_shard_bounds():
    assert shard_bounds(100_000, 8).tolist() == [0, 12500, 25000, 37500, 50000, 62500, 75000, 87500, 100000]
    assert shard_bounds(10, 4).tolist() == [0, 3, 6, 9, 10]
    assert shard_bounds(3, 4).tolist() == [0, 1, 2, 3, 3]
    assert shard_bounds(0, 2).tolist() == [0, 0, 0]
    assert shard_range(3600, 1, 2) == (1800, 3600)
    covered = [shard_range(17, r, 5) for r in range(5)]
    assert covered[0][0] == 0 and covered[-1][1] == 17
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    with pytest.raises(ValueError):
        shard_bounds(5, 0)


def _worker(rank, world, port, n_total, dim, result_dir):
    import torch
    import torch.distributed as dist

    from ser_amd.distributed import all_gather_rows, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.arange(n_total * dim, dtype=torch.float32).reshape(n_total, dim)
    lo, hi = shard_range(n_total, rank, world)
    gathered = all_gather_rows(full[lo:hi].clone(), n_total)
    ok = torch.equal(gathered, full)
    # the non-blocking form: even shards give a work handle, uneven ones fall back to the blocking path
    from ser_amd.distributed import all_gather_rows_async

    g2, work = all_gather_rows_async(full[lo:hi].clone(), n_total)
    if work is not None:
        work.wait()
    ok = ok and torch.equal(g2, full) and ((work is not None) == ((hi - lo) * world == n_total))
    with open(os.path.join(result_dir, f"rank{rank}.txt"), "w") as f:
        f.write("ok" if ok else "mismatch")
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7, 1])
def test_all_gather_rows_gloo_world2(tmp_path, n_total):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, n_total, 5, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert (tmp_path / f"rank{r}.txt").read_text() == "ok"


def test_all_gather_single_process_passthrough():
    import torch

    from ser_amd.distributed import all_gather_rows

    x = torch.ones((4, 3))
    assert all_gather_rows(x, 4) is x
    from ser_amd.distributed import all_gather_rows_async

    full, work = all_gather_rows_async(x, 4)
    assert full is x and work is None
    with pytest.raises(ValueError):
        all_gather_rows(x, 5)
    assert np.asarray(x).shape == (4, 3)


def _fake_featurize(jobs, *, feature_flags, audio_read_config, workers=8, device_prepare=True):
    """Stand-in for the GPU featuriser: a row derived from the file name; '*bad*' files fail."""
    rows = np.full((len(jobs), 4), np.nan, dtype=np.float32)
    errors = [None] * len(jobs)
    for i, (path, _s, _d) in enumerate(jobs):
        if "bad" in path:
            errors[i] = OSError(f"cannot decode {path}")
        else:
            k = int(path.split("_")[1].split(".")[0])
            rows[i] = [k, 10 * k, 100 * k, 0.5]
    return rows, errors


def _files_worker(rank, world, port, files, result_dir):
    import torch.distributed as dist

    from ser_amd.data_loader import LoaderSettings, featurize_files
    from ser_amd.types import FeatureFlags

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    settings = LoaderSettings(feature_flags=FeatureFlags(tonnetz=False))
    matrix, errors = featurize_files(files, settings=settings, distributed=True, featurize_jobs=_fake_featurize)
    np.save(os.path.join(result_dir, f"matrix{rank}.npy"), matrix)
    with open(os.path.join(result_dir, f"errors{rank}.txt"), "w") as f:
        f.write("\n".join("" if e is None else f"{type(e).__name__}: {e}" for e in errors))
    dist.destroy_process_group()


def test_featurize_files_distributed_gloo_world2(tmp_path):
    """files -> contiguous shards -> one gather: every rank ends with the single-process matrix (uneven
    shards: 7 files over 2 ranks); a file that fails on rank 1 is NaN everywhere, reported with its own
    exception on rank 1 and with a RuntimeError naming rank 1 on rank 0."""
    import torch.multiprocessing as mp

    from ser_amd.data_loader import LoaderSettings, featurize_files
    from ser_amd.types import FeatureFlags

    files = [f"/d/clip_{k}.wav" for k in range(5)] + ["/d/bad_5.wav", "/d/clip_6.wav"]
    single, single_err = featurize_files(files, settings=LoaderSettings(feature_flags=FeatureFlags(tonnetz=False)),
                                         featurize_jobs=_fake_featurize)
    assert single.dtype == np.float64 and single.shape == (7, 4) and isinstance(single_err[5], OSError)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_files_worker, args=(2, port, files, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = np.load(tmp_path / f"matrix{r}.npy")
        assert got.dtype == np.float64
        np.testing.assert_array_equal(got, single)          # NaN row included
        errs = (tmp_path / f"errors{r}.txt").read_text().split("\n")
        assert [bool(e) for e in errs] == [False] * 5 + [True, False]
        assert errs[5] == ("OSError: cannot decode /d/bad_5.wav" if r == 1 else
                           "RuntimeError: feature extraction of /d/bad_5.wav failed on rank 1")
