"""ctypes binding of ``libserhip.so`` (C ABI declared in ``include/serhip.h``).

The library is built in-tree by ``__graft_entry__.build()`` (``hipcc
--offload-arch=gfx950``).  There is no CPU fallback: if the shared object is
missing, or a compute entry point is called without a GPU, this module raises.
"""

from __future__ import annotations

import ctypes
import os
import threading
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p
from pathlib import Path

LIB_NAME = "libserhip.so"
LIB_PATH = Path(__file__).resolve().parent / LIB_NAME

FEAT_MFCC, FEAT_CHROMA, FEAT_MEL, FEAT_CONTRAST = 1, 2, 4, 8

#: every symbol ``include/serhip.h`` declares: name -> (restype, argtypes)
SIGNATURES: dict[str, tuple] = {
    "serhip_version": (c_int, []),
    "serhip_last_error": (c_char_p, []),
    "serhip_create": (c_int, [c_int, POINTER(c_void_p)]),
    "serhip_destroy": (None, [c_void_p]),
    "serhip_plan_fast": (c_int, [c_void_p, c_int, c_uint32, POINTER(c_void_p)]),
    "serhip_plan_destroy": (None, [c_void_p]),
    "serhip_plan_feature_dim": (c_int, [c_void_p]),
    "serhip_fast_features": (
        c_int,
        [c_void_p, c_void_p, c_int64, POINTER(c_int64), POINTER(c_int64), c_int64, c_void_p, c_void_p, c_void_p],
    ),
    "serhip_fast_features_noalloc": (
        c_int,
        [c_void_p, c_void_p, c_int64, POINTER(c_int64), POINTER(c_int64), c_int64, c_void_p, c_void_p, c_void_p],
    ),
    "serhip_pool_windows": (
        c_int,
        [c_void_p, c_void_p, c_int32, c_int32, POINTER(c_int32), POINTER(c_int32), c_int32, c_int, c_void_p, c_void_p],
    ),
    "serhip_whisper_logmel": (c_int, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p]),
    "serhip_prepare_audio": (
        c_int,
        [c_void_p, c_void_p, c_int64, c_int, POINTER(c_int64), POINTER(c_int64), POINTER(c_int32), POINTER(c_int64),
         c_int64, c_void_p, c_int64, c_void_p],
    ),
    "serhip_mlp_predict": (
        c_int,
        [c_void_p, c_void_p, c_int, c_int64, c_int32, c_void_p, c_void_p, c_int32, POINTER(c_int32),
         POINTER(c_void_p), POINTER(c_void_p), c_int, c_int, c_void_p, c_void_p, c_void_p],
    ),
    "serhip_tuning_from_peaks": (c_int, [c_void_p, POINTER(c_float), POINTER(c_int32), c_int32, POINTER(c_float)]),
    "serhip_comm_unique_id": (c_int, [c_void_p]),
    "serhip_comm_create": (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "serhip_allgather_rows": (c_int, [c_void_p, c_void_p, POINTER(c_int64), c_int32, c_void_p, c_void_p]),
    "serhip_comm_destroy": (None, [c_void_p]),
    "serhip_profile_enable": (c_int, [c_void_p, c_int]),
    "serhip_profile_read": (c_int, [c_void_p, c_int, POINTER(c_double), POINTER(c_int64)]),
    "serhip_read_wav_batch": (
        c_int,
        [c_void_p, POINTER(c_int64), c_int64, c_void_p, c_void_p, POINTER(c_int64), POINTER(c_int64), c_void_p, c_int64,
         c_int, c_void_p],
    ),
    "serhip_stat_files": (c_int, [c_void_p, POINTER(c_int64), c_int64, c_int, POINTER(c_int64)]),
    "serhip_parallel_memcpy": (c_int, [c_void_p, c_void_p, c_int64, c_int]),
    "serhip_gather_f32": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), c_int64, c_void_p, c_int64, c_int, POINTER(c_int32)]),
    "serhip_flac_probe": (c_int, [c_void_p, c_int64, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int64)]),
    "serhip_flac_decode": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, POINTER(c_int64)]),
    "serhip_host_table": (c_int64, [c_int, c_int, c_int, POINTER(c_float), c_int64]),
    "serhip_frame_count": (c_int64, [c_int64]),
}


class SerhipError(RuntimeError):
    """A libserhip entry point returned a negative status."""

    def __init__(self, code: int, message: str) -> None:
        super().__init__(f"libserhip error {code}: {message}")
        self.code = code


_lib = None
_lib_lock = threading.Lock()


def load_library() -> ctypes.CDLL:
    """Load ``libserhip.so`` once and attach the prototypes.  Raises if it is not built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        path = Path(os.environ.get("SERHIP_LIBRARY", str(LIB_PATH)))
        if not path.exists():
            raise ImportError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  ser_amd has no CPU fallback."
            )
        # torch ships its own HIP runtime (libamdhip64.so.7 under torch/lib) and loads it by absolute
        # path.  libserhip.so must bind to THAT copy -- it is handed torch's device pointers and
        # streams -- so torch has to be in the process first; loaded the other way round, the system
        # runtime and torch's both initialise and the second one finds no device.
        try:
            import torch  # noqa: F401
        except ImportError:  # pragma: no cover - a torch-less caller owns its HIP runtime
            pass
        lib = ctypes.CDLL(str(path))
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here means the .so is stale
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
        return lib


def check(status: int) -> int:
    if status < 0:
        message = load_library().serhip_last_error()
        raise SerhipError(int(status), message.decode("utf-8", "replace") if message else "")
    return status


def host_table(sample_rate: int, which: int, arg: int = 0):
    """Host-built constant table of the fast plan as a float32 numpy array (no GPU needed)."""
    import numpy as np

    lib = load_library()
    n = check(lib.serhip_host_table(sample_rate, which, arg, None, 0))
    out = np.empty(n, dtype=np.float32)
    check(lib.serhip_host_table(sample_rate, which, arg, out.ctypes.data_as(POINTER(c_float)), n))
    return out


def frame_count(n_samples: int) -> int:
    return int(check(load_library().serhip_frame_count(int(n_samples))))


class Context:
    """One ``serhip_ctx`` (one GPU).  Not thread-safe; create one per thread or serialise calls."""

    def __init__(self, device_index: int = 0) -> None:
        self._lib = load_library()
        handle = c_void_p()
        check(self._lib.serhip_create(int(device_index), ctypes.byref(handle)))
        self._handle = handle
        self.device_index = int(device_index)
        self._plans: dict[tuple[int, int], "Plan"] = {}

    @property
    def handle(self) -> c_void_p:
        if self._handle is None:
            raise RuntimeError("serhip context already closed")
        return self._handle

    def plan(self, sample_rate: int, flags: int) -> "Plan":
        key = (int(sample_rate), int(flags))
        plan = self._plans.get(key)
        if plan is None:
            plan = Plan(self, int(sample_rate), int(flags))
            self._plans[key] = plan
        return plan

    def profile_enable(self, on: bool) -> None:
        check(self._lib.serhip_profile_enable(self.handle, 1 if on else 0))

    def profile_read(self, kind: int) -> tuple[float, int]:
        """(summed milliseconds, launch count) of the recorded launches of ``kind``; clears them."""
        ms, n = c_double(0.0), c_int64(0)
        check(self._lib.serhip_profile_read(self.handle, int(kind), ctypes.byref(ms), ctypes.byref(n)))
        return float(ms.value), int(n.value)

    def close(self) -> None:
        if self._handle is not None:
            for plan in self._plans.values():
                plan.close()
            self._plans.clear()
            self._lib.serhip_destroy(self._handle)
            self._handle = None

    def __del__(self) -> None:  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """One ``serhip_plan``: constant tables for (sample_rate, feature flags) resident in HBM."""

    def __init__(self, ctx: Context, sample_rate: int, flags: int) -> None:
        self._ctx = ctx
        self._lib = ctx._lib
        handle = c_void_p()
        check(self._lib.serhip_plan_fast(ctx.handle, sample_rate, flags, ctypes.byref(handle)))
        self._handle = handle
        self.sample_rate = sample_rate
        self.flags = flags
        self.feature_dim = int(self._lib.serhip_plan_feature_dim(handle))

    @property
    def handle(self) -> c_void_p:
        if self._handle is None:
            raise RuntimeError("serhip plan already closed")
        return self._handle

    def close(self) -> None:
        if self._handle is not None:
            self._lib.serhip_plan_destroy(self._handle)
            self._handle = None
