"""libserhip.so on a CPU-only box: it loads, exports every symbol include/serhip.h declares,
and its host-built tables / frame bookkeeping equal the oracle's.  No GPU compute is called."""

import re
import subprocess
from pathlib import Path

import numpy as np
import pytest
import scipy.fft
import scipy.signal

from oracle import librosa_restated as lr

ROOT = Path(__file__).resolve().parents[1]


def test_header_symbols_are_exported_and_bound(native_lib):
    from ser_amd import _native

    header = (ROOT / "include" / "serhip.h").read_text()
    declared = set(re.findall(r"\b(serhip_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    for name in declared:
        assert hasattr(native_lib, name), name
    assert native_lib.serhip_version() >= 100
    assert native_lib.serhip_last_error() is not None


def test_host_tables_equal_oracle(native_lib):
    from ser_amd._native import host_table

    for sr in (16000, 22050, 48000):
        np.testing.assert_array_equal(host_table(sr, 0), scipy.signal.get_window("hann", 2048).astype(np.float32))
        np.testing.assert_array_equal(host_table(sr, 1).reshape(128, 1025), lr.mel_filter(sr=sr, n_fft=2048))
        edges = np.linspace(-0.5, 0.5, 101)
        for idx in (0, 1, 37, 50, 99, 100):
            got = host_table(sr, 2, idx).reshape(12, 1025)
            want = lr.chroma_filter(sr=sr, n_fft=2048, tuning=edges[idx])
            np.testing.assert_allclose(got, want, rtol=2e-7, atol=0)
        dct = scipy.fft.dct(np.eye(128), axis=0, type=2, norm="ortho")[:40]
        np.testing.assert_allclose(host_table(sr, 3).reshape(40, 128), dct, atol=1e-7)
        freqs = lr.fft_frequencies(sr=sr, n_fft=2048)
        mask = np.flatnonzero((150.0 <= freqs) & (freqs < min(4000.0, sr / 2)))
        assert host_table(sr, 4).tolist() == [float(mask[0]), float(mask[-1])]


def test_host_table_errors(native_lib):
    from ser_amd._native import SerhipError, host_table

    with pytest.raises(SerhipError, match="sample_rate"):
        host_table(0, 0)
    with pytest.raises(SerhipError, match="tuning index"):
        host_table(16000, 2, 101)
    with pytest.raises(SerhipError, match="unknown table"):
        host_table(16000, 9)


def test_frame_count_matches_stft_bookkeeping(native_lib):
    from ser_amd._native import frame_count

    for n in (1, 100, 511, 512, 513, 1023, 1024, 1500, 2047, 2048, 2049, 5937, 24000, 48000, 57_600_000):
        padded = max(n, 512)
        n_fft = min(padded, 2048)
        assert frame_count(n) == lr.stft_frame_count(padded, n_fft, n_fft // 4), n
    assert frame_count(48000) == 94 and frame_count(24000) == 47


def test_radix16_butterfly_on_host(tmp_path):
    """The per-lane radix-16 FFT code shared with the kernels, compiled for the host."""
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include "fft_math.hpp"
#include <cstdio>
using namespace serhip;
int main() {
  v2f v[16];
  for (int i = 0; i < 16; ++i) v[i] = v2f{(float)(0.25 * i - 1.0), (float)((i * 7) % 5) - 2.0f};
  fft16(v);
  for (int k = 0; k < 16; ++k) printf("%.9g %.9g\n", v[SERHIP_REV4(k)].x, v[SERHIP_REV4(k)].y);
  return 0;
}''')
    clang = Path("/opt/rocm/lib/llvm/bin/clang++")
    if not clang.exists():
        pytest.skip("amdclang++ not available")
    exe = tmp_path / "t"
    subprocess.run([str(clang), "-O2", "-I", str(ROOT / "ser_amd" / "csrc"), "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    got = np.asarray(out, dtype=np.float64).reshape(16, 2)
    x = np.asarray([0.25 * i - 1.0 + 1j * (((i * 7) % 5) - 2.0) for i in range(16)])
    want = np.fft.fft(x)
    np.testing.assert_allclose(got[:, 0] + 1j * got[:, 1], want, atol=1e-5)


def test_fft_lane_mapping_emulation():
    """numpy emulation of the wave64 decomposition used by k_stft_frames (16 x 16 x 4 with two
    transposes and the real-FFT unpack): same index arithmetic as the kernel, checked vs rfft."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal(2048)
    z = x[0::2] + 1j * x[1::2]
    W = lambda N, e: np.exp(-2j * np.pi * e / N)  # noqa: E731
    lane = np.arange(64)
    A = np.fft.fft(np.stack([z[64 * n1 + lane] for n1 in range(16)], axis=1), axis=1)
    A = A * W(1024, lane[:, None] * np.arange(16)[None, :])
    lds = np.zeros(16 * 68, complex)
    for k1 in range(16):
        lds[k1 * 68 + lane] = A[:, k1]
    k1b, q = lane >> 2, lane & 3
    B = np.fft.fft(np.stack([lds[k1b * 68 + 4 * m1 + q] for m1 in range(16)], axis=1), axis=1)
    B = B * W(64, q[:, None] * np.arange(16)[None, :])
    Bq = B.reshape(16, 4, 16)
    u0, u1, u2, u3 = Bq[:, 0] + Bq[:, 2], Bq[:, 1] + Bq[:, 3], Bq[:, 0] - Bq[:, 2], -1j * (Bq[:, 1] - Bq[:, 3])
    vq = np.stack([u0 + u1, u0 - u1, u2 + u3, u2 - u3], axis=1).reshape(64, 16)
    j2 = ((q & 1) << 1) | (q >> 1)
    buf = np.zeros(1088, complex)
    for j1 in range(16):
        buf[k1b + 16 * j1 + 260 * j2] = vq[:, j1]
    X = np.zeros(1025, complex)
    for r in range(8):
        k = lane + 64 * r
        kk = (1024 - k) & 1023
        za, zb = buf[k + 4 * (r >> 2)], buf[kk + 4 * (kk >> 8)]
        e2, o2 = za + np.conj(zb), -1j * (za - np.conj(zb))
        t2 = o2 * W(2048, k)
        X[k] = 0.5 * (e2 + t2)
        X[1024 - k] = np.conj(0.5 * (e2 - t2))
    X[512] = np.conj(buf[512 + 8])
    np.testing.assert_allclose(X, np.fft.rfft(x), atol=1e-10)


def test_header_is_plain_c(tmp_path):
    """include/serhip.h is the drop-in boundary: it must compile as C99 (extern "C" guards, stdint types,
    no C++ or torch types) so cgo / JNI / ctypes-style bindings can consume it."""
    import shutil
    import subprocess
    from pathlib import Path

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    header = Path(__file__).resolve().parent.parent / "include" / "serhip.h"
    src = tmp_path / "use.c"
    src.write_text('#include "serhip.h"\nint main(void) { return serhip_version() > 0 ? 0 : 1; }\n')
    res = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-c", str(src),
                          "-I", str(header.parent), "-o", str(tmp_path / "use.o")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_hand_counted_waits_touch_no_inflight_register():
    """k_stft_frames loads its samples with inline asm and waits with a hand-counted vmcnt: the compiler does
    not know those registers are in flight.  scripts/check_inflight_regs.py compiles the device code and proves
    on the ISA that nothing reads, writes, copies or spills them before the wait, and that the store block the
    count relies on has exactly that many vector-memory operations (hipcc only, no GPU)."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    done = subprocess.run([sys.executable, str(root / "scripts" / "check_inflight_regs.py")], capture_output=True, text=True)
    assert done.returncode == 0, done.stdout + done.stderr
    assert "ok (3 kernels)" in done.stdout
    # the checker is not vacuous: a copy of a register in flight, or one store fewer in the block, is caught
    import re

    sys.path.insert(0, str(root / "scripts"))
    import check_inflight_regs as chk

    text = chk.device_asm().split("\n")
    start = next(i for i, l in enumerate(text) if l.startswith("_ZN6serhip13k_stft_framesILb1ELb1EE"))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    body = text[start + 1:end]
    assert chk.check_kernel("clean", body) == []
    loads = [i for i, l in enumerate(body) if "global_load_dwordx2" in l and "ASMSTART" in body[i - 1]]
    dest = re.search(r"global_load_dwordx2 v\[(\d+):", body[loads[-1]]).group(1)
    bad = body[:loads[-1] + 2] + [f"\tv_mov_b32_e32 v127, v{dest}"] + body[loads[-1] + 2:]
    assert any("touches registers in flight" in p for p in chk.check_kernel("copy", bad))
    stores = [i for i, l in enumerate(body) if "global_store_dwordx4" in l]
    fewer = body[:stores[-1]] + body[stores[-1] + 1:]
    assert any("sure to follow the loads" in p for p in chk.check_kernel("fewer", fewer))


def test_comm_entry_points_reject_bad_arguments_without_touching_rccl(native_lib):
    """serhip_comm_* argument checks come before anything loads RCCL or touches a GPU."""
    import ctypes

    assert native_lib.serhip_comm_unique_id(None) == -1
    assert b"NULL" in native_lib.serhip_last_error()
    handle = ctypes.c_void_p()
    assert native_lib.serhip_comm_create(None, None, 0, 1, ctypes.byref(handle)) == -1
    assert native_lib.serhip_allgather_rows(None, None, None, 187, None, None) == -1
    native_lib.serhip_comm_destroy(None)   # a no-op, like free(NULL)


def test_rccl_that_cannot_be_loaded_is_reported_not_crashed():
    """with RCCL unloadable the comm entry points must return SERHIP_E_UNSUPPORTED (-4) with a message
    (one dlerror() call; the old code built a std::string from NULL).  The loader's result is cached per process,
    so this runs in a child whose SERHIP_RCCL_LIBRARY names a file that does not exist.  No GPU is touched:
    serhip_comm_unique_id consults the loader before anything else."""
    import os
    import sys

    code = (
        "import ctypes, sys\n"
        f"lib = ctypes.CDLL({str(ROOT / 'ser_amd' / 'libserhip.so')!r})\n"
        "lib.serhip_last_error.restype = ctypes.c_char_p\n"
        "buf = (ctypes.c_ubyte * 128)()\n"
        "rc = lib.serhip_comm_unique_id(buf)\n"
        "print(rc, lib.serhip_last_error().decode())\n"
    )
    env = dict(os.environ, SERHIP_RCCL_LIBRARY="/nonexistent/librccl-not-here.so.1")
    done = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert done.returncode == 0, done.stderr
    rc, message = done.stdout.strip().split(" ", 1)
    assert int(rc) == -4
    assert "RCCL could not be loaded" in message and "librccl-not-here" in message
