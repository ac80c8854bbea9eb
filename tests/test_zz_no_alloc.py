"""The allocation-free entry point ``serhip_fast_features_noalloc`` against ``serhip_fast_features``
(prompt item 7; no counterpart in the reference).  Called through the C ABI with raw pointers.

Bar: bit-identical outputs (same kernels, integer-exact planning), on two calls in a row, the second with
a smaller input; limits rejected with a clear error; device memory in use does not grow across the calls.
"""

import ctypes
from ctypes import POINTER, c_int64, c_void_p

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SR = 16000


def _call(lib, fn_name, plan, audio, off, length, out, tuning, stream):
    off = np.ascontiguousarray(off, dtype=np.int64)
    length = np.ascontiguousarray(length, dtype=np.int64)
    return getattr(lib, fn_name)(plan.handle, c_void_p(audio.data_ptr()), int(audio.numel()),
                                 off.ctypes.data_as(POINTER(c_int64)), length.ctypes.data_as(POINTER(c_int64)),
                                 int(off.size), c_void_p(out.data_ptr()),
                                 c_void_p(tuning.data_ptr()) if tuning is not None else None, c_void_p(stream))


def test_noalloc_entry_point_matches_fast_features():
    import torch

    from ser_amd import FeatureFlags, _native
    from ser_amd.engine import get_engine
    from ser_amd.synth import sample_wav_like

    eng = get_engine()
    lib = _native.load_library()
    flags = FeatureFlags(tonnetz=False)
    plan = eng._ctx.plan(SR, flags.native_bits())
    stream = torch.cuda.current_stream(eng.device).cuda_stream

    def windows(n_samples, step=SR, size=3 * SR):
        starts = np.arange(0, n_samples, step, dtype=np.int64)
        return starts, np.minimum(starts + size, n_samples) - starts

    # call 1: a 20.4 s recording, 21 sliding windows incl. a tail shorter than 2048 samples (generic path);
    # call 2 (smaller, stale scratch of call 1 underneath): a 4.37 s recording, 5 windows
    cases = []
    for seed, n in ((5, 20 * SR + 1500), (6, 69937)):
        audio = torch.from_numpy(sample_wav_like(seed=seed, n_samples=n)).to(eng.device)
        off, length = windows(n)
        cases.append((audio, off, length))
    assert cases[0][2].min() < 2048 <= cases[0][2].max() and len(cases[1][1]) == 5

    expected = []
    for audio, off, length in cases:     # the existing entry point (also loads every kernel's code object)
        out = torch.empty((len(off), 187), dtype=torch.float32, device=eng.device)
        tun = torch.empty((len(off),), dtype=torch.float32, device=eng.device)
        _native.check(_call(lib, "serhip_fast_features", plan, audio, off, length, out, tun, stream))
        torch.cuda.synchronize()
        expected.append((out.cpu().numpy(), tun.cpu().numpy()))

    outs = [(torch.full((len(off), 187), float("nan"), dtype=torch.float32, device=eng.device),
             torch.full((len(off),), float("nan"), dtype=torch.float32, device=eng.device)) for _a, off, _l in cases]
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(eng.device)[0]
    for (audio, off, length), (out, tun) in zip(cases, outs):        # two calls in a row, no sync between
        _native.check(_call(lib, "serhip_fast_features_noalloc", plan, audio, off, length, out, tun, stream))
    torch.cuda.synchronize()
    free_after = torch.cuda.mem_get_info(eng.device)[0]
    assert free_after >= free_before, (free_before, free_after)      # nothing was allocated on the device
    for (want, want_t), (out, tun) in zip(expected, outs):
        got = out.cpu().numpy()
        assert np.isfinite(got).all()
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(tun.cpu().numpy(), want_t)

    # once more on a side stream without a host sync in between: the library orders the shared scratch
    side = torch.cuda.Stream(device=eng.device)
    audio, off, length = cases[0]
    out_a = torch.empty((len(off), 187), dtype=torch.float32, device=eng.device)
    out_b = torch.empty((len(cases[1][1]), 187), dtype=torch.float32, device=eng.device)
    _native.check(_call(lib, "serhip_fast_features_noalloc", plan, audio, off, length, out_a, None, stream))
    _native.check(_call(lib, "serhip_fast_features_noalloc", plan, cases[1][0], cases[1][1], cases[1][2], out_b, None,
                        side.cuda_stream))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out_a.cpu().numpy(), expected[0][0])
    np.testing.assert_array_equal(out_b.cpu().numpy(), expected[1][0])

    # the engine wrapper
    out_c = torch.empty((5, 187), dtype=torch.float32, device=eng.device)
    eng.features_into(cases[1][0], cases[1][1], cases[1][2], SR, flags, out_c)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out_c.cpu().numpy(), expected[1][0])

    # limits: too many frame slots, too many segments -> clear errors, nothing launched
    big = torch.zeros(70 * 3 * SR, dtype=torch.float32, device=eng.device)
    off = np.arange(70, dtype=np.int64) * 3 * SR
    out = torch.empty((70, 187), dtype=torch.float32, device=eng.device)
    rc = _call(lib, "serhip_fast_features_noalloc", plan, big, off, np.full(70, 3 * SR), out, None, stream)
    assert rc == -4 and b"6144" in lib.serhip_last_error() and b"serhip_fast_features" in lib.serhip_last_error()
    off = np.arange(300, dtype=np.int64) * 600
    out = torch.empty((300, 187), dtype=torch.float32, device=eng.device)
    rc = _call(lib, "serhip_fast_features_noalloc", plan, big, off, np.full(300, 600), out, None, stream)
    assert rc == -4 and b"256" in lib.serhip_last_error()
    rc = _call(lib, "serhip_fast_features_noalloc", plan, big, [big.numel() - 10], [100], out, None, stream)
    assert rc == -1 and b"outside the audio buffer" in lib.serhip_last_error()
    # exactly at the limit: 64 windows of 3 s = 6144 slots
    off = np.arange(64, dtype=np.int64) * 3 * SR
    out = torch.empty((64, 187), dtype=torch.float32, device=eng.device)
    _native.check(_call(lib, "serhip_fast_features_noalloc", plan, big, off, np.full(64, 3 * SR), out, None, stream))
    torch.cuda.synchronize()
    assert np.isfinite(out.cpu().numpy()).all()
