"""Shared tolerance rules for comparing GPU features with the oracle (BASELINE.md parity gates)."""

import numpy as np

MFCC = slice(0, 40)
CHROMA = slice(40, 52)
MEL = slice(52, 180)
CONTRAST = slice(180, 187)

#: log-mel / MFCC gate: rtol 1e-4 with atol 2e-3 (dB-scaled units, values span +-hundreds)
MFCC_RTOL, MFCC_ATOL = 1e-4, 2e-3
#: mel power and chroma gate: rtol 1e-4 plus atol 1e-6 * max(vector)
LIN_RTOL, LIN_ATOL_REL = 1e-4, 1e-6


def block_errors(got, want):
    """Per-block worst normalised error (<= 1 passes) for 187-dim rows."""
    got = np.asarray(got, dtype=np.float64).reshape(-1, 187)
    want = np.asarray(want, dtype=np.float64).reshape(-1, 187)
    out = {}
    d = np.abs(got[:, MFCC] - want[:, MFCC])
    out["mfcc"] = float(np.max(d / (MFCC_ATOL + MFCC_RTOL * np.abs(want[:, MFCC]))))
    for name, sl in (("mel", MEL), ("chroma", CHROMA)):
        d = np.abs(got[:, sl] - want[:, sl])
        scale = np.max(np.abs(want[:, sl]), axis=1, keepdims=True)
        out[name] = float(np.max(d / (LIN_ATOL_REL * scale + LIN_RTOL * np.abs(want[:, sl]) + 1e-300)))
    out["contrast"] = float(np.max(np.abs(got[:, CONTRAST])))
    return out


def assert_features_close(got, want, *, skip_chroma_rows=()):
    got = np.asarray(got, dtype=np.float64).reshape(-1, 187)
    want = np.asarray(want, dtype=np.float64).reshape(-1, 187)
    assert got.shape == want.shape
    np.testing.assert_allclose(got[:, MFCC], want[:, MFCC], rtol=MFCC_RTOL, atol=MFCC_ATOL, err_msg="mfcc")
    scale = np.max(np.abs(want[:, MEL]), axis=1, keepdims=True)
    assert np.all(np.abs(got[:, MEL] - want[:, MEL]) <= LIN_ATOL_REL * scale + LIN_RTOL * np.abs(want[:, MEL])), "mel"
    keep = np.ones(got.shape[0], dtype=bool)
    keep[list(skip_chroma_rows)] = False
    scale = np.max(np.abs(want[:, CHROMA]), axis=1, keepdims=True)
    ok = np.abs(got[:, CHROMA] - want[:, CHROMA]) <= LIN_ATOL_REL * scale + LIN_RTOL * np.abs(want[:, CHROMA])
    assert np.all(ok[keep]), f"chroma rows {np.nonzero(~ok.all(axis=1) & keep)[0]}"
    assert not got[:, CONTRAST].any(), "contrast must be identically 0 (reference quirk)"
