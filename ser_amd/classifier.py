"""The fast profile's emotion classifier on the GPU.

The reference trains ``Pipeline(StandardScaler(), MLPClassifier(hidden_layer_sizes=(300,), ...))``
(``ser/_internal/models/training_support.py:87-106``) and at inference calls ``model.predict`` and
``model.predict_proba`` on the float64 feature matrix (``ser/_internal/models/fast_path.py:180-192``).
:class:`GpuEmotionClassifier` takes the fitted sklearn object, uploads its parameters once and
runs the same forward pass (float64) through ``serhip_mlp_predict``; it exposes ``predict``,
``predict_proba`` and ``classes_`` like the sklearn model, so it drops into
``predict_emotions_detailed_with_model`` unchanged, and ``predict_device`` for feature matrices
that are already in HBM (no D2H of the features).  Training stays with sklearn.
"""

from __future__ import annotations

import ctypes
from ctypes import POINTER, c_int32, c_void_p

import numpy as np

from . import _native
from .engine import get_engine

_ACTIVATIONS = {"identity": 0, "relu": 1, "tanh": 2, "logistic": 3}


def _unwrap(model):
    """(scaler or None, MLPClassifier) from a fitted Pipeline or a bare classifier."""
    steps = getattr(model, "steps", None)
    if steps is None:
        return None, model
    scaler = None
    for name, step in steps[:-1]:
        if type(step).__name__ == "StandardScaler":
            if scaler is not None:
                raise NotImplementedError("more than one StandardScaler step")
            scaler = step
        elif step is not None and step != "passthrough":
            raise NotImplementedError(f"pipeline step {name!r} ({type(step).__name__}) has no GPU counterpart")
    return scaler, steps[-1][1]


class GpuEmotionClassifier:
    """GPU forward pass of a fitted sklearn ``MLPClassifier`` (optionally behind a ``StandardScaler``)."""

    def __init__(self, *, classes, coefs, intercepts, hidden_activation: str = "relu",
                 out_activation: str = "softmax", mean=None, scale=None, device_index: int | None = None) -> None:
        if hidden_activation not in _ACTIVATIONS:
            raise ValueError(f"unknown activation {hidden_activation!r}")
        if out_activation not in ("softmax", "logistic"):
            raise NotImplementedError(f"output activation {out_activation!r} is not a classifier output")
        if len(coefs) != len(intercepts) or not coefs:
            raise ValueError("coefs and intercepts must be non-empty and the same length")
        self.classes_ = np.asarray(classes)
        self._engine = get_engine(device_index)
        torch = self._engine._torch
        dev = self._engine.device

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).to(dev)

        self._weights = [up(w) for w in coefs]
        self._biases = [up(b) for b in intercepts]
        sizes = [int(self._weights[0].shape[0])] + [int(w.shape[1]) for w in self._weights]
        for w, b, n_in, n_out in zip(self._weights, self._biases, sizes[:-1], sizes[1:]):
            if tuple(w.shape) != (n_in, n_out) or tuple(b.shape) != (n_out,):
                raise ValueError("layer shapes are inconsistent")
        self._sizes = np.asarray(sizes, dtype=np.int32)
        self._mean = up(mean) if mean is not None else None
        self._scale = up(scale) if scale is not None else None
        for v in (self._mean, self._scale):
            if v is not None and tuple(v.shape) != (sizes[0],):
                raise ValueError("scaler statistics do not match the input width")
        self._hidden = _ACTIVATIONS[hidden_activation]
        self._out = 0 if out_activation == "softmax" else 1
        self.n_classes = sizes[-1] if self._out == 0 else 2
        if self.classes_.shape != (self.n_classes,):
            raise ValueError(f"{self.classes_.shape[0]} class labels for {self.n_classes} outputs")
        self.n_features_in_ = sizes[0]

    @classmethod
    def from_sklearn(cls, model, *, device_index: int | None = None) -> "GpuEmotionClassifier":
        scaler, mlp = _unwrap(model)
        if not hasattr(mlp, "coefs_"):
            raise ValueError("the classifier is not fitted")
        mean = scale = None
        if scaler is not None:
            mean = scaler.mean_ if getattr(scaler, "with_mean", True) else None
            scale = scaler.scale_ if getattr(scaler, "with_std", True) else None
        return cls(classes=mlp.classes_, coefs=mlp.coefs_, intercepts=mlp.intercepts_,
                   hidden_activation=mlp.activation, out_activation=mlp.out_activation_, mean=mean, scale=scale,
                   device_index=device_index)

    # ------------------------------------------------------------------ device path
    def predict_device(self, features):
        """``(label_index int32 [n], proba float64 [n, n_classes])`` device tensors for a feature
        matrix that is a float32/float64 torch tensor (kept where it is) or a numpy array."""
        eng = self._engine
        torch = eng._torch
        if isinstance(features, torch.Tensor):
            x = features.to(eng.device)
            if x.dtype not in (torch.float32, torch.float64):
                x = x.to(torch.float64)
            x = x.contiguous()
        else:
            arr = np.asarray(features)
            arr = arr.astype(np.float64 if arr.dtype != np.float32 else np.float32, copy=False)
            x = eng.to_device_staged(np.ascontiguousarray(arr))   # a large matrix goes through the pinned ring
        if x.dim() != 2 or int(x.shape[1]) != self.n_features_in_:
            raise ValueError(
                f"X has {tuple(x.shape)[1] if x.dim() == 2 else '?'} features, but the classifier "
                f"is expecting {self.n_features_in_} features as input."
            )
        n = int(x.shape[0])
        n_layers = len(self._weights)
        wptr = (c_void_p * n_layers)(*[c_void_p(w.data_ptr()) for w in self._weights])
        bptr = (c_void_p * n_layers)(*[c_void_p(b.data_ptr()) for b in self._biases])
        with torch.cuda.device(eng.device), eng._lock:
            proba = torch.empty((n, self.n_classes), dtype=torch.float64, device=eng.device)
            label = torch.empty((n,), dtype=torch.int32, device=eng.device)
            if n:
                _native.check(
                    eng._lib.serhip_mlp_predict(
                        eng._ctx.handle, c_void_p(x.data_ptr()), 1 if x.dtype == torch.float64 else 0, n,
                        self.n_features_in_,
                        c_void_p(self._mean.data_ptr()) if self._mean is not None else None,
                        c_void_p(self._scale.data_ptr()) if self._scale is not None else None,
                        n_layers, self._sizes.ctypes.data_as(POINTER(c_int32)),
                        ctypes.cast(wptr, POINTER(c_void_p)), ctypes.cast(bptr, POINTER(c_void_p)),
                        self._hidden, self._out, c_void_p(proba.data_ptr()), c_void_p(label.data_ptr()),
                        eng._stream(),
                    )
                )
        return label, proba

    # ------------------------------------------------------------------ sklearn-shaped API
    def predict_proba(self, X) -> np.ndarray:
        return self.predict_device(X)[1].cpu().numpy()

    def predict(self, X) -> np.ndarray:
        return self.classes_[self.predict_device(X)[0].cpu().numpy()]
