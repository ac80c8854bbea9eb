"""ser_amd -- MI355X-native fast-profile acoustic features for jsugg/ser.

Drop-in for the reference's handcrafted feature path: the
call signatures of ``ser/_internal/utils/dsp.py``, ``repr/handcrafted.py``,
``repr/backend.py``, ``features/feature_extractor.py`` and ``pool/*.py`` over
hand-written gfx950 HIP kernels reached through the C ABI in
``include/serhip.h``.  Importing this package touches neither the GPU nor the
shared library; the first extraction call does, and raises if either is missing.
"""

from .types import (  # noqa: F401
    FAST_PATH_FLAGS,
    AudioReadConfig,
    EncodedSequence,
    FeatureBackend,
    FeatureFlags,
    FeatureFrame,
    PoolingWindow,
    VectorFeatureBackend,
    coerce_feature_flags,
    default_feature_flags,
    overlap_frame_mask,
    overlap_frame_range,
)

__version__ = "0.1.0"

_LAZY = {
    "HandcraftedBackend": ("handcrafted", "HandcraftedBackend"),
    "extract_feature_from_signal": ("dsp", "extract_feature_from_signal"),
    "extract_feature": ("feature_extractor", "extract_feature"),
    "extract_feature_frames": ("feature_extractor", "extract_feature_frames"),
    "extended_extract_feature": ("feature_extractor", "extended_extract_feature"),
    "FeatureSettings": ("feature_extractor", "FeatureSettings"),
    "mean_std_pool": ("pool", "mean_std_pool"),
    "temporal_pooling_windows": ("pool", "temporal_pooling_windows"),
    "read_audio_file": ("audio_io", "read_audio_file"),
    "log_mel_spectrogram": ("whisper_frontend", "log_mel_spectrogram"),
    "get_engine": ("engine", "get_engine"),
    "load_data": ("data_loader", "load_data"),
    "load_checked_fast_data": ("data_loader", "load_checked_fast_data"),
    "featurize_files": ("data_loader", "featurize_files"),
    "process_file": ("data_loader", "process_file"),
    "LoaderSettings": ("data_loader", "LoaderSettings"),
    "write_prepared_feature_payload": ("prepared_cache", "write_prepared_feature_payload"),
    "read_prepared_feature_payload": ("prepared_cache", "read_prepared_feature_payload"),
    "GpuEmotionClassifier": ("classifier", "GpuEmotionClassifier"),
    "merge_label_runs": ("label_runs", "merge_label_runs"),
    "classify_sequence": ("label_runs", "classify_sequence"),
    "LabelRuns": ("label_runs", "LabelRuns"),
    "default_split_utterances": ("data_loader", "default_split_utterances"),
}


def __getattr__(name: str):
    target = _LAZY.get(name)
    if target is None:
        raise AttributeError(f"module 'ser_amd' has no attribute {name!r}")
    import importlib

    return getattr(importlib.import_module(f".{target[0]}", __name__), target[1])


__all__ = [
    "FAST_PATH_FLAGS", "AudioReadConfig", "EncodedSequence", "FeatureBackend", "FeatureFlags", "FeatureFrame",
    "PoolingWindow", "VectorFeatureBackend", "coerce_feature_flags", "default_feature_flags", "overlap_frame_mask", "overlap_frame_range", *_LAZY,
]
