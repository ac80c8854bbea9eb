"""CPU oracle for the fast-profile acoustic feature path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the timed CPU baseline.  ``ser_amd`` never
imports this package and fails loudly when its HIP library is missing.

What is restated
----------------
* ``oracle.librosa_restated`` -- the arithmetic that the reference delegates to
  third-party ``librosa==0.11.0`` (reference ``uv.lock:867-868``), which is NOT
  vendored under ``/root/reference`` and NOT installed here.  It is restated
  from librosa's published algorithm (numpy + scipy only).
* ``oracle.ser_fast_path`` -- the reference's own glue around it
  (``ser/_internal/utils/dsp.py:38-151``, ``repr/handcrafted.py:65-137``,
  ``repr/backend.py:81-111``, ``pool/*.py``, ``utils/audio_utils.py:28-60``).
* ``oracle.whisper_logmel`` -- ``whisper.audio.log_mel_spectrogram``
  (openai-whisper 20250625, reference ``uv.lock:1448-1449``), restated.

Parity pinning status
---------------------
PARITY PARTLY UNPINNED.  The reference's tests stub every librosa call on
this path (``tests/suites/unit/utils/test_dsp.py:21-48``) and hold no golden
feature vectors, and the reference package cannot be imported under this
container's Python 3.10 (it needs >=3.12, PEP 695 syntax).  What pins the
oracle instead (``tests/test_oracle_pins.py``):

* every known-answer the reference's tests DO hold for the path (windowing,
  pooling, timestamps, error strings, 193/187 dims) -- SURVEY.md appendix B;
* independent in-container implementations of the same published formulas:
  ``transformers.audio_utils`` (``mel_filter_bank`` slaney/slaney,
  ``chroma_filter_bank``, ``power_to_db``, ``spectrogram``), ``torch.stft``,
  ``scipy.fft.dct`` and the HF ``WhisperFeatureExtractor`` numpy path.

``piptrack`` / ``estimate_tuning`` (the chroma tuning estimate) and the exact
dtype placement inside librosa have no independent implementation here and
stay "parity unpinned".
"""
