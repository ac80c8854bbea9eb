"""Containers other than RIFF/WAVE PCM16 (``ser_amd/audio_io.decode_bytes``): the lossless and uncompressed formats
the reference reads through ``librosa.load`` -> soundfile (``ser/_internal/utils/audio_utils.py:99-154``).

What pins each decoder (no libsndfile, soundfile or sample files exist in this image):
* AIFF / AIFF-C, AU, A-law / mu-law -- files written by the Python standard library (``aifc``, ``sunau``, ``audioop``),
  i.e. by code that is not this repository's;
* FLAC -- ``tests/flac_writer.py`` (an encoder written for these tests), ``hashlib.md5`` against the decoder's own
  MD5, and the stream's CRC-8 / CRC-16: parity with libsndfile's FLAC reader itself is unpinned.
All CPU tests: the decoders are host code (the FLAC one is native, reached through the C ABI)."""

import io
import struct
import warnings

import numpy as np
import pytest

from flac_writer import encode_frame, encode_stream, pcm_md5, stream_header
from ser_amd import audio_io
from ser_amd.audio_io import AudioDecodeError, decode_bytes, read_audio_file


def _signal(n, channels, bits, seed=0, tone=True):
    rng = np.random.default_rng(seed)
    top = (1 << (bits - 1)) - 1
    t = np.arange(n)
    cols = []
    for c in range(channels):
        x = 0.55 * top * np.sin(2 * np.pi * t * (220.0 + 35 * c) / 16000.0) if tone else np.zeros(n)
        x = x + rng.integers(-max(1, top // 200), max(2, top // 200), n)
        cols.append(np.clip(np.rint(x), -top - 1, top).astype(np.int64))
    return np.stack(cols, axis=1)


def _as_float(pcm, bits):
    return (pcm.astype(np.float64) / float(1 << (bits - 1))).astype(np.float32)


def _decode(data):
    x, rate = decode_bytes(data)
    return (x[:, None] if x.ndim == 1 else x), rate


# ------------------------------------------------------------------------------------------------------- FLAC
def test_flac_every_subframe_type_and_stereo_mode(native_lib):
    pcm = _signal(4096 + 300, 2, 16, seed=1)
    pcm[:, 1] = (pcm[:, 0] * 0.9).astype(np.int64) + np.random.default_rng(2).integers(-20, 20, len(pcm))
    frames = [
        {"block": 1024, "mode": "independent",
         "subframes": [{"type": "verbatim"}, {"type": "fixed", "order": 2, "partition_order": 3}]},
        {"block": 1024, "mode": "left_side",
         "subframes": [{"type": "fixed", "order": 4, "method": 1, "partition_order": 2, "escape_partitions": (1,)},
                       {"type": "fixed", "order": 1}]},
        {"block": 1024, "mode": "side_right",
         "subframes": [{"type": "lpc", "order": 3, "coefs": [24, -20, 7], "precision": 7, "shift": 3, "partition_order": 4},
                       {"type": "fixed", "order": 0}]},
        {"block": 1024, "mode": "mid_side",
         "subframes": [{"type": "lpc", "order": 8, "coefs": [900, -400, 120, -60, 30, -10, 4, -1], "precision": 12, "shift": 9},
                       {"type": "fixed", "order": 3, "partition_order": 1}]},
        {"block": 300, "mode": "mid_side", "subframes": [{"type": "fixed", "order": 2}, {"type": "verbatim"}]},
    ]
    x, rate = _decode(encode_stream(pcm, bits=16, rate=16000, frames=frames))
    assert rate == 16000 and np.array_equal(x, _as_float(pcm, 16))


@pytest.mark.parametrize("bits", [8, 12, 16, 20, 24, 32])
def test_flac_sample_sizes_including_33_bit_side_channels(native_lib, bits):
    pcm = _signal(1152 + 576, 2, bits, seed=bits)
    if bits == 32:   # extremes: the side channel needs all 33 bits
        pcm[5] = [(1 << 31) - 1, -(1 << 31)]
        pcm[6] = [-(1 << 31), (1 << 31) - 1]
    frames = [{"block": 1152, "mode": "mid_side", "subframes": [{"type": "fixed", "order": 1, "method": 1}, {"type": "verbatim"}]},
              {"block": 576, "mode": "left_side", "subframes": [{"type": "verbatim"}, {"type": "fixed", "order": 0, "method": 1}]}]
    x, _ = _decode(encode_stream(pcm, bits=bits, rate=44100, frames=frames))
    assert np.array_equal(x, _as_float(pcm, bits))


def test_flac_constant_wasted_bits_and_many_channels(native_lib):
    n = 512
    pcm = _signal(n, 8, 16, seed=3)
    pcm[:, 0] = 1234                      # constant subframe
    pcm[:, 1] = (pcm[:, 1] >> 5) << 5     # five wasted bits
    pcm[:, 2] = 0                         # digital silence: constant 0
    pcm[:, 3] = (pcm[:, 3] >> 1) << 1     # one wasted bit, through an LPC subframe
    subs = [{"type": "constant"}, {"type": "fixed", "order": 2, "wasted": 5}, {"type": "constant"},
            {"type": "lpc", "order": 2, "coefs": [15, -7], "precision": 5, "shift": 3, "wasted": 1},
            {"type": "verbatim"}, {"type": "fixed", "order": 3}, {"type": "fixed", "order": 4, "partition_order": 5},
            {"type": "lpc", "order": 32, "coefs": [3] + [0] * 30 + [-1], "precision": 3, "shift": 2}]
    for channels in (1, 3, 5, 8):
        data = encode_stream(pcm[:, :channels], bits=16, rate=22050, frames=[{"block": n, "subframes": subs[:channels]}])
        x, rate = _decode(data)
        assert rate == 22050 and np.array_equal(x, _as_float(pcm[:, :channels], 16)), channels


def test_flac_block_size_and_sample_rate_codings(native_lib):
    pcm = _signal(200 + 2000 + 1 + 4608 + 37, 1, 16, seed=4)
    frames = [{"block": 200}, {"block": 2000, "subframes": [{"type": "fixed", "order": 2, "partition_order": 4}]},
              {"block": 1}, {"block": 4608, "explicit_block": True, "subframes": [{"type": "fixed", "order": 1, "partition_order": 8}]},
              {"block": 37, "subframes": [{"type": "fixed", "order": 4}]}]
    for rate, coding in ((16000, "table"), (16000, "streaminfo"), (16000, "khz"), (11025, "hz"), (352800, "hz"), (96000, "table")):
        x, got = _decode(encode_stream(pcm, bits=16, rate=rate, frames=[dict(f, rate_coding=coding) for f in frames], variable=True))
        assert got == rate and np.array_equal(x, _as_float(pcm, 16)), (rate, coding)
    # sample numbers beyond 2^31 in the coded number of a variable-block-size stream (7-byte form is exercised by the
    # writer's own round trip of frame numbers; here: a long fixed-block stream's large frame numbers)
    frame = encode_frame(pcm[:192], bits=16, rate=16000, number=(1 << 35) + 5, variable=True)
    head = stream_header(rate=16000, channels=1, bits=16, total=192, min_block=192, max_block=192, md5=pcm_md5(pcm[:192], 16))
    x, _ = _decode(head + frame)
    assert np.array_equal(x, _as_float(pcm[:192], 16))


def test_flac_tags_metadata_unknown_length_and_rice_extremes(native_lib):
    pcm = _signal(1024, 1, 24, seed=5, tone=False)
    pcm[100:130, 0] = np.random.default_rng(6).integers(-(1 << 22), 1 << 22, 30)   # a burst: huge Rice parameters next to tiny ones
    frames = [{"block": 1024, "subframes": [{"type": "fixed", "order": 0, "method": 1, "partition_order": 5, "escape_partitions": (0, 31)}]}]
    id3 = b"ID3\x04\x00\x00" + bytes([0, 0, 0, 17]) + b"x" * 17
    padding = (1, bytes(40))
    comment = (4, struct.pack("<I", 4) + b"test" + struct.pack("<I", 0))
    data = encode_stream(pcm, bits=24, rate=48000, frames=frames, id3=id3, extra_blocks=(padding, comment))
    x, rate = _decode(data)
    assert rate == 48000 and np.array_equal(x, _as_float(pcm, 24))
    # STREAMINFO without sample count and without MD5, an ID3v1-style trailer after the last frame
    data = encode_stream(pcm, bits=24, rate=48000, frames=frames, with_total=False, with_md5=False, trailer=b"TAG" + bytes(125))
    x, _ = _decode(data)
    assert np.array_equal(x, _as_float(pcm, 24))
    # an all-zero partition escaped with width 0
    silent = np.zeros((256, 1), dtype=np.int64)
    data = encode_stream(silent, bits=16, rate=8000, frames=[{"block": 256, "subframes": [{"type": "fixed", "order": 0, "partition_order": 2, "escape_partitions": (0, 1, 2, 3)}]}])
    x, _ = _decode(data)
    assert x.shape == (256, 1) and not x.any()


def test_flac_damage_is_an_error_never_a_partial_result(native_lib):
    pcm = _signal(2048, 2, 16, seed=7)
    frames = [{"block": 1024, "mode": "mid_side", "subframes": [{"type": "fixed", "order": 2}, {"type": "fixed", "order": 2}]}] * 2
    good = encode_stream(pcm, bits=16, rate=16000, frames=frames)
    decode_bytes(good)
    body_at = good.index(b"\xff\xf8")
    cases = {
        "flipped bit in a subframe": good[: body_at + 40] + bytes([good[body_at + 40] ^ 0x10]) + good[body_at + 41 :],
        "flipped bit in a frame header": good[: body_at + 3] + bytes([good[body_at + 3] ^ 0x01]) + good[body_at + 4 :],
        "truncated": good[: len(good) - 50],
        "second frame missing": good[: body_at + (len(good) - body_at) // 2 - 200],
    }
    for name, data in cases.items():
        with pytest.raises(AudioDecodeError):
            decode_bytes(data)
    wrong_md5 = bytearray(good)
    md5_at = good.index(b"fLaC") + 4 + 4 + 18
    wrong_md5[md5_at] ^= 0xFF
    with pytest.raises(AudioDecodeError, match="MD5"):
        decode_bytes(bytes(wrong_md5))
    with pytest.raises(AudioDecodeError, match="STREAMINFO|metadata"):
        decode_bytes(b"fLaC" + bytes([0x80, 0, 0, 10]) + bytes(10))
    # the C ABI itself: bad arguments are refused, not dereferenced
    from ctypes import byref, c_int64

    assert native_lib.serhip_flac_probe(None, 10, None, None, None, None) == -1
    assert native_lib.serhip_flac_decode(good, len(good), None, 0, 1, None) == -1
    done = c_int64(-1)
    assert native_lib.serhip_flac_decode(good, len(good), None, 0, 1, byref(done)) == 0 and done.value == 2048
    small = np.zeros((100, 2), dtype=np.int32)
    assert native_lib.serhip_flac_decode(good, len(good), small.ctypes.data, 100, 1, byref(done)) == -1
    assert b"longer than the output buffer" in native_lib.serhip_last_error()


# ------------------------------------------------------------------------------- AIFF, AU, G.711 (stdlib-written files)
class _KeptBytesIO(io.BytesIO):
    """The stdlib writers close the file object they were given; the bytes are wanted afterwards."""

    def close(self):
        pass


def _stdlib_aiff(pcm, bits, rate, *, aifc_ulaw=False):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        import aifc

    buf = _KeptBytesIO()
    w = aifc.open(buf, "wb")
    w.setnchannels(pcm.shape[1])
    w.setsampwidth(bits // 8)
    w.setframerate(rate)
    if aifc_ulaw:
        w.setcomptype(b"ULAW", b"mu-law")
    width = bits // 8
    # uncompressed frames are written as given (AIFF is big-endian); the module's mu-law compander takes native order
    w.writeframes(b"".join(int(v).to_bytes(width, "little" if aifc_ulaw else "big", signed=True) for v in pcm.reshape(-1)))
    w.close()
    return buf.getvalue()


@pytest.mark.parametrize("bits", [8, 16, 24, 32])
def test_aiff_written_by_the_standard_library(bits):
    pcm = _signal(777, 2, bits, seed=bits + 10)
    x, rate = _decode(_stdlib_aiff(pcm, bits, 22050))
    assert rate == 22050 and np.array_equal(x, _as_float(pcm, bits))
    x, rate = _decode(_stdlib_aiff(pcm[:, :1], bits, 8000))
    assert rate == 8000 and np.array_equal(x, _as_float(pcm[:, :1], bits))


def test_aiffc_variants():
    import audioop

    pcm = _signal(500, 2, 16, seed=21)
    # mu-law AIFF-C written by the standard library: its own compander, expanded here
    x, rate = _decode(_stdlib_aiff(pcm, 16, 16000, aifc_ulaw=True))
    linear = pcm.astype("<i2").tobytes()
    want = np.frombuffer(audioop.ulaw2lin(audioop.lin2ulaw(linear, 2), 2), dtype="<i2").reshape(-1, 2)
    assert rate == 16000 and np.array_equal(x, _as_float(want.astype(np.int64), 16))

    def aifc_file(kind, payload, bits, channels=2, n_frames=500):
        exp = 16383 + 13                                     # 16000 = 0.9765625 * 2^14 -> mantissa 16000 << 50
        comm = struct.pack(">hIh", channels, n_frames, bits) + struct.pack(">HQ", exp, 16000 << 50) + kind + b"\x00\x00"
        ssnd = struct.pack(">II", 0, 0) + payload
        body = b"AIFC" + b"FVER" + struct.pack(">II", 4, 0xA2805140) + b"COMM" + struct.pack(">I", len(comm)) + comm
        body += b"SSND" + struct.pack(">I", len(ssnd)) + ssnd + (b"\0" if len(ssnd) & 1 else b"")
        return b"FORM" + struct.pack(">I", len(body)) + body

    x, rate = _decode(aifc_file(b"sowt", pcm.astype("<i2").tobytes(), 16))          # little-endian PCM
    assert rate == 16000 and np.array_equal(x, _as_float(pcm, 16))
    f = (pcm / 40000.0).astype(">f4")
    x, _ = _decode(aifc_file(b"fl32", f.tobytes(), 32))
    assert np.array_equal(x, f.astype(np.float32))
    x, _ = _decode(aifc_file(b"fl64", f.astype(">f8").tobytes(), 64))
    assert np.array_equal(x, f.astype(np.float32))
    x, _ = _decode(aifc_file(b"alaw", audioop.lin2alaw(linear, 2), 8))
    want = np.frombuffer(audioop.alaw2lin(audioop.lin2alaw(linear, 2), 2), dtype="<i2").reshape(-1, 2)
    assert np.array_equal(x, _as_float(want.astype(np.int64), 16))
    with pytest.raises(AudioDecodeError, match="compression"):
        decode_bytes(aifc_file(b"ima4", bytes(100), 16))


def test_au_written_by_the_standard_library():
    import audioop

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        import sunau

    for bits in (8, 16, 24, 32):
        pcm = _signal(333, 2, bits, seed=bits + 30)
        buf = _KeptBytesIO()
        w = sunau.open(buf, "wb")
        w.setnchannels(2)
        w.setsampwidth(bits // 8)
        w.setframerate(11025)
        w.setcomptype("NONE", "not compressed")     # the module's default is mu-law
        w.writeframes(b"".join(int(v).to_bytes(bits // 8, "big", signed=True) for v in pcm.reshape(-1)))
        w.close()
        x, rate = _decode(buf.getvalue())
        assert rate == 11025 and np.array_equal(x, _as_float(pcm, bits)), bits
    pcm = _signal(400, 1, 16, seed=40)
    linear = pcm.astype("<i2").tobytes()
    buf = _KeptBytesIO()
    w = sunau.open(buf, "wb")
    w.setnchannels(1)
    w.setsampwidth(2)
    w.setframerate(8000)
    w.setcomptype("ULAW", "mu-law")
    w.writeframes(linear)
    w.close()
    x, rate = _decode(buf.getvalue())
    want = np.frombuffer(audioop.ulaw2lin(audioop.lin2ulaw(linear, 2), 2), dtype="<i2")[:, None]
    assert rate == 8000 and np.array_equal(x, _as_float(want.astype(np.int64), 16))
    # A-law (27), float32 (6) and float64 (7) by hand; unknown data size (0xFFFFFFFF) means "to the end of the file"
    def au(encoding, payload, size=None):
        return b".snd" + struct.pack(">IIIII", 28, len(payload) if size is None else size, encoding, 8000, 1) + b"note" + payload
    x, _ = _decode(au(27, audioop.lin2alaw(linear, 2)))
    want = np.frombuffer(audioop.alaw2lin(audioop.lin2alaw(linear, 2), 2), dtype="<i2")[:, None]
    assert np.array_equal(x, _as_float(want.astype(np.int64), 16))
    f = (pcm[:, 0] / 33000.0)
    assert np.array_equal(_decode(au(6, f.astype(">f4").tobytes(), 0xFFFFFFFF))[0][:, 0], f.astype(">f4").astype(np.float32))
    assert np.array_equal(_decode(au(7, f.astype(">f8").tobytes()))[0][:, 0], f.astype(np.float32))
    with pytest.raises(AudioDecodeError, match="encoding"):
        decode_bytes(au(23, bytes(64)))     # G.721 ADPCM


def test_wave_g711_and_rf64():
    import audioop

    pcm = _signal(600, 2, 16, seed=50)
    linear = pcm.astype("<i2").tobytes()

    def wav(code, bits, payload, *, rf64=False):
        fmt = struct.pack("<HHIIHH", code, 2, 8000, 8000 * 2 * bits // 8, 2 * bits // 8, bits)
        chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
        if rf64:
            ds64 = struct.pack("<QQQI", 0, len(payload), len(payload) // (2 * bits // 8), 0)
            chunks = b"ds64" + struct.pack("<I", len(ds64)) + ds64 + chunks
            chunks += b"data" + struct.pack("<I", 0xFFFFFFFF) + payload
            return b"RF64" + struct.pack("<I", 0xFFFFFFFF) + b"WAVE" + chunks
        chunks += b"data" + struct.pack("<I", len(payload)) + payload
        return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks

    for code, to_law, from_law in ((6, audioop.lin2alaw, audioop.alaw2lin), (7, audioop.lin2ulaw, audioop.ulaw2lin)):
        x, rate = _decode(wav(code, 8, to_law(linear, 2)))
        want = np.frombuffer(from_law(to_law(linear, 2), 2), dtype="<i2").reshape(-1, 2)
        assert rate == 8000 and np.array_equal(x, _as_float(want.astype(np.int64), 16))
    x, _ = _decode(wav(1, 16, linear, rf64=True))
    assert np.array_equal(x, _as_float(pcm, 16))


# ------------------------------------------------------------------------------------------- the file-level reader
def test_read_audio_file_is_container_agnostic(tmp_path, native_lib):
    """The same PCM as WAV, FLAC, AIFF and AU gives the same prepared samples (mix-down, scrub, peak normalisation of
    audio_utils.py:28-60) -- also for a segment read -- whatever the file is called."""
    import wave

    pcm = _signal(16000, 2, 16, seed=60)
    with wave.open(str(tmp_path / "a.wav"), "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(pcm.astype("<i2").tobytes())
    (tmp_path / "a.flac").write_bytes(encode_stream(pcm, bits=16, rate=16000, frames=[
        {"block": 4096, "mode": "mid_side", "subframes": [{"type": "fixed", "order": 2, "partition_order": 4}] * 2}] * 3
        + [{"block": 16000 - 3 * 4096, "mode": "left_side", "subframes": [{"type": "fixed", "order": 1}] * 2}]))
    (tmp_path / "a.aiff").write_bytes(_stdlib_aiff(pcm, 16, 16000))
    (tmp_path / "misnamed.wav").write_bytes((tmp_path / "a.flac").read_bytes())     # magic decides, not the extension
    want, rate = read_audio_file(str(tmp_path / "a.wav"))
    assert rate == 16000 and want.ndim == 1 and float(np.max(np.abs(want))) == 1.0
    for name in ("a.flac", "a.aiff", "misnamed.wav"):
        got, got_rate = read_audio_file(str(tmp_path / name))
        assert got_rate == 16000 and np.array_equal(got, want), name
        seg, _ = read_audio_file(str(tmp_path / name), start_seconds=0.25, duration_seconds=0.5)
        seg_want, _ = read_audio_file(str(tmp_path / "a.wav"), start_seconds=0.25, duration_seconds=0.5)
        assert seg.shape == (8000,) and np.array_equal(seg, seg_want), name


def test_lossy_containers_are_named_and_refused_without_retries(tmp_path, monkeypatch):
    import time

    monkeypatch.setattr(time, "sleep", lambda s: (_ for _ in ()).throw(AssertionError("a deterministic verdict must not sleep")))
    for name, head, word in (("a.ogg", b"OggS\x00\x02" + bytes(40), "Ogg"), ("a.mp3", b"\xff\xfb\x90\x00" + bytes(40), "MP3"),
                             ("b.mp3", b"ID3\x03\x00\x00\x00\x00\x00\x05" + bytes(5) + b"\xff\xfb\x90\x00" + bytes(40), "MP3"),
                             ("c.bin", bytes(64), "not a RIFF/WAVE, FLAC, AIFF or AU file")):
        p = tmp_path / name
        p.write_bytes(head)
        with pytest.raises(AudioDecodeError, match=word):
            read_audio_file(str(p))
