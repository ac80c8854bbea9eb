"""A small FLAC *encoder* for the decoder's tests (RFC 9639), written independently of the decoder in
``ser_amd/csrc/flac_decoder.hpp``: plain Python integers, ``hashlib.md5`` for the STREAMINFO signature, bitwise
CRCs.  It does not try to compress well -- it lets a test choose, frame by frame, every syntactic feature the
decoder must handle (subframe type, predictor order, Rice code book, partition order, escape partitions, wasted
bits, stereo mode, block size and sample rate coding).

No FLAC produced by a third-party encoder exists in this image (no libFLAC, no soundfile, no sample files), so the
decoder is pinned by this encoder, by the format's own checks (CRC-8, CRC-16, MD5 against hashlib) and by nothing
else: "parity unpinned" against libsndfile, stated where the feature is documented."""

from __future__ import annotations

import hashlib
import struct

import numpy as np

BLOCK_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13,
               16384: 14, 32768: 15}
RATE_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10,
              96000: 11}
BITS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


class BitWriter:
    def __init__(self) -> None:
        self.value = 0
        self.count = 0

    def put(self, v: int, n: int) -> None:
        assert 0 <= v < (1 << n) or n == 0, (v, n)
        self.value = (self.value << n) | v
        self.count += n

    def put_signed(self, v: int, n: int) -> None:
        assert -(1 << (n - 1)) <= v < (1 << (n - 1)), (v, n)
        self.put(v & ((1 << n) - 1), n)

    def put_unary(self, zeros: int) -> None:
        self.put(1, zeros + 1)

    def pad_to_byte(self) -> None:
        if self.count % 8:
            self.put(0, 8 - self.count % 8)

    def to_bytes(self) -> bytes:
        assert self.count % 8 == 0
        return self.value.to_bytes(self.count // 8, "big")


def crc8(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def coded_number(v: int) -> bytes:
    """The UTF-8-like coding of frame / sample numbers (up to 36 bits)."""
    if v < 0x80:
        return bytes([v])
    for n_bytes, limit in ((2, 1 << 11), (3, 1 << 16), (4, 1 << 21), (5, 1 << 26), (6, 1 << 31), (7, 1 << 36)):
        if v < limit:
            lead = (0xFF << (8 - n_bytes)) & 0xFF
            rest = [0x80 | ((v >> (6 * i)) & 0x3F) for i in range(n_bytes - 2, -1, -1)]
            return bytes([lead | (v >> (6 * (n_bytes - 1)))] + rest)
    raise ValueError(v)


def _residual(w: BitWriter, res: list[int], block: int, order: int, *, method: int, partition_order: int,
              escape_partitions=()) -> None:
    w.put(method, 2)
    w.put(partition_order, 4)
    parts = 1 << partition_order
    assert block % parts == 0 and block >> partition_order >= order
    param_bits, escape = (4, 15) if method == 0 else (5, 31)
    at = 0
    for part in range(parts):
        count = (block >> partition_order) - (order if part == 0 else 0)
        chunk = res[at : at + count]
        at += count
        if part in escape_partitions:
            width = max([0] + [(v if v >= 0 else ~v).bit_length() + 1 for v in chunk])
            if all(v == 0 for v in chunk):
                width = 0
            w.put(escape, param_bits)
            w.put(width, 5)
            for v in chunk:
                if width:
                    w.put_signed(v, width)
            continue
        zig = [2 * v if v >= 0 else -2 * v - 1 for v in chunk]
        mean = (sum(zig) // max(1, len(zig))) if zig else 0
        k = min(max(0, mean.bit_length() - 1), escape - 1)
        w.put(k, param_bits)
        for u in zig:
            w.put_unary(u >> k)
            if k:
                w.put(u & ((1 << k) - 1), k)
    assert at == len(res)


def _subframe(w: BitWriter, samples: list[int], bps: int, spec: dict) -> None:
    """spec: {"type": "constant" | "verbatim" | "fixed" | "lpc", "order", "coefs", "precision", "shift", "method",
    "partition_order", "escape_partitions", "wasted"}"""
    kind = spec.get("type", "verbatim")
    wasted = int(spec.get("wasted", 0))
    if wasted:
        assert all(v % (1 << wasted) == 0 for v in samples)
        samples = [v >> wasted for v in samples]
        bps -= wasted
    block = len(samples)
    w.put(0, 1)
    if kind == "constant":
        assert len(set(samples)) == 1
        code = 0
    elif kind == "verbatim":
        code = 1
    elif kind == "fixed":
        code = 8 + spec["order"]
    else:
        code = 31 + spec["order"]
    w.put(code, 6)
    if wasted:
        w.put(1, 1)
        w.put_unary(wasted - 1)
    else:
        w.put(0, 1)
    if kind == "constant":
        w.put_signed(samples[0], bps)
        return
    if kind == "verbatim":
        for v in samples:
            w.put_signed(v, bps)
        return
    order = spec["order"]
    for v in samples[:order]:
        w.put_signed(v, bps)
    if kind == "fixed":
        taps = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order]
        res = [samples[i] - sum(t * samples[i - 1 - j] for j, t in enumerate(taps)) for i in range(order, block)]
    else:
        coefs, precision, shift = spec["coefs"], spec["precision"], spec["shift"]
        assert len(coefs) == order
        w.put(precision - 1, 4)
        w.put_signed(shift, 5)
        for c in coefs:
            w.put_signed(c, precision)
        res = [samples[i] - (sum(c * samples[i - 1 - j] for j, c in enumerate(coefs)) >> shift) for i in range(order, block)]
    _residual(w, res, block, order, method=spec.get("method", 0), partition_order=spec.get("partition_order", 0),
              escape_partitions=spec.get("escape_partitions", ()))


def encode_frame(pcm: np.ndarray, *, bits: int, rate: int, number: int, variable: bool, mode: str = "independent",
                 subframes=None, explicit_block: bool = False, rate_coding: str = "table", bits_from_streaminfo: bool = False) -> bytes:
    """One frame for ``pcm`` ``[block, channels]`` (Python ints or an integer array)."""
    block, channels = pcm.shape
    cols = [[int(v) for v in pcm[:, c]] for c in range(channels)]
    widths = [bits] * channels
    if mode == "independent":
        assign = channels - 1
    else:
        assert channels == 2
        left, right = cols
        side = [a - b for a, b in zip(left, right)]
        if mode == "left_side":
            assign, cols, widths = 8, [left, side], [bits, bits + 1]
        elif mode == "side_right":
            assign, cols, widths = 9, [side, right], [bits + 1, bits]
        else:
            assign, cols, widths = 10, [[(a + b) >> 1 for a, b in zip(left, right)], side], [bits, bits + 1]
    w = BitWriter()
    w.put(0xFFF8 | (1 if variable else 0), 16)
    if block in BLOCK_CODES and not explicit_block:
        bs_code, bs_extra = BLOCK_CODES[block], None
    elif block <= 256:
        bs_code, bs_extra = 6, (block - 1, 8)
    else:
        bs_code, bs_extra = 7, (block - 1, 16)
    if rate_coding == "streaminfo":
        sr_code, sr_extra = 0, None
    elif rate_coding == "table" and rate in RATE_CODES:
        sr_code, sr_extra = RATE_CODES[rate], None
    elif rate % 1000 == 0 and rate // 1000 < 256 and rate_coding != "hz":
        sr_code, sr_extra = 12, (rate // 1000, 8)
    elif rate < 65536:
        sr_code, sr_extra = 13, (rate, 16)
    else:
        assert rate % 10 == 0
        sr_code, sr_extra = 14, (rate // 10, 16)
    w.put(bs_code, 4)
    w.put(sr_code, 4)
    w.put(assign, 4)
    w.put(0 if bits_from_streaminfo or bits not in BITS_CODES else BITS_CODES[bits], 3)
    w.put(0, 1)
    for b in coded_number(number):
        w.put(b, 8)
    if bs_extra:
        w.put(*bs_extra)
    if sr_extra:
        w.put(*sr_extra)
    w.put(crc8(w.to_bytes()), 8)
    subframes = subframes or [{"type": "verbatim"}] * channels
    for col, width, spec in zip(cols, widths, subframes):
        _subframe(w, col, width, spec)
    w.pad_to_byte()
    body = w.to_bytes()
    return body + struct.pack(">H", crc16(body))


def stream_header(*, rate: int, channels: int, bits: int, total: int, min_block: int, max_block: int, md5: bytes,
                  extra_blocks=(), id3: bytes = b"") -> bytes:
    info = BitWriter()
    info.put(min_block, 16)
    info.put(max_block, 16)
    info.put(0, 24)
    info.put(0, 24)
    info.put(rate, 20)
    info.put(channels - 1, 3)
    info.put(bits - 1, 5)
    info.put(total, 36)
    body = info.to_bytes() + md5
    out = id3 + b"fLaC"
    blocks = [(0, body)] + list(extra_blocks)
    for i, (kind, data) in enumerate(blocks):
        last = 0x80 if i == len(blocks) - 1 else 0
        out += bytes([last | kind]) + len(data).to_bytes(3, "big") + data
    return out


def pcm_md5(pcm: np.ndarray, bits: int) -> bytes:
    width = (bits + 7) // 8
    raw = b"".join(int(v).to_bytes(width, "little", signed=True) for v in pcm.reshape(-1))
    return hashlib.md5(raw).digest()


def encode_stream(pcm: np.ndarray, *, bits: int, rate: int, frames, variable: bool = False, with_md5: bool = True,
                  with_total: bool = True, extra_blocks=(), id3: bytes = b"", trailer: bytes = b"") -> bytes:
    """``frames``: list of dicts ``{"block": n, "mode": ..., "subframes": [...], ...}`` consumed in order over ``pcm``
    ``[n, channels]``; the keys other than ``block`` go to :func:`encode_frame`."""
    pcm = np.asarray(pcm)
    if pcm.ndim == 1:
        pcm = pcm[:, None]
    at, body, sizes = 0, b"", []
    for index, spec in enumerate(frames):
        spec = dict(spec)
        block = spec.pop("block")
        chunk = pcm[at : at + block]
        assert len(chunk) == block, "frames must cover the signal exactly"
        body += encode_frame(chunk, bits=bits, rate=rate, number=at if variable else index, variable=variable, **spec)
        sizes.append(block)
        at += block
    assert at == len(pcm)
    head = stream_header(rate=rate, channels=pcm.shape[1], bits=bits, total=len(pcm) if with_total else 0,
                         min_block=min(sizes) if variable else sizes[0], max_block=max(sizes),
                         md5=pcm_md5(pcm, bits) if with_md5 else bytes(16), extra_blocks=extra_blocks, id3=id3)
    return head + body + trailer
