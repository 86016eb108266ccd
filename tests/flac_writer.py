"""Test-only FLAC ENCODER written from the format specification (RFC 9639), independent of the decoders under
test (espnet_amd/fileio/sound_scp.py:read_flac and csrc/host_io.cpp): it lets the tests produce every subframe
type, residual coding variant and channel assignment on demand, which no tool in this image can (there is no
libFLAC / libsndfile / ffmpeg here).  Not used by the product."""
import hashlib
import struct
from typing import List, Optional, Sequence

import numpy as np

FIXED = ((), (1,), (2, -1), (3, -3, 1), (4, -6, 4, -1))


class BitWriter:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, value: int, bits: int):
        if bits:
            self.v = (self.v << bits) | (int(value) & ((1 << bits) - 1))
            self.n += bits

    def unary(self, q: int):  # q zeros, then a one
        self.put(1, q + 1)

    def align(self):
        if self.n % 8:
            self.put(0, 8 - self.n % 8)

    def bytes(self) -> bytes:
        assert self.n % 8 == 0
        return self.v.to_bytes(self.n // 8, "big") if self.n else b""


def crc(data: bytes, poly: int, width: int) -> int:
    c, top, mask = 0, 1 << (width - 1), (1 << width) - 1
    for byte in data:
        c ^= byte << (width - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


def utf8_number(v: int) -> bytes:
    if v < 0x80:
        return bytes([v])
    n = 2
    while v >= 1 << (5 * n + 1):  # payload bits of an n-byte code: 5n + 1
        n += 1
    out = [((0xFF << (8 - n)) & 0xFF) | (v >> (6 * (n - 1)))]
    for k in range(n - 2, -1, -1):
        out.append(0x80 | ((v >> (6 * k)) & 0x3F))
    return bytes(out)


def write_residual(bw: BitWriter, res: Sequence[int], bs: int, order: int, po: int, method: int, escape: bool):
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    bw.put(method, 2)
    bw.put(po, 4)
    assert bs % (1 << po) == 0 and (bs >> po) >= order
    i = 0
    for pt in range(1 << po):
        cnt = (bs >> po) - (order if pt == 0 else 0)
        part = [int(r) for r in res[i : i + cnt]]
        i += cnt
        if escape and pt % 2 == 0:
            nb = max([1] + [(r if r >= 0 else ~r).bit_length() + 1 for r in part])
            bw.put(esc, pbits)
            bw.put(nb, 5)
            for r in part:
                bw.put(r, nb)
            continue
        mean = sum(abs(r) for r in part) / max(1, len(part))
        param = min(esc - 1, max(0, int(mean).bit_length()))
        bw.put(param, pbits)
        for r in part:
            u = 2 * r if r >= 0 else -2 * r - 1  # zig-zag folding
            bw.unary(u >> param)
            bw.put(u & ((1 << param) - 1), param)
    assert i == len(res)


def write_subframe(bw: BitWriter, s: Sequence[int], bps: int, kind: str, order: int = 0,
                   coefs: Optional[Sequence[int]] = None, prec: int = 0, shift: int = 0, po: int = 0,
                   method: int = 0, escape: bool = False, use_wasted: bool = True):
    s = [int(v) for v in s]
    bs = len(s)
    wasted = 0
    if use_wasted and any(s):
        while all(v % (1 << (wasted + 1)) == 0 for v in s) and wasted + 1 < bps:
            wasted += 1
    if wasted:
        s = [v >> wasted for v in s]
        bps -= wasted
    code = {"constant": 0, "verbatim": 1}.get(kind)
    if kind == "fixed":
        code = 8 + order
    elif kind == "lpc":
        code = 32 + order - 1
    bw.put(0, 1)
    bw.put(code, 6)
    if wasted:
        bw.put(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    if kind == "constant":
        assert all(v == s[0] for v in s)
        bw.put(s[0], bps)
    elif kind == "verbatim":
        for v in s:
            bw.put(v, bps)
    else:
        c = list(FIXED[order]) if kind == "fixed" else [int(v) for v in coefs]
        assert len(c) == order <= bs
        for v in s[:order]:
            bw.put(v, bps)
        if kind == "lpc":
            bw.put(prec - 1, 4)
            bw.put(shift, 5)
            for v in c:
                bw.put(v, prec)
        res = []
        for k in range(order, bs):
            acc = sum(c[j] * s[k - 1 - j] for j in range(order))
            res.append(s[k] - ((acc >> shift) if kind == "lpc" else acc))
        write_residual(bw, res, bs, order, po, method, escape)


def lpc_coefficients(x: np.ndarray, order: int, prec: int):
    """Quantised forward-prediction coefficients (autocorrelation method); any integers are VALID for the format,
    these just make the residual small like a real encoder's."""
    x = np.asarray(x, dtype=np.float64)
    r = np.array([np.dot(x[: len(x) - k], x[k:]) for k in range(order + 1)])
    r[0] = r[0] * (1 + 1e-9) + 1e-6
    a = np.zeros(order)
    err = r[0]
    for i in range(order):  # Levinson-Durbin
        k = (r[i + 1] - np.dot(a[:i], r[i:0:-1])) / err
        a[:i] = a[:i] - k * a[:i][::-1]
        a[i] = k
        err *= max(1e-12, 1 - k * k)
    amax = max(1e-9, float(np.abs(a).max()))
    shift = max(0, min(15, prec - 1 - int(np.ceil(np.log2(amax))) - 1))
    lim = (1 << (prec - 1)) - 1
    q = np.clip(np.rint(a * (1 << shift)), -lim - 1, lim).astype(np.int64)
    return [int(v) for v in q], shift


BLOCK_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13,
               16384: 14, 32768: 15}
SIZE_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def encode_flac(channels: Sequence[np.ndarray], bits: int, rate: int, blocksize: int, plan=None,
                assignment: str = "independent", extra_blocks: Sequence[tuple] = (), total_in_header: bool = True,
                trailer: bytes = b"") -> bytes:
    """channels: C integer arrays of equal length.  plan(frame_index, channel_index, samples, bps) -> kwargs of
    write_subframe (default: verbatim).  assignment: independent | left_side | right_side | mid_side (2 ch)."""
    ch = [np.asarray(c, dtype=np.int64) for c in channels]
    n, nch = len(ch[0]), len(ch)
    frames = []
    for fi, a in enumerate(range(0, n, blocksize)):
        blk = [c[a : a + blocksize] for c in ch]
        bs = len(blk[0])
        if nch == 2 and assignment != "independent":
            left, right = blk
            side = left - right
            if assignment == "left_side":
                subs, bpss, code = [left, side], [bits, bits + 1], 8
            elif assignment == "right_side":
                subs, bpss, code = [side, right], [bits + 1, bits], 9
            else:
                subs, bpss, code = [(left + right) >> 1, side], [bits, bits + 1], 10
        else:
            subs, bpss, code = blk, [bits] * nch, nch - 1
        bw = BitWriter()
        bw.put(0x3FFE, 14)
        bw.put(0, 1)
        bw.put(0, 1)  # fixed block size stream: frame NUMBER is coded
        bcode = BLOCK_CODES.get(bs, 6 if bs <= 256 else 7)
        bw.put(bcode, 4)
        bw.put({44100: 9, 16000: 5, 8000: 4, 22050: 6, 48000: 10}.get(rate, 0), 4)
        bw.put(code, 4)
        bw.put(SIZE_CODES.get(bits, 0), 3)
        bw.put(0, 1)
        for byte in utf8_number(fi):
            bw.put(byte, 8)
        if bcode == 6:
            bw.put(bs - 1, 8)
        elif bcode == 7:
            bw.put(bs - 1, 16)
        bw.put(crc(bw.bytes(), 0x07, 8), 8)
        for ci, (s, bps) in enumerate(zip(subs, bpss)):
            kw = plan(fi, ci, s, bps) if plan is not None else dict(kind="verbatim")
            write_subframe(bw, s, bps, **kw)
        bw.align()
        body = bw.bytes()
        frames.append(body + struct.pack(">H", crc(body, 0x8005, 16)))
    width = (bits + 7) // 8
    inter = np.stack(ch, axis=1).reshape(-1)
    pcm = b"".join(int(v).to_bytes(width, "little", signed=True) for v in inter)
    si = BitWriter()
    si.put(blocksize, 16)
    si.put(blocksize, 16)
    si.put(min(map(len, frames)) if frames else 0, 24)
    si.put(max(map(len, frames)) if frames else 0, 24)
    si.put(rate, 20)
    si.put(nch - 1, 3)
    si.put(bits - 1, 5)
    si.put(n if total_in_header else 0, 36)
    out = b"fLaC"
    blocks = [(0, si.bytes() + hashlib.md5(pcm).digest())] + list(extra_blocks)
    for k, (kind, body) in enumerate(blocks):
        out += bytes([(0x80 if k == len(blocks) - 1 else 0) | kind]) + len(body).to_bytes(3, "big") + body
    return out + b"".join(frames) + trailer
