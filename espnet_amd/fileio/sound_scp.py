"""Audio readers for the decode CLI (espnet2/fileio/sound_scp.py:13-155, npy_scp.py, and the
`sound` / `npy` entries of espnet2/train/iterable_dataset.py:44-67).

The reference reads through `soundfile` (libsndfile), which is not in this image.  The two containers the
recipes produce (`format_wav_scp.sh --audio-format wav|flac`; flac is the default, egs2/TEMPLATE/asr1/asr.sh:56)
are decoded here directly, with the sample convention libsndfile uses for float reads:
  * RIFF/WAVE: integer PCM of width w bytes -> value / 2**(8w-1) (8-bit is unsigned, offset 128); IEEE float
    taken as is;
  * FLAC (RFC 9639): `read_flac`, every subframe type and channel assignment, CRCs verified -> value / 2**(bits-1).
Anything else (ogg, sph, pipes `cmd |`) raises NotImplementedError naming the file.  `WavBatchReader` is the
batched fast path of the CLI on the C-ABI host reader (csrc/host_io.cpp), which decodes mono wav and flac natively.
"""
import collections.abc
import struct
import time
from pathlib import Path
from typing import Optional, Tuple, Union

import numpy as np

from espnet_amd.fileio.read_text import read_2columns_text

_WAVE_FORMAT_PCM, _WAVE_FORMAT_IEEE_FLOAT, _WAVE_FORMAT_EXTENSIBLE = 0x0001, 0x0003, 0xFFFE


class _Bits:
    """MSB-first bit reader over bytes (FLAC is a big-endian bit stream)."""

    def __init__(self, data: bytes, pos: int = 0):
        self.d, self.pos = data, pos * 8

    def bits(self, k: int) -> int:
        if k == 0:
            return 0
        a, b = self.pos >> 3, (self.pos + k + 7) >> 3
        if b > len(self.d):
            raise EOFError("FLAC stream ends inside a frame")
        v = int.from_bytes(self.d[a:b], "big") >> (b * 8 - self.pos - k)
        self.pos += k
        return v & ((1 << k) - 1)

    def sbits(self, k: int) -> int:
        v = self.bits(k)
        return v - (1 << k) if k and v >> (k - 1) else v

    def unary(self) -> int:
        q = 0
        while True:  # byte at a time: count leading zeros of what is left of the current byte
            i, off = self.pos >> 3, self.pos & 7
            if i >= len(self.d):
                raise EOFError("FLAC stream ends inside a Rice code")
            rest = (self.d[i] << off) & 0xFF
            if rest == 0:
                q += 8 - off
                self.pos += 8 - off
                continue
            lz = 8 - rest.bit_length()
            self.pos += lz + 1
            return q + lz


def _crc(data: bytes, poly: int, width: int) -> int:
    c, top, mask = 0, 1 << (width - 1), (1 << width) - 1
    for byte in data:
        c ^= byte << (width - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


_FIXED = ((), (1,), (2, -1), (3, -3, 1), (4, -6, 4, -1))


def _flac_subframe(br: _Bits, bs: int, bps: int) -> list:
    if br.bits(1):
        raise ValueError("subframe padding bit set")
    kind = br.bits(6)
    wasted = br.unary() + 1 if br.bits(1) else 0
    bps -= wasted
    if bps <= 0:
        raise ValueError("wasted bits >= sample size")
    if kind == 0:
        out = [br.sbits(bps)] * bs
    elif kind == 1:
        out = [br.sbits(bps) for _ in range(bs)]
    elif 8 <= kind <= 12 or kind >= 32:
        lpc = kind >= 32
        order = (kind & 31) + 1 if lpc else kind - 8
        if order > bs:
            raise ValueError("predictor order > block size")
        out = [br.sbits(bps) for _ in range(order)]
        shift = 0
        if lpc:
            prec = br.bits(4) + 1
            shift = br.sbits(5)
            if prec == 16 or shift < 0:
                raise ValueError("invalid LPC precision / shift")
            coef = [br.sbits(prec) for _ in range(order)]
        else:
            coef = list(_FIXED[order])
        method = br.bits(2)
        if method > 1:
            raise ValueError("reserved residual coding method")
        pbits, esc = (4, 15) if method == 0 else (5, 31)
        po = br.bits(4)
        if po and (bs % (1 << po) or (bs >> po) < order):
            raise ValueError("invalid partition order")
        res = []
        for pt in range(1 << po):
            cnt = (bs >> po) - (order if pt == 0 else 0)
            param = br.bits(pbits)
            if param == esc:
                nb = br.bits(5)
                res.extend(br.sbits(nb) for _ in range(cnt))
            else:
                for _ in range(cnt):
                    u = (br.unary() << param) | br.bits(param)
                    res.append((u >> 1) ^ -(u & 1))
        if len(res) != bs - order:
            raise ValueError("residual count mismatch")
        for r in res:  # s[k] = r + (sum_j coef[j] * s[k-1-j] >> shift)
            acc = 0
            for j in range(order):
                acc += coef[j] * out[-1 - j]
            out.append(r + (acc >> shift))
    else:
        raise ValueError(f"reserved subframe type {kind}")
    return [v << wasted for v in out] if wasted else out


_native_single = None  # WavBatchReader(1), created on first use; False when the library is not built


def _read_flac_native(path: str):
    """Mono stream of known length and <= 24 bits through the C-ABI decoder -> (float32 samples, rate), else None."""
    global _native_single
    if _native_single is None:
        try:
            _native_single = WavBatchReader(1)
        except Exception:  # library not built: the Python decoder below needs nothing
            _native_single = False
    if _native_single is False:
        return None
    probed = _native_single.probe([path])
    if probed is None or probed[1][0].bits > 24:
        return None
    try:
        out, _ = _native_single.load(probed, [0])
    except OSError:
        return None  # let the Python decoder name the problem
    return out[0].numpy(), int(probed[1][0].rate)


def read_flac(path: Union[Path, str], dtype="float64", always_2d: bool = False,
              native: Optional[bool] = None) -> Tuple[np.ndarray, int]:
    """FLAC decoder written from the format specification (RFC 9639), the Python counterpart of
    csrc/host_io.cpp (which decodes mono streams for the batched fast path): all subframe types, Rice / Rice2
    residuals with escaped partitions, wasted bits, every channel assignment, CRC-8 and CRC-16 verified.
    Samples -> value / 2**(bits-1), like libsndfile's float read of a FLAC file.
    native=None: mono streams go through the (200x faster, bit-identical) native decoder when the library is
    built; native=False forces the Python decoder (the tests compare the two)."""
    if native is not False:
        got = _read_flac_native(str(path))
        if got is not None:
            x, rate = got  # float32 is exact for <= 24-bit samples, so any requested dtype is exact too
            x = x.reshape(-1, 1) if always_2d else x
            return x.astype(dtype, copy=False), rate
        if native:
            raise RuntimeError(f"{path}: not decodable by the native FLAC reader")
    data = Path(path).read_bytes()
    if data[:4] != b"fLaC":
        raise NotImplementedError(f"{path}: not a FLAC stream")
    o, last, si = 4, False, None
    while not last:
        if o + 4 > len(data):
            raise RuntimeError(f"{path}: truncated FLAC metadata")
        last, kind = bool(data[o] & 0x80), data[o] & 0x7F
        n = int.from_bytes(data[o + 1 : o + 4], "big")
        if si is None:
            if kind != 0 or n < 34:
                raise RuntimeError(f"{path}: STREAMINFO must be the first metadata block")
            s = _Bits(data, o + 4)
            s.bits(16), s.bits(16), s.bits(24), s.bits(24)
            si = dict(rate=s.bits(20), channels=s.bits(3) + 1, bits=s.bits(5) + 1, total=s.bits(36))
        o += 4 + n
    nch, bits, total = si["channels"], si["bits"], si["total"]
    chans = [[] for _ in range(nch)]
    sizes = {1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}
    while o + 2 <= len(data) and (total == 0 or len(chans[0]) < total):
        if data[o] != 0xFF or (data[o + 1] & 0xFE) != 0xF8:
            if total == 0:
                break  # trailing tag after the last frame of a stream of unknown length
            raise RuntimeError(f"{path}: lost FLAC frame sync at byte {o}")
        br = _Bits(data, o)
        br.bits(16)
        bs_code, sr_code, ch_code, sz_code = br.bits(4), br.bits(4), br.bits(4), br.bits(3)
        if br.bits(1):
            raise RuntimeError(f"{path}: reserved frame header bit set")
        lead = br.bits(8)
        if lead & 0x80:
            extra = 0
            while lead & (0x40 >> extra):
                extra += 1
            if not 1 <= extra <= 6:
                raise RuntimeError(f"{path}: bad coded frame number")
            for _ in range(extra):
                if br.bits(8) & 0xC0 != 0x80:
                    raise RuntimeError(f"{path}: bad coded frame number")
        if bs_code == 0 or sr_code == 15 or sz_code == 3:
            raise RuntimeError(f"{path}: reserved frame header code")
        bs = (192 if bs_code == 1 else 576 << (bs_code - 2) if bs_code <= 5 else br.bits(8) + 1 if bs_code == 6
              else br.bits(16) + 1 if bs_code == 7 else 256 << (bs_code - 8))
        if sr_code == 12:
            br.bits(8)
        elif sr_code in (13, 14):
            br.bits(16)
        bps = bits if sz_code == 0 else sizes[sz_code]
        hdr = br.pos // 8 - o
        if _crc(data[o : o + hdr], 0x07, 8) != br.bits(8):
            raise RuntimeError(f"{path}: FLAC frame header CRC mismatch at byte {o}")
        n_sub = ch_code + 1 if ch_code < 8 else 2
        if ch_code > 10 or n_sub != nch or bps != bits:
            raise RuntimeError(f"{path}: frame layout differs from STREAMINFO")
        # the side channel of a decorrelated pair carries one more bit
        side = {8: 1, 9: 0, 10: 1}.get(ch_code)
        sub = [_flac_subframe(br, bs, bps + (1 if side == c else 0)) for c in range(n_sub)]
        if ch_code == 8:  # left, side
            sub[1] = [l - s for l, s in zip(sub[0], sub[1])]
        elif ch_code == 9:  # side, right
            sub[0] = [s + r for s, r in zip(sub[0], sub[1])]
        elif ch_code == 10:  # mid, side
            left, right = [], []
            for m, s in zip(sub[0], sub[1]):
                m = (m << 1) | (s & 1)
                left.append((m + s) >> 1)
                right.append((m - s) >> 1)
            sub = [left, right]
        br.pos = (br.pos + 7) & ~7
        end = br.pos // 8
        if end + 2 > len(data) or _crc(data[o:end], 0x8005, 16) != int.from_bytes(data[end : end + 2], "big"):
            raise RuntimeError(f"{path}: FLAC frame CRC mismatch at byte {o}")
        o = end + 2
        for c in range(nch):
            chans[c].extend(sub[c])
    if total and len(chans[0]) < total:
        raise RuntimeError(f"{path}: {len(chans[0])} samples decoded, STREAMINFO announces {total}")
    x = np.array(chans, dtype=np.int64).T  # (N, C)
    if total:
        x = x[:total]
    ft = np.float32 if np.dtype(dtype) == np.float32 and bits <= 24 else np.float64
    x = x.astype(ft) * ft(1.0 / (1 << (bits - 1)))
    if nch == 1 and not always_2d:
        x = x[:, 0]
    return x.astype(dtype, copy=False), si["rate"]


def read_wav(path: Union[Path, str], dtype="float64", always_2d: bool = False) -> Tuple[np.ndarray, int]:
    """(samples, rate): mono -> (N,), multi-channel -> (N, C); float in [-1, 1) like soundfile.read.
    RIFF/WAVE is parsed here; a FLAC stream (the recipes' default audio_format) goes to `read_flac`."""
    path = str(path)
    if path.rstrip().endswith("|"):
        raise NotImplementedError(f"piped wav.scp entries are not supported: {path!r}")
    with open(path, "rb") as f:
        head = f.read(12)
        if head[:4] == b"fLaC":
            return read_flac(path, dtype=dtype, always_2d=always_2d)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise NotImplementedError(f"{path}: only RIFF/WAVE (PCM or IEEE float) and FLAC are read natively")
        fmt = None
        data = None
        while True:
            ck = f.read(8)
            if len(ck) < 8:
                break
            cid, size = ck[:4], struct.unpack("<I", ck[4:])[0]
            if cid == b"fmt ":
                raw = f.read(size)
                tag, nch, rate, _, _, bits = struct.unpack("<HHIIHH", raw[:16])
                if tag == _WAVE_FORMAT_EXTENSIBLE and size >= 26:
                    tag = struct.unpack("<H", raw[24:26])[0]
                fmt = (tag, nch, rate, bits)
            elif cid == b"data":
                data = f.read(size)
                break
            else:
                f.seek(size, 1)
            if size & 1:
                f.seek(1, 1)
    if fmt is None or data is None:
        raise RuntimeError(f"{path}: missing 'fmt ' or 'data' chunk")
    tag, nch, rate, bits = fmt
    bps = bits // 8
    n = len(data) // (bps * nch) * nch
    # integers of up to 24 bits times a power of two are exact in float32: when float32 is what the
    # caller wants (the decode CLI), skip the float64 detour — same values, half the host traffic
    ft = np.float32 if np.dtype(dtype) == np.float32 else np.float64
    if tag == _WAVE_FORMAT_IEEE_FLOAT:
        x = np.frombuffer(data, dtype="<f4" if bits == 32 else "<f8", count=n).astype(np.float64)
    elif tag == _WAVE_FORMAT_PCM:
        if bits == 8:
            x = (np.frombuffer(data, dtype=np.uint8, count=n).astype(ft) - ft(128.0)) / ft(128.0)
        elif bits == 16:
            # one pass: int16 samples are converted and scaled by the ufunc itself (exact: a power of two)
            x = np.multiply(np.frombuffer(data, dtype="<i2", count=n), ft(1.0 / 32768.0), dtype=ft)
        elif bits == 24:
            b = np.frombuffer(data, dtype=np.uint8, count=n * 3).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v & 0x800000, v - 0x1000000, v)
            x = v.astype(ft) * ft(1.0 / 8388608.0)
        elif bits == 32:
            x = np.frombuffer(data, dtype="<i4", count=n).astype(np.float64) / 2147483648.0
        else:
            raise NotImplementedError(f"{path}: {bits}-bit PCM")
    else:
        raise NotImplementedError(f"{path}: WAVE format tag {tag:#x}")
    if nch > 1 or always_2d:
        x = x.reshape(-1, nch)
    return x.astype(dtype, copy=False), rate


def write_wav_pcm16(path: Union[Path, str], samples: np.ndarray, rate: int) -> None:
    """Float samples in [-1, 1) -> 16-bit PCM RIFF/WAVE (test fixtures and `SoundScpWriter`-style
    dumps; rounding like libsndfile's default float->short conversion: scale by 32768, clip)."""
    x = np.asarray(samples)
    nch = 1 if x.ndim == 1 else x.shape[1]
    pcm = np.clip(np.rint(x.astype(np.float64) * 32768.0), -32768, 32767).astype("<i2").tobytes()
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(pcm), b"WAVE", b"fmt ", 16, _WAVE_FORMAT_PCM,
                      nch, rate, rate * nch * 2, nch * 2, 16, b"data", len(pcm))
    with open(path, "wb") as f:
        f.write(hdr + pcm)


class WavBatchReader:
    """Window-level reader of the decode CLI on the C-ABI host functions `em_wav_probe` / `em_wav_load_rows`
    (csrc/host_io.cpp): headers of a whole window are parsed first (lengths for the length bucketing), then
    every batch is decoded by a native thread pool straight into its zero-padded (B, Lmax) float32 matrix —
    the combined effect of `read_wav` per file and `common_collate_fn`, bit for bit, without holding the GIL.
    `probe` returns None when any file is outside what the native reader handles (neither RIFF/WAVE nor FLAC,
    multi-channel, a FLAC stream of unknown length, unreadable): the caller then takes the Python reader for that window, which raises the descriptive error."""

    def __init__(self, threads: int = 4):
        import ctypes as C

        from espnet_amd import lib as L

        self._C, self._L, self._lib, self.threads = C, L, L.load(), max(1, int(threads))
        self.seconds = dict(probe=0.0, alloc=0.0, decode=0.0)  # where the reader's time goes (tools/cli_bench.py)

    def probe(self, paths):
        C, L = self._C, self._L
        n = len(paths)
        if any(p.rstrip().endswith("|") for p in paths):
            return None
        t0 = time.perf_counter()
        arr = (C.c_char_p * n)(*[p.encode() for p in paths])
        info = (L.EmWavInfo * n)()
        rc = self._lib.em_wav_probe(arr, n, info, self.threads)
        self.seconds["probe"] += time.perf_counter() - t0
        if rc != L.EM_OK:
            return None
        return arr, info

    def load(self, probed, index, pin_memory: bool = False):
        """Rows `index` of a probed window -> ((B, Lmax) float32 tensor, lengths list)."""
        import torch

        C, L = self._C, self._L
        arr, info = probed
        n = len(index)
        sub = (C.c_char_p * n)(*[arr[i] for i in index])
        sinfo = (L.EmWavInfo * n)(*[info[i] for i in index])
        lens = [int(sinfo[i].frames) for i in range(n)]
        t0 = time.perf_counter()
        out = torch.empty((n, max(lens)), dtype=torch.float32, pin_memory=pin_memory and torch.cuda.is_available())
        t1 = time.perf_counter()
        rc = self._lib.em_wav_load_rows(sub, sinfo, n, out.data_ptr(), out.size(1), self.threads)
        self.seconds["alloc"] += t1 - t0
        self.seconds["decode"] += time.perf_counter() - t1
        if rc != L.EM_OK:
            raise OSError(f"em_wav_load_rows: {self._lib.em_error_string(rc).decode()} "
                          f"({[arr[i].decode() for i in index]})")
        return out, lens


def load_entry(value: str, kind: str) -> np.ndarray:
    """One scp value -> array, by data type name (iterable_dataset.py DATA_TYPES)."""
    if kind == "sound":
        return read_wav(value, dtype="float32")[0]  # the dataset casts float arrays to float32 anyway
    if kind == "npy":
        return np.load(value)
    raise NotImplementedError(f"data type {kind!r}: the decode CLI reads 'sound' (wav) and 'npy' entries")


class SoundScpReader(collections.abc.Mapping):
    """`reader[key] -> (rate, array)` over a `wav.scp` (sound_scp.py:82-155)."""

    def __init__(self, fname, dtype=None, always_2d: bool = False):
        self.fname, self.dtype, self.always_2d = fname, dtype, always_2d
        self.data = read_2columns_text(fname)

    def __getitem__(self, key) -> Tuple[int, np.ndarray]:
        array, rate = read_wav(self.data[key], dtype=self.dtype or "float64", always_2d=self.always_2d)
        return rate, array

    def get_path(self, key):
        return self.data[key]

    def __contains__(self, item):
        return item in self.data

    def __len__(self):
        return len(self.data)

    def __iter__(self):
        return iter(self.data)

    def keys(self):
        return self.data.keys()
