"""Audio readers for the decode CLI (espnet2/fileio/sound_scp.py:13-155, npy_scp.py, and the
`sound` / `npy` entries of espnet2/train/iterable_dataset.py:44-67).

The reference reads through `soundfile` (libsndfile), which is not in this image; RIFF/WAVE PCM is
decoded here directly with the same sample convention libsndfile uses for float reads:
integer PCM of width w bytes -> value / 2**(8w-1) (8-bit is unsigned, offset 128); IEEE float is
taken as is.  Other containers (flac, sph, pipes `cmd |`) raise NotImplementedError naming the file:
convert them to wav with the recipe's `format_wav_scp.sh` stage, as the recipes already do.
"""
import collections.abc
import struct
import time
from pathlib import Path
from typing import Tuple, Union

import numpy as np

from espnet_amd.fileio.read_text import read_2columns_text

_WAVE_FORMAT_PCM, _WAVE_FORMAT_IEEE_FLOAT, _WAVE_FORMAT_EXTENSIBLE = 0x0001, 0x0003, 0xFFFE


def read_wav(path: Union[Path, str], dtype="float64", always_2d: bool = False) -> Tuple[np.ndarray, int]:
    """(samples, rate): mono -> (N,), multi-channel -> (N, C); float in [-1, 1) like soundfile.read."""
    path = str(path)
    if path.rstrip().endswith("|"):
        raise NotImplementedError(f"piped wav.scp entries are not supported: {path!r}")
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise NotImplementedError(f"{path}: only RIFF/WAVE (PCM or IEEE float) is read natively")
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
    `probe` returns None when any file is outside what the native reader handles (not RIFF/WAVE, multi-channel,
    unreadable): the caller then takes the Python reader for that window, which raises the descriptive error."""

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
