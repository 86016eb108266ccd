"""LogMel parameter holder + Slaney mel filterbank (host side, load time only).

Mirrors espnet2/layers/log_mel.py:9-84: the filterbank is a persistent buffer `melmat`
(n_fft/2+1, n_mels) so reference checkpoints load unchanged.  The matrix itself follows
librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False) — Slaney scale, slaney norm — which
the reference calls at log_mel.py:50.  The log-mel arithmetic runs in csrc/frontend.hip.
"""
import math

import numpy as np
import torch


def _slaney_hz_to_mel(f: np.ndarray) -> np.ndarray:
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3.0)
    log = 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / math.log(6.4))
    return np.where(f >= 1000.0, log, lin)


def _slaney_mel_to_hz(m: np.ndarray) -> np.ndarray:
    m = np.asarray(m, dtype=np.float64)
    lin = m * (200.0 / 3.0)
    log = 1000.0 * np.exp((m - 15.0) * (math.log(6.4) / 27.0))
    return np.where(m >= 15.0, log, lin)


def mel_filterbank(fs: int, n_fft: int, n_mels: int, fmin: float, fmax: float, htk: bool = False) -> np.ndarray:
    """(n_fft//2+1, n_mels) float32, i.e. already transposed like the reference buffer."""
    if htk:
        raise NotImplementedError("htk=True mel scale is outside the MI355X fast path")
    edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(fmin), _slaney_hz_to_mel(fmax), n_mels + 2))
    bins = np.fft.rfftfreq(n_fft, 1.0 / fs)
    up = (bins[None, :] - edges[:-2, None]) / (edges[1:-1] - edges[:-2])[:, None]
    down = (edges[2:, None] - bins[None, :]) / (edges[2:] - edges[1:-1])[:, None]
    tri = np.maximum(0.0, np.minimum(up, down)).astype(np.float32)
    tri *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return np.ascontiguousarray(tri.T.astype(np.float32))


def pack_banded(melmat: torch.Tensor):
    """Band-pack the filterbank for the HIP kernel: every filter m is non-zero on a contiguous
    bin range [lo_m, hi_m); returns (packed [maxlen][n_mels] f32, lo [n_mels] i32, maxlen) with
    packed[s][m] = melmat[lo_m + s][m].  Exact for ANY matrix (interior zeros stay zeros; an
    all-zero column gets lo=0, len=1)."""
    m = melmat.detach().cpu().float().numpy()
    nbin, nmel = m.shape
    lo = np.zeros(nmel, dtype=np.int32)
    hi = np.ones(nmel, dtype=np.int32)
    for j in range(nmel):
        nz = np.nonzero(m[:, j])[0]
        if len(nz):
            lo[j], hi[j] = nz[0], nz[-1] + 1
    maxlen = int((hi - lo).max())
    packed = np.zeros((maxlen, nmel), dtype=np.float32)
    for j in range(nmel):
        n = hi[j] - lo[j]
        packed[:n, j] = m[lo[j]:hi[j], j]
    return torch.from_numpy(packed), torch.from_numpy(lo), maxlen


class LogMel(torch.nn.Module):
    """Parameter container matching espnet2.layers.log_mel.LogMel (buffer `melmat`)."""

    def __init__(self, fs: int = 16000, n_fft: int = 512, n_mels: int = 80, fmin: float = None,
                 fmax: float = None, htk: bool = False, log_base: float = None):
        super().__init__()
        fmin = 0 if fmin is None else fmin
        fmax = fs / 2 if fmax is None else fmax
        if log_base is not None:
            raise NotImplementedError("log_base != None is outside the MI355X fast path")
        self.mel_options = dict(sr=fs, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax, htk=htk)
        self.log_base = log_base
        self.register_buffer("melmat", torch.from_numpy(mel_filterbank(fs, n_fft, n_mels, fmin, fmax, htk)))

    def extra_repr(self):
        return ", ".join(f"{k}={v}" for k, v in self.mel_options.items())
