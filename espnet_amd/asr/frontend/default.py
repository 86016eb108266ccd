"""DefaultFrontend: STFT -> power -> log-mel on the MI355X (csrc/frontend.hip).

Mirrors espnet2/asr/frontend/default.py:17-131 (constructor arguments, `output_size()`,
`forward(input, input_lengths) -> (feats, feats_lens)`) for the single-channel path without
WPE/beamformer (`frontend_conf=None`), which is the only configuration the ASR recipes of
BASELINE.json use.  State-dict key: `logmel.melmat`.
"""
from typing import Optional, Tuple, Union

import torch

from espnet_amd import lib as L
from espnet_amd.layers.log_mel import LogMel, pack_banded
from espnet_amd.nets_utils import stft_frame_lengths


def _parse_fs(fs: Union[int, str]) -> int:
    if isinstance(fs, str):
        s = fs.strip().lower()
        mult = 1
        if s.endswith("k"):
            mult, s = 1000, s[:-1]
        return int(float(s) * mult)
    return int(fs)


class DefaultFrontend(torch.nn.Module):
    def __init__(self, fs: Union[int, str] = 16000, n_fft: int = 512, win_length: Optional[int] = None,
                 hop_length: int = 128, window: Optional[str] = "hann", center: bool = True,
                 normalized: bool = False, onesided: bool = True, n_mels: int = 80,
                 fmin: Optional[int] = None, fmax: Optional[int] = None, htk: bool = False,
                 frontend_conf: Optional[dict] = None, apply_stft: bool = True):
        super().__init__()
        fs = _parse_fs(fs)
        if n_fft != 512 or not center or normalized or not onesided or not apply_stft:
            raise NotImplementedError(
                "MI355X frontend fast path: n_fft=512, center=True, normalized=False, onesided=True")
        if frontend_conf is not None:
            raise NotImplementedError("WPE/beamformer frontend_conf is outside the MI355X fast path")
        if window not in ("hann", None):
            raise NotImplementedError(f"window={window}")
        self.fs, self.n_fft, self.hop_length = fs, n_fft, hop_length
        self.win_length = n_fft if win_length is None else win_length
        self.window = window
        self.n_mels = n_mels
        self.frontend_type = "default"
        self.logmel = LogMel(fs=fs, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax, htk=htk)
        self._packed = None

    def output_size(self) -> int:
        return self.n_mels

    # ---- load-time packing (host) -------------------------------------------------------------
    def _window_padded(self) -> torch.Tensor:
        """torch.stft semantics (stft.py:76-79,94): hann(win_length) periodic, zero padded on both
        sides to n_fft with left = (n_fft - win_length) // 2."""
        if self.window is None:
            w = torch.ones(self.win_length)
        else:
            w = torch.hann_window(self.win_length, dtype=torch.float32)
        left = (self.n_fft - self.win_length) // 2
        out = torch.zeros(self.n_fft)
        out[left:left + self.win_length] = w
        return out

    def pack(self, device):
        packed, lo, maxlen = pack_banded(self.logmel.melmat)
        self._packed = dict(window=self._window_padded().to(device), mel=packed.to(device),
                            lo=lo.to(device), maxlen=maxlen, device=torch.device(device))
        return self._packed

    # ---- forward --------------------------------------------------------------------------------
    def feature_lengths(self, input_lengths) -> list:
        return stft_frame_lengths([int(n) for n in input_lengths], self.n_fft, self.hop_length)

    def forward_device(self, speech: torch.Tensor, flens_dev: torch.Tensor,
                       wlens_dev: torch.Tensor = None) -> torch.Tensor:
        """speech (B, N) f32 on the GPU; flens_dev (B,) i32 on the GPU.  Returns feats (B,T_f,n_mels).
        wlens_dev (B,) i32 sample counts: reflect-pad every utterance at its own end (the utterance
        decoded alone) instead of at the padded length N (torch.stft on the padded batch)."""
        L.require_gpu(speech, "speech")
        if self._packed is None or self._packed["device"] != speech.device:
            self.pack(speech.device)
        pk = self._packed
        B, N = speech.shape
        if N <= self.n_fft // 2:
            raise ValueError(f"input of {N} samples is too short for reflect padding of {self.n_fft // 2}")
        T_f = 1 + N // self.hop_length
        feats = torch.empty(B, T_f, self.n_mels, dtype=torch.float32, device=speech.device)
        L.check(L.load().em_frontend_logmel_f32(
            L.ptr(speech), B, N, self.hop_length, L.ptr(pk["window"]), L.ptr(pk["mel"]),
            L.ptr(pk["lo"]), pk["maxlen"], self.n_mels, L.ptr(flens_dev), L.ptr(wlens_dev), T_f,
            L.ptr(feats), L.current_stream_ptr()), "em_frontend_logmel_f32")
        return feats

    def forward(self, input: torch.Tensor, input_lengths: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        L.require_gpu(input, "input")
        speech = input.to(torch.float32).contiguous()
        flens = self.feature_lengths(input_lengths.tolist())
        flens_dev = torch.tensor(flens, dtype=torch.int32).to(speech.device, non_blocking=True)
        feats = self.forward_device(speech, flens_dev)
        return feats, torch.tensor(flens, dtype=torch.long, device=input.device)
