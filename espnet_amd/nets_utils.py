"""Host-side length bookkeeping (integers only; no device work).

Mirrors espnet2/legacy/nets/pytorch_backend/nets_utils.py:65 `make_pad_mask` and the length
formulas of espnet2/layers/stft.py:108-115 and transformer/subsampling.py:448-449.
"""
from typing import List, Sequence

import torch


def make_pad_mask(lengths, maxlen: int = None) -> torch.Tensor:
    """True at padded positions (nets_utils.py:65)."""
    lengths = torch.as_tensor(lengths, dtype=torch.long)
    maxlen = int(lengths.max()) if maxlen is None else maxlen
    return torch.arange(maxlen)[None, :] >= lengths[:, None]


def stft_frame_lengths(ilens: Sequence[int], n_fft: int, hop: int, center: bool = True) -> List[int]:
    """olens = (ilens + 2*(n_fft//2) - n_fft) // hop + 1   (stft.py:108-115)."""
    pad = n_fft // 2 if center else 0
    return [(int(n) + 2 * pad - n_fft) // hop + 1 for n in ilens]


def conv2d_subsampled_lengths(flens: Sequence[int], tmax: int) -> List[int]:
    """Valid counts after `mask[:, :, :-2:2][:, :, :-2:2]` (subsampling.py:448-449).  The slices
    act on the padded mask of length tmax: count = #{even i < tmax-2 : i < len}, applied twice."""

    def once(n, t):
        kept = len(range(0, max(t - 2, 0), 2))
        return min((n + 1) // 2, kept), kept

    out = []
    for n in flens:
        a, t1 = once(int(n), tmax)
        b, _ = once(a, t1)
        out.append(b)
    return out
