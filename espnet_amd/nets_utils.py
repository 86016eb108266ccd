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


# mask slices of the Conv2dSubsampling variants as (trim, step): mask[:, :, :-trim:step] per conv
# (subsampling.py:448-449 conv2d, :758 conv2d6, :851 conv2d8); minimum input frames (check_short_utt :31-49)
SUBSAMPLING_MASK_SLICES = {"conv2d": [(2, 2), (2, 2)], "conv2d6": [(2, 2), (4, 3)], "conv2d8": [(2, 2), (2, 2), (2, 2)]}
SUBSAMPLING_MIN_FRAMES = {"conv2d": 7, "conv2d6": 11, "conv2d8": 15}
SUBSAMPLING_CONVS = {"conv2d": [(3, 2), (3, 2)], "conv2d6": [(3, 2), (5, 3)], "conv2d8": [(3, 2), (3, 2), (3, 2)]}


def conv_out_size(n: int, input_layer: str = "conv2d") -> int:
    """Output extent of the conv stack along one axis (time or frequency)."""
    for k, s in SUBSAMPLING_CONVS[input_layer]:
        n = (n - k) // s + 1
    return n


def conv2d_subsampled_lengths(flens: Sequence[int], tmax: int, input_layer: str = "conv2d") -> List[int]:
    """Valid counts after the mask slices of the input layer, e.g. `mask[:, :, :-2:2][:, :, :-2:2]`
    (subsampling.py:448-449).  Every slice acts on the padded mask of the current length t:
    count = #{i in range(0, t - trim, step) : i < len}."""
    out = []
    for n in flens:
        n, t = int(n), tmax
        for trim, step in SUBSAMPLING_MASK_SLICES[input_layer]:
            kept = len(range(0, max(t - trim, 0), step))
            n = min((n + step - 1) // step, kept)
            t = kept
        out.append(n)
    return out
