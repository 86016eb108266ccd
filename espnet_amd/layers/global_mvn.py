"""GlobalMVN (espnet2/layers/global_mvn.py:13-100) on the MI355X: the statistics file handling
(Kaldi-style .npy or .npz with count / sum / sum_square) is the reference's, the normalisation
itself is csrc/frontend.hip `em_global_mvn_f32`.  Buffers `mean` / `std` as in the reference."""
from pathlib import Path
from typing import Tuple, Union

import numpy as np
import torch

from espnet_amd import lib as L


class GlobalMVN(torch.nn.Module):
    def __init__(self, stats_file: Union[Path, str], norm_means: bool = True, norm_vars: bool = True,
                 eps: float = 1.0e-20):
        super().__init__()
        self.norm_means, self.norm_vars, self.eps = norm_means, norm_vars, eps
        self.stats_file = Path(stats_file)
        stats = np.load(self.stats_file)
        if isinstance(stats, np.ndarray):  # Kaldi like stats (global_mvn.py:44-48)
            count = stats[0].flatten()[-1]
            mean = stats[0, :-1] / count
            var = stats[1, :-1] / count - mean * mean
        else:  # npz (:49-55)
            count = stats["count"]
            mean = stats["sum"] / count
            var = stats["sum_square"] / count - mean * mean
        std = np.sqrt(np.maximum(var, eps))
        self.register_buffer("mean", torch.as_tensor(np.asarray(mean)).float())
        self.register_buffer("std", torch.as_tensor(np.asarray(std)).float())

    def extra_repr(self):
        return f"stats_file={self.stats_file}, norm_means={self.norm_means}, norm_vars={self.norm_vars}"

    def forward_device(self, feats: torch.Tensor, flens_dev: torch.Tensor = None) -> torch.Tensor:
        """feats (B, T_f, D) f32 on the GPU, normalised IN PLACE (as the reference does, :87-97)."""
        L.require_gpu(feats, "feats")
        B, T_f, D = feats.shape
        mean = self.mean.to(feats.device) if self.norm_means else None
        std = self.std.to(feats.device) if self.norm_vars else None
        L.check(L.load().em_global_mvn_f32(L.ptr(feats), L.ptr(flens_dev), L.ptr(mean), L.ptr(std), B,
                                           T_f, D, L.current_stream_ptr()), "em_global_mvn_f32")
        return feats

    def forward(self, x: torch.Tensor, ilens: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if ilens is None:
            ilens = x.new_full([x.size(0)], x.size(1))
        flens_dev = ilens.to(device=x.device, dtype=torch.int32)
        return self.forward_device(x.to(torch.float32).contiguous(), flens_dev), ilens
