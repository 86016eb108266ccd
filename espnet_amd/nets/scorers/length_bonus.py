"""LengthBonus (espnet2/legacy/nets/scorers/length_bonus.py:10-58): a constant 1.0 per emitted token,
weighted by `penalty`.  Inside the fused device search the constant is folded into the candidate totals
(csrc/search.hip `full_score`); `score` / `batch_score` are the reference's per-step calls."""
from typing import Any, List, Tuple

import torch

from espnet_amd.nets.scorer_interface import BatchScorerInterface


class LengthBonus(BatchScorerInterface):
    def __init__(self, n_vocab: int):
        self.n = n_vocab

    def score(self, y, state, x):
        """length_bonus.py:24-37: (ones (V,), None) on x's device / dtype."""
        return torch.tensor([1.0], device=x.device, dtype=x.dtype).expand(self.n), None

    def batch_score(self, ys: torch.Tensor, states: List[Any], xs: torch.Tensor) -> Tuple[torch.Tensor, List[Any]]:
        """length_bonus.py:39-58: (ones (n, V), None)."""
        return torch.tensor([1.0], device=xs.device, dtype=xs.dtype).expand(ys.shape[0], self.n), None
