"""Hypothesis container of the beam search (espnet2/legacy/nets/beam_search.py:15-31)."""
from typing import Any, Dict, List, NamedTuple, Union

import torch


class Hypothesis(NamedTuple):
    yseq: torch.Tensor
    score: Union[float, torch.Tensor] = 0
    scores: Dict[str, Union[float, torch.Tensor]] = dict()
    states: Dict[str, Any] = dict()
    hs: List[torch.Tensor] = []

    def asdict(self) -> dict:
        return self._replace(
            yseq=self.yseq.tolist(), score=float(self.score),
            scores={k: float(v) for k, v in self.scores.items()})._asdict()
