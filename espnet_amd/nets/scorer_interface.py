"""Scorer interface of the label-synchronous beam search (espnet2/legacy/nets/scorer_interface.py:7-190):
the method names, argument meaning and defaults the reference's `BeamSearch` / `BatchBeamSearch` call.

The MI355X scorers (TransformerDecoder, CTCPrefixScorer, LengthBonus, TransformerLM, SequentialRNNLM)
implement it twice: inside the fused device search (`espnet_amd.nets.batch_beam_search`, no per-step host
work) and, through the classes below, as one device call per search step, so that a search written
against this interface -- the reference's own BatchBeamSearch -- can be driven by the same kernels.
States are opaque to the search (it only hands them back through `select_state` / `batch_score*`)."""
from typing import Any, List, Tuple

import torch


class ScorerInterface:
    """scorer_interface.py:10-64."""

    def init_state(self, x: torch.Tensor) -> Any:
        return None

    def select_state(self, state: Any, i: int, new_id: int = None) -> Any:
        return None if state is None else state[i]

    def score(self, y: torch.Tensor, state: Any, x: torch.Tensor) -> Tuple[torch.Tensor, Any]:
        raise NotImplementedError

    def final_score(self, state: Any) -> float:
        return 0.0


class BatchScorerInterface(ScorerInterface):
    """scorer_interface.py:67-122: `batch_score(ys (n, L) int64, states list[n], xs (n, T, d))
    -> (log-probs (n, V), states list[n])`."""

    def batch_init_state(self, x: torch.Tensor) -> Any:
        return self.init_state(x)

    def batch_score(self, ys: torch.Tensor, states: List[Any], xs: torch.Tensor) -> Tuple[torch.Tensor, List[Any]]:
        raise NotImplementedError


class PartialScorerInterface(ScorerInterface):
    """scorer_interface.py:125-157: `score_partial(y, next_tokens, state, x)`."""

    def score_partial(self, y: torch.Tensor, next_tokens: torch.Tensor, state: Any, x: torch.Tensor):
        raise NotImplementedError


class BatchPartialScorerInterface(BatchScorerInterface, PartialScorerInterface):
    """scorer_interface.py:160-190: `batch_score_partial(ys (n, L), next_tokens (n, S), states, xs)
    -> (scores (n, V), states)`."""

    def batch_score_partial(self, ys: torch.Tensor, next_tokens: torch.Tensor, states: List[Any], xs: torch.Tensor):
        raise NotImplementedError
