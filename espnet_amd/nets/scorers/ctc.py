"""CTCPrefixScorer (espnet2/legacy/nets/scorers/ctc.py:10-157) on the MI355X.

In the reference this class owns a `CTCPrefixScoreTH` (legacy/nets/ctc_prefix_score.py:13-270) and is called
once per search step from Python (`batch_score_partial`, `select_state`).  Inside the fused device search
(`espnet_amd.nets.batch_beam_search`, csrc/search.hip `candidate_kernel` / `ctc_state_kernel`) the object only
carries the CTC head and the <eos> id.  The methods below are the reference's per-step interface on the same
device code (`em_ctc_log_probs_t`, `em_ctc_prefix_init`, `em_ctc_prefix_score`, `em_ctc_prefix_state`), so the
reference's BatchBeamSearch can drive it unchanged.

State of one hypothesis (opaque to the search, like the reference's `(r, s, f_min, f_max)`):
`(r (T, 2) f32 forward variables r^n | r^b, s = log psi of the prefix, last token)`."""
from typing import Any, List

import torch

from espnet_amd import lib as L
from espnet_amd.nets.scorer_interface import BatchPartialScorerInterface

LOGZERO = -10000000000.0  # ctc_prefix_score.py:34


class _BatchState:
    """What `batch_score_partial` returns for the n prefixes of a step: the prefix states plus the candidate
    table; `select_state` materialises the forward variables of the chosen (row, label) pairs (the reference
    computes r for every candidate, ctc_prefix_score.py:158-164, and indexes it, scorers/ctc.py:54-62)."""

    def __init__(self, scorer, out_len, r_prev, last, log_psi_full):
        self.scorer, self.out_len, self.r_prev, self.last, self.log_psi = scorer, out_len, r_prev, last, log_psi_full
        self._cache = {}


class CTCPrefixScorer(BatchPartialScorerInterface):
    def __init__(self, ctc, eos: int):
        self.ctc = ctc
        self.eos = eos
        self.blank = 0
        self._lpT = None

    # ---------------------------------------------------------------- per-utterance set-up
    def batch_init_state(self, x: torch.Tensor):
        """scorers/ctc.py:88-101: log-softmax of the whole utterance once.  x (T, d) on the GPU."""
        L.require_gpu(x, "x")
        ctc = self.ctc
        T, d = x.shape
        act = ctc._to_act(x.unsqueeze(0))
        p = ctc._pack(x.device)
        V = ctc.odim
        lpT = torch.empty(V, T, dtype=torch.float32, device=x.device)
        lib = L.load()
        L.check(lib.em_ctc_log_probs_t(ctc.em_dtype, L.ptr(act), 1, T, d, L.ptr(p["w"]), L.ptr(p["b"]), V,
                                       L.ptr(lpT), L.current_stream_ptr()), "em_ctc_log_probs_t")
        xlens = torch.tensor([T], dtype=torch.int32, device=x.device)
        r0 = torch.empty(1, T, 2, dtype=torch.float32, device=x.device)
        L.check(lib.em_ctc_prefix_init(L.ptr(lpT), L.ptr(xlens), 1, T, self.blank, L.ptr(r0),
                                       L.current_stream_ptr()), "em_ctc_prefix_init")
        self._lpT, self._xlens, self._r0, self._T, self._V = lpT, xlens, r0, T, V
        return None

    def init_state(self, x: torch.Tensor):
        """scorers/ctc.py:27-38 (the non-batched set-up): same device tables."""
        self.batch_init_state(x)
        return None

    # ---------------------------------------------------------------- per-step scoring
    @torch.no_grad()
    def batch_score_partial(self, y: torch.Tensor, ids: torch.Tensor, state: List[Any], x: torch.Tensor):
        """scorers/ctc.py:103-126 -> CTCPrefixScoreTH.__call__ (ctc_prefix_score.py:71-191).
        y (n, L) int64 prefixes incl. <sos>; ids (n, S) int64 pre-beam candidates or None (all labels);
        state list[n] (None at the first step).  Returns (scores (n, V) f32 = log psi - s_prev with logzero
        outside the candidates, batch state for `select_state`)."""
        if self._lpT is None:
            raise RuntimeError("batch_init_state(x) must run before batch_score_partial")
        dev = self._lpT.device
        n, Lc = y.shape
        T, V = self._T, self._V
        out_len = Lc - 1
        last = y[:, -1].to(device=dev, dtype=torch.int32).contiguous()
        if state is None or state[0] is None:
            r_prev = self._r0.expand(n, T, 2).contiguous()
            s_prev = torch.zeros(n, dtype=torch.float32, device=dev)
        else:
            r_prev = torch.stack([s[0] for s in state], 0).contiguous()
            s_prev = torch.stack([s[1] for s in state], 0).to(torch.float32).contiguous()
        lib = L.load()
        if ids is not None:
            S = ids.size(1)
            cand = ids.to(device=dev, dtype=torch.int32).contiguous()
            psi = torch.empty(n, S + 1, dtype=torch.float32, device=dev)
            L.check(lib.em_ctc_prefix_score(L.ptr(self._lpT), L.ptr(self._xlens), L.ptr(r_prev), L.ptr(s_prev),
                                            L.ptr(last), L.ptr(cand), 1, n, S, T, V, out_len, self.eos, self.blank,
                                            L.ptr(psi), None, L.current_stream_ptr()), "em_ctc_prefix_score")
            log_psi = torch.full((n, V), LOGZERO, dtype=torch.float32, device=dev)
            log_psi.scatter_(1, ids.to(dev), psi[:, :S])  # :170-175
            log_psi[:, self.eos] = psi[:, S]               # :184-186
            if self.eos != self.blank:
                log_psi[:, self.blank] = LOGZERO            # :188-190
        else:
            log_psi = torch.empty(n, V, dtype=torch.float32, device=dev)
            L.check(lib.em_ctc_prefix_score(L.ptr(self._lpT), L.ptr(self._xlens), L.ptr(r_prev), L.ptr(s_prev),
                                            L.ptr(last), None, 1, n, V, T, V, out_len, self.eos, self.blank,
                                            L.ptr(log_psi), None, L.current_stream_ptr()), "em_ctc_prefix_score")
        scores = log_psi - s_prev.unsqueeze(1)
        return scores, _BatchState(self, out_len, r_prev, last, log_psi)

    def score_partial(self, y, ids, state, x):
        """scorers/ctc.py:65-86, through the batched entry (one prefix)."""
        scores, st = self.batch_score_partial(y.unsqueeze(0), None if ids is None else ids.unsqueeze(0),
                                              [state], x)
        return scores[0], st

    def select_state(self, state, i, new_id=None):
        """scorers/ctc.py:40-63: the state of prefix i extended by `new_id`."""
        if state is None:
            return None
        if not isinstance(state, _BatchState):  # a list of per-hypothesis states (batchfy): scorers/ctc.py:63
            return state[i]
        i, new_id = int(i), int(new_id)
        key = (i, new_id)
        if key not in state._cache:
            dev = state.r_prev.device
            T = self._T
            r = torch.full((1, T, 2), LOGZERO, dtype=torch.float32, device=dev)
            rows = torch.tensor([i], dtype=torch.int32, device=dev)
            toks = torch.tensor([new_id], dtype=torch.int32, device=dev)
            L.check(L.load().em_ctc_prefix_state(L.ptr(self._lpT), L.ptr(self._xlens), L.ptr(state.r_prev),
                                                 L.ptr(state.last), L.ptr(rows), L.ptr(toks), 1, 1,
                                                 state.r_prev.size(0), T, state.out_len, self.blank, L.ptr(r),
                                                 L.current_stream_ptr()), "em_ctc_prefix_state")
            state._cache[key] = (r[0], state.log_psi[i, new_id], new_id)
        return state._cache[key]

    def extend_prob(self, x: torch.Tensor):
        raise NotImplementedError("streaming: BatchBeamSearchOnline drives em_search_online_extend directly")

    def extend_state(self, state):
        raise NotImplementedError("streaming: BatchBeamSearchOnline drives em_search_online_extend directly")
