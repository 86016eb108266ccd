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


class _PartialStates:
    """The per-candidate states of `score_partial`: `st[i]` = state of the prefix extended by ids[i] (the reference
    holds an array of all of them, scorers/ctc.py:83-86; here each is materialised when the search selects it)."""

    def __init__(self, scorer, bst, ids):
        self.scorer, self.bst, self.ids = scorer, bst, ids

    def __getitem__(self, i):
        return self.scorer.select_state(self.bst, 0, int(self.ids[int(i)]))

    def __len__(self):
        return int(self.ids.numel())


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
        """scorers/ctc.py:25-38 (the non-batched set-up): `(0, initial state)`; same device tables as the batched
        set-up (the reference switches to the numpy CTCPrefixScore here, ctc_prefix_score.py:273-370: same recursion,
        one prefix at a time)."""
        self.batch_init_state(x)
        return 0.0, None

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
        """scorers/ctc.py:65-86: scores of the candidates `ids` ONLY, shape (len(ids),), and the next state
        `(presub_score (len(ids),), per-candidate states)` that `select_state(state, i)` indexes by the POSITION i
        inside ids (scorers/ctc.py:51-55).  Runs through the batched device entry with one prefix."""
        prev_score, st = state
        scores, bst = self.batch_score_partial(y.unsqueeze(0), ids.unsqueeze(0), [st], x)
        idx = ids.to(scores.device)
        presub = bst.log_psi[0].index_select(0, idx)
        tscore = (presub - prev_score).to(dtype=x.dtype if x.dtype.is_floating_point else torch.float32)
        return tscore, (presub, _PartialStates(self, bst, idx))

    def select_state(self, state, i, new_id=None):
        """scorers/ctc.py:40-63: the state of prefix i extended by `new_id`."""
        if state is None:
            return None
        if type(state) is tuple and len(state) == 2:  # score_partial's state: index by position in ids (:51-55)
            sc, st = state
            return sc[int(i)], st[int(i)]
        if not isinstance(state, _BatchState):  # a list of per-hypothesis states (batchfy): scorers/ctc.py:63
            return state[i]
        if new_id is None:
            raise ValueError("select_state on a batch_score_partial state needs new_id (scorers/ctc.py:56-62)")
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

    # ---------------------------------------------------------------- streaming (block-synchronous) decoding
    @torch.no_grad()
    def extend_prob(self, x: torch.Tensor):
        """scorers/ctc.py:128-139 -> CTCPrefixScoreTH.extend_prob (ctc_prefix_score.py:226-246): the visible memory
        grew to x (T_new, d).  Log-probs of all T_new frames; the frames already seen keep the values they had
        (:243), and nothing happens unless the memory actually grew (:231)."""
        if self._lpT is None:
            raise RuntimeError("batch_init_state(x) must run before extend_prob")
        L.require_gpu(x, "x")
        T_new = x.size(0)
        if T_new <= self._T:
            return
        old, T_old = self._lpT, self._T
        self.batch_init_state(x)              # lpT (V, T_new), xlens, r0 of the empty prefix over T_new frames
        self._lpT[:, :T_old].copy_(old)
        if T_old > 0:                         # r0 follows the kept frames
            L.check(L.load().em_ctc_prefix_init(L.ptr(self._lpT), L.ptr(self._xlens), 1, T_new, self.blank,
                                                L.ptr(self._r0), L.current_stream_ptr()), "em_ctc_prefix_init")

    @torch.no_grad()
    def extend_state(self, state):
        """scorers/ctc.py:141-157 -> CTCPrefixScoreTH.extend_state (ctc_prefix_score.py:248-270) for every
        hypothesis state of the list: the forward variables continue over the new frames along the blank path
        (Eq. 14 of arXiv:2006.14941).  One device launch for the whole list (`em_ctc_prefix_extend`)."""
        if self._lpT is None:
            raise RuntimeError("batch_init_state(x) must run before extend_state")
        live = [k for k, s in enumerate(state) if s is not None]
        out = list(state)
        if not live:
            return out
        T_new = self._T
        T_old = state[live[0]][0].size(0)
        if any(state[k][0].size(0) != T_old for k in live):
            raise ValueError("extend_state: the states of one beam cover different frame counts")
        if T_old >= T_new:
            return out
        dev = self._lpT.device
        r_old = torch.stack([state[k][0] for k in live], 0).to(device=dev, dtype=torch.float32).contiguous()
        r_new = torch.empty(len(live), T_new, 2, dtype=torch.float32, device=dev)
        L.check(L.load().em_ctc_prefix_extend(L.ptr(self._lpT), T_new, self.blank, L.ptr(r_old), len(live), T_old,
                                              L.ptr(r_new), L.current_stream_ptr()), "em_ctc_prefix_extend")
        for j, k in enumerate(live):
            out[k] = (r_new[j],) + tuple(state[k][1:])
        return out
