"""Test infrastructure: the control flow of the reference's label-synchronous search, restated over the scorer
interface only (`batch_init_state`, `batch_score`, `batch_score_partial`, `select_state`), so that scorers can
be exercised exactly the way the reference's BatchBeamSearch exercises them, on a box where /root/reference
does not exist.

Follows espnet2/legacy/nets/beam_search.py:36-126 (scorer bookkeeping), :385-498 (`forward`: length bounds,
main loop, end detection, n-best sort) and espnet2/legacy/nets/batch_beam_search.py:98-122 (`batch_beam`),
:124-153 (`init_hyp`), :253-357 (`search`), :359-423 (`post_process`).  Pinned to the reference itself by
tests/test_cpu_reference_binding.py (reference scorers through this driver reproduce the reference's own
n-best fixture)."""
import math
from typing import Dict, List

import torch


def end_detect(ended: List[dict], i: int, M: int = 3, D_end: float = math.log(1 * math.exp(-10))) -> bool:
    """espnet2/legacy/nets/e2e_asr_common.py:14-44."""
    if not ended:
        return False
    best = max(ended, key=lambda h: h["score"])
    count = 0
    for m in range(M):
        same = [h for h in ended if len(h["yseq"]) == i - m]
        if same and max(same, key=lambda h: h["score"])["score"] - best["score"] < D_end:
            count += 1
    return count == M


def _is_partial(scorer) -> bool:
    return hasattr(scorer, "batch_score_partial")


def drive_search(scorers: Dict[str, object], weights: Dict[str, float], beam_size: int, vocab_size: int, sos: int,
                 eos: int, x: torch.Tensor, maxlenratio: float = 0.0, minlenratio: float = 0.0,
                 pre_beam_ratio: float = 1.5, pre_beam_score_key: str = "full") -> List[dict]:
    """x (T, d) encoder output.  Returns the n-best: dict(yseq list, score float, scores {name: float})."""
    used = {k: v for k, v in scorers.items() if weights.get(k, 0) != 0 and v is not None}
    full = {k: v for k, v in used.items() if not _is_partial(v)}
    part = {k: v for k, v in used.items() if _is_partial(v)}
    pre_beam_size = int(pre_beam_ratio * beam_size)
    do_pre_beam = pre_beam_score_key is not None and pre_beam_size < vocab_size and len(part) > 0
    T = x.shape[0]
    maxlen = T if maxlenratio == 0 else (-int(maxlenratio) if maxlenratio < 0 else max(1, int(maxlenratio * T)))
    minlen = -int(minlenratio) if minlenratio < 0 else int(minlenratio * T)

    # init_hyp: one hypothesis holding <sos>
    hyps = [dict(yseq=[sos], score=0.0, scores={k: 0.0 for k in used},
                 states={k: s.batch_init_state(x) for k, s in used.items()})]
    ended: List[dict] = []
    for i in range(maxlen):
        n = len(hyps)
        yseq = torch.tensor([h["yseq"] for h in hyps], dtype=torch.int64, device=x.device)
        weighted = torch.zeros(n, vocab_size, dtype=x.dtype, device=x.device)
        scores, states = {}, {}
        for k, s in full.items():
            scores[k], states[k] = s.batch_score(yseq, [h["states"][k] for h in hyps], x.expand(n, *x.shape))
            weighted += weights[k] * scores[k]
        part_ids = None
        if do_pre_beam:
            src = weighted if pre_beam_score_key == "full" else scores[pre_beam_score_key]
            part_ids = torch.topk(src, pre_beam_size, dim=-1)[1]
        part_scores, part_states = {}, {}
        for k, s in part.items():
            part_scores[k], part_states[k] = s.batch_score_partial(yseq, part_ids, [h["states"][k] for h in hyps], x)
            weighted += weights[k] * part_scores[k]
        weighted += torch.tensor([h["score"] for h in hyps], dtype=x.dtype, device=x.device).unsqueeze(1)

        top = weighted.view(-1).topk(beam_size)[1]
        prev_ids = torch.div(top, vocab_size, rounding_mode="trunc").tolist()
        new_toks = (top % vocab_size).tolist()
        new_hyps = []
        for p, t in zip(prev_ids, new_toks):
            h = hyps[p]
            sc = dict(h["scores"])
            for k in full:
                sc[k] = h["scores"][k] + float(scores[k][p, t])
            for k in part:
                sc[k] = h["scores"][k] + float(part_scores[k][p, t])
            st = {k: full[k].select_state(v, p) for k, v in states.items()}
            st.update({k: part[k].select_state(v, p, t) for k, v in part_states.items()})
            new_hyps.append(dict(yseq=h["yseq"] + [t], score=float(weighted[p, t]), scores=sc, states=st))

        # post_process
        if i == maxlen - 1:
            for h in new_hyps:
                h["yseq"] = h["yseq"] + [eos]
        hyps = []
        for h in new_hyps:
            if h["yseq"][-1] == eos:
                if i >= minlen:
                    ended.append(h)
            else:
                hyps.append(h)
        if maxlenratio == 0.0 and end_detect(ended, i):
            break
        if not hyps:
            break
    nbest = sorted(ended, key=lambda h: h["score"], reverse=True)
    if not nbest and minlenratio >= 0.1:
        return drive_search(scorers, weights, beam_size, vocab_size, sos, eos, x, maxlenratio,
                            max(0.0, minlenratio - 0.1), pre_beam_ratio, pre_beam_score_key)
    return [dict(yseq=h["yseq"], score=h["score"], scores=h["scores"]) for h in nbest]
