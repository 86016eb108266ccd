"""End detection of the label-synchronous search (espnet2/legacy/nets/e2e_asr_common.py:14-44)."""
import math
from typing import List


def end_detect(ended_hyps: List[dict], i: int, M: int = 3, D_end: float = math.log(1 * math.exp(-10))) -> bool:
    """True when, for each of the M most recent lengths i, i-1, .., the best ended hypothesis of that
    length (len(yseq) == i - m) scores more than |D_end| below the best ended hypothesis overall —
    lengths without an ended hypothesis do not count (Watanabe et al., Eq. 50)."""
    if not ended_hyps:
        return False
    best = max(h["score"] for h in ended_hyps)
    count = 0
    for m in range(M):
        same = [h["score"] for h in ended_hyps if len(h["yseq"]) == i - m]
        if same and max(same) - best < D_end:
            count += 1
    return count == M
