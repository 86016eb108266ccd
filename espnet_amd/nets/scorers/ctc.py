"""CTCPrefixScorer (espnet2/legacy/nets/scorers/ctc.py:10-126) as a search-configuration object.

In the reference this class owns a `CTCPrefixScoreTH` and is called once per search step from
Python (`batch_score_partial`, `select_state`).  On the MI355X the prefix-score recurrence, the
state selection and the rest of the step run inside one device-resident search
(csrc/search.hip: `candidate_kernel`, `update_kernel`), so the object only carries what the
search needs: the CTC head (for `log_softmax`) and the <eos> id."""


class CTCPrefixScorer:
    def __init__(self, ctc, eos: int):
        self.ctc = ctc
        self.eos = eos
