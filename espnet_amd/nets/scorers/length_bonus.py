"""LengthBonus (espnet2/legacy/nets/scorers/length_bonus.py:10-58): a constant 1.0 per emitted
token, weighted by `penalty`.  Folded into the candidate totals on the device (csrc/search.hip)."""


class LengthBonus:
    def __init__(self, n_vocab: int):
        self.n = n_vocab
