"""SequentialRNNLM (LSTM) as a beam-search scorer on the MI355X.

Mirrors espnet2/lm/seq_rnn_lm.py:14-177 (constructor keywords; state-dict keys `encoder.weight`,
`rnn.{weight,bias}_{ih,hh}_l{k}`, `decoder.{weight,bias}`).  `batch_score` (:140-177) is fulfilled inside
the fused device search (csrc/search.hip `lstm_lm_step`): per step the last token's embedding and the
PARENT hypothesis' (h, c) go through the LSTM cells; the logits' log-softmax is summed with the other
full scorers in the pre-beam kernel.  Hidden sizes are zero-padded to the GEMM K step at pack time
(unit = 650 is the class default), which leaves every product unchanged.
The torch.nn layers are parameter containers only.
"""
import ctypes as C
from typing import Optional

import torch

from espnet_amd import lib as L


class SequentialRNNLM(torch.nn.Module):
    def __init__(self, vocab_size: int, unit: int = 650, nhid: Optional[int] = None, nlayers: int = 2,
                 dropout_rate: float = 0.0, tie_weights: bool = False, rnn_type: str = "lstm",
                 ignore_id: int = 0, compute_dtype: str = "bfloat16"):
        super().__init__()
        if rnn_type.upper() != "LSTM":
            raise NotImplementedError(f"rnn_type={rnn_type!r}: only the LSTM (the class default) is on the MI355X path")
        nhid = unit if nhid is None else nhid
        self.vocab_size, self.unit, self.nhid, self.nlayers = vocab_size, unit, nhid, nlayers
        self.compute_dtype = compute_dtype
        self.encoder = torch.nn.Embedding(vocab_size, unit, padding_idx=ignore_id)
        self.rnn = torch.nn.LSTM(unit, nhid, nlayers, dropout=dropout_rate, batch_first=True)
        self.decoder = torch.nn.Linear(nhid, vocab_size)
        if tie_weights:
            if nhid != unit:
                raise ValueError("When using the tied flag, nhid must be equal to emsize")
            self.decoder.weight = self.encoder.weight
        self._packed = None

    @property
    def em_dtype(self) -> int:
        return L.DTYPES[self.compute_dtype]

    def invalidate(self):
        self._packed = None

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self.invalidate()
        return r

    @staticmethod
    def _pad(v: int) -> int:
        return (v + 63) // 64 * 64

    def search_key(self):
        return ("seq_rnn", self.unit, self.nhid, self.nlayers)

    def search_buffers(self, n, V, Lmax, B, cap):
        d, eu = self._pad(self.nhid), self._pad(self.unit)
        return dict(lm_e=(n, eu), lm_logp=(n, V), rnn_hs=(3, self.nlayers, n, d), rnn_cs=(3, self.nlayers, n, d),
                    rnn_hin=(self.nlayers, n, d), rnn_gates=(n, 4 * self.nhid), run_slm=(n,), end_slm=(B, cap))

    def pack(self, device, pe_len: int = 0):
        dev = torch.device(device)
        act = torch.bfloat16 if self.em_dtype == L.EM_BF16 else torch.float32
        d, eu, nh = self._pad(self.nhid), self._pad(self.unit), self.nhid
        keep = []

        def padk(t, k):  # zero-pad the contraction dimension
            out = torch.zeros(t.size(0), k, dtype=torch.float32)
            out[:, : t.size(1)] = t.detach().to(torch.float32)
            return out

        def A(t):
            t = t.contiguous().to(act).to(dev)
            keep.append(t)
            return t

        def F(t):
            t = t.detach().to(torch.float32).contiguous().to(dev)
            keep.append(t)
            return t

        w = L.EmLmWeights()
        w.kind, w.d, w.nhid, w.embed_unit, w.num_blocks, w.vocab = L.EM_LM_LSTM, d, nh, eu, self.nlayers, self.vocab_size
        w.heads = w.ff = 0
        top = dict(embed=F(padk(self.encoder.weight, eu)), out_w=A(padk(self.decoder.weight, d)),
                   out_b=F(self.decoder.bias))
        for k, v in top.items():
            setattr(w, k, v.data_ptr())
        layers = (L.EmRnnLayer * self.nlayers)()
        for l in range(self.nlayers):
            w_ih, w_hh = getattr(self.rnn, f"weight_ih_l{l}"), getattr(self.rnn, f"weight_hh_l{l}")
            b = getattr(self.rnn, f"bias_ih_l{l}").detach().float() + getattr(self.rnn, f"bias_hh_l{l}").detach().float()
            lt = dict(w_ih=A(padk(w_ih, eu if l == 0 else d)), w_hh=A(padk(w_hh, d)), bias=F(b))
            for k, v in lt.items():
                setattr(layers[l], k, v.data_ptr())
        w.rnn = C.cast(layers, C.POINTER(L.EmRnnLayer))
        self._packed = dict(w=w, layers=layers, keep=keep, device=dev, dtype=self.em_dtype, pe_len=1 << 30)
        return self._packed

    def ensure_packed(self, device, pe_len: int = 0):
        p = self._packed
        if p is None or p["device"] != device or p["dtype"] != self.em_dtype:
            p = self.pack(device)
        return p
