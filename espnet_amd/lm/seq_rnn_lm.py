"""SequentialRNNLM (LSTM / GRU / tanh- / relu-RNN) as a beam-search scorer on the MI355X.

Mirrors espnet2/lm/seq_rnn_lm.py:14-177 (constructor keywords; state-dict keys `encoder.weight`,
`rnn.{weight,bias}_{ih,hh}_l{k}`, `decoder.{weight,bias}`).  `batch_score` (:140-177) is fulfilled inside
the fused device search (csrc/search.hip `rnn_lm_step`) and, as the reference's per-step call, by `batch_score`
below on the same code (`em_lm_step`): per step the last token's embedding and the
PARENT hypothesis' state ((h, c) for the LSTM, h otherwise, :80-89) go through the recurrent cells; the logits' log-softmax is summed with the other
full scorers in the pre-beam kernel.  Hidden sizes are zero-padded to the GEMM K step at pack time
(unit = 650 is the class default), which leaves every product unchanged.
The torch.nn layers are parameter containers only.
"""
import ctypes as C
from typing import Optional

import torch

from espnet_amd import lib as L
from espnet_amd.nets.scorer_interface import BatchScorerInterface


_KINDS = {"LSTM": L.EM_LM_LSTM, "GRU": L.EM_LM_GRU, "RNN_TANH": L.EM_LM_RNN_TANH, "RNN_RELU": L.EM_LM_RNN_RELU}


class SequentialRNNLM(torch.nn.Module, BatchScorerInterface):
    def __init__(self, vocab_size: int, unit: int = 650, nhid: Optional[int] = None, nlayers: int = 2,
                 dropout_rate: float = 0.0, tie_weights: bool = False, rnn_type: str = "lstm",
                 ignore_id: int = 0, compute_dtype: str = "bfloat16"):
        super().__init__()
        rnn_type = rnn_type.upper()
        if rnn_type not in _KINDS:  # seq_rnn_lm.py:48-52
            raise ValueError("An invalid option for `--model` was supplied, "
                             "options are ['LSTM', 'GRU', 'RNN_TANH' or 'RNN_RELU']")
        nhid = unit if nhid is None else nhid
        self.rnn_type = rnn_type
        self.vocab_size, self.unit, self.nhid, self.nlayers = vocab_size, unit, nhid, nlayers
        self.compute_dtype = compute_dtype
        self.encoder = torch.nn.Embedding(vocab_size, unit, padding_idx=ignore_id)
        if rnn_type in ("LSTM", "GRU"):
            self.rnn = getattr(torch.nn, rnn_type)(unit, nhid, nlayers, dropout=dropout_rate, batch_first=True)
        else:
            self.rnn = torch.nn.RNN(unit, nhid, nlayers, nonlinearity=rnn_type[4:].lower(), dropout=dropout_rate,
                                    batch_first=True)
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
        return ("seq_rnn", self.rnn_type, self.unit, self.nhid, self.nlayers)

    @property
    def gate_blocks(self) -> int:
        """Gate blocks of nhid rows per layer on the device (EmRnnLayer): the GRU carries 4, see `pack`."""
        return 1 if self.rnn_type.startswith("RNN") else 4

    def search_buffers(self, n, V, Lmax, B, cap):
        d, eu = self._pad(self.nhid), self._pad(self.unit)
        return dict(lm_e=(n, eu), lm_logp=(n, V), rnn_hs=(3, self.nlayers, n, d), rnn_cs=(3, self.nlayers, n, d),
                    rnn_hin=(self.nlayers, n, d), rnn_gates=(n, self.gate_blocks * self.nhid), run_slm=(n,), end_slm=(B, cap))

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
        w.kind, w.d, w.nhid, w.embed_unit, w.num_blocks, w.vocab = _KINDS[self.rnn_type], d, nh, eu, self.nlayers, self.vocab_size
        w.heads = w.ff = 0
        top = dict(embed=F(padk(self.encoder.weight, eu)), out_w=A(padk(self.decoder.weight, d)),
                   out_b=F(self.decoder.bias))
        for k, v in top.items():
            setattr(w, k, v.data_ptr())
        layers = (L.EmRnnLayer * self.nlayers)()
        for l in range(self.nlayers):
            w_ih, w_hh = (getattr(self.rnn, f"weight_{k}_l{l}").detach().float().cpu() for k in ("ih", "hh"))
            b_ih, b_hh = (getattr(self.rnn, f"bias_{k}_l{l}").detach().float().cpu() for k in ("ih", "hh"))
            if self.rnn_type == "GRU":
                # r | z | n -> r | z | n_x | n_h: the candidate gate's input and hidden parts stay separate
                # (n = tanh(W_in x + b_in + r * (W_hn h + b_hn)), torch.nn.GRU), zero blocks keep them apart
                zi, zh = torch.zeros(nh, w_ih.size(1)), torch.zeros(nh, w_hh.size(1))
                w_ih = torch.cat([w_ih, zi])
                w_hh = torch.cat([w_hh[: 2 * nh], zh, w_hh[2 * nh:]])
                b = torch.cat([b_ih[: 2 * nh] + b_hh[: 2 * nh], b_ih[2 * nh:], b_hh[2 * nh:]])
            else:
                b = b_ih + b_hh
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

    # ------------------------------------------------------------------ scorer interface (one call per step)
    @torch.no_grad()
    def batch_score(self, ys: torch.Tensor, states, xs: torch.Tensor):
        """seq_rnn_lm.py:140-177.  ys (n, L) int64 prefixes (only the last token is consumed, :92); states
        list[n] of None (zero state, :155-156) or this class's opaque state: (h in the compute dtype, f32 master
        state), each (nlayers, padded nhid).  Returns (log-probs (n, V) f32, states list[n])."""
        from espnet_amd.lm.step import lm_step

        L.require_gpu(xs, "xs")
        dev = xs.device
        n = ys.size(0)
        act = torch.bfloat16 if self.em_dtype == L.EM_BF16 else torch.float32
        d = self._pad(self.nhid)
        hs = torch.zeros(3, self.nlayers, n, d, dtype=act, device=dev)
        cs = torch.zeros(3, self.nlayers, n, d, dtype=torch.float32, device=dev)
        first = states is None or states[0] is None
        i = 0 if first else 1  # step i reads ring slot (i - 1) % 3 and writes slot i % 3
        if not first:
            hs[0] = torch.stack([s[0] for s in states], 1)
            cs[0] = torch.stack([s[1] for s in states], 1)
        tok = torch.zeros(3, n, dtype=torch.int32, device=dev)
        tok[i] = ys[:, -1].to(device=dev, dtype=torch.int32)
        logp, _ = lm_step(self, dev, n, 3, i, tok, dict(rnn_hs=hs, rnn_cs=cs))
        return logp, [(hs[i, :, r], cs[i, :, r]) for r in range(n)]

    @torch.no_grad()
    def forward(self, input: torch.Tensor, hidden=None):
        """seq_rnn_lm.py:91-116 (AbsLM.forward): input (B, L) int64, hidden = None or the list of per-row states
        `batch_score` returns -> (logits (B, L, V) f32, hidden), one recurrent step per position."""
        from espnet_amd.lm.step import lm_step

        L.require_gpu(input, "input")
        dev = input.device
        n, Lc = input.shape
        act = torch.bfloat16 if self.em_dtype == L.EM_BF16 else torch.float32
        d = self._pad(self.nhid)
        out = torch.empty(n, Lc, self.vocab_size, dtype=torch.float32, device=dev)
        states = hidden
        for pos in range(Lc):
            hs = torch.zeros(3, self.nlayers, n, d, dtype=act, device=dev)
            cs = torch.zeros(3, self.nlayers, n, d, dtype=torch.float32, device=dev)
            i = 0 if states is None else 1
            if states is not None:
                hs[0] = torch.stack([s[0] for s in states], 1)
                cs[0] = torch.stack([s[1] for s in states], 1)
            tok = torch.zeros(3, n, dtype=torch.int32, device=dev)
            tok[i] = input[:, pos].to(torch.int32)
            out[:, pos] = lm_step(self, dev, n, 3, i, tok, dict(rnn_hs=hs, rnn_cs=cs), log_softmax=False)[0]
            states = [(hs[i, :, r], cs[i, :, r]) for r in range(n)]
        return out, states

    def score(self, y: torch.Tensor, state, x: torch.Tensor):
        """seq_rnn_lm.py:118-138: one hypothesis."""
        logp, st = self.batch_score(y.unsqueeze(0), [state], x.unsqueeze(0))
        return logp[0], st[0]
