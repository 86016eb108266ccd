"""TransformerLM as a beam-search scorer on the MI355X.

Mirrors espnet2/lm/transformer_lm.py:12-137 (constructor keywords, state-dict keys `embed`,
`encoder.embed.{0,1}`, `encoder.encoders.N.{self_attn,feed_forward,norm1,norm2}`,
`encoder.after_norm`, `decoder`) and espnet2/lm/espnet_model.py:13-22 (`ESPnetLanguageModel` with
`.lm`).  Inside the fused device search the LM step runs in csrc/search.hip `lm_step` over the same
token-tree K/V cache mechanism as the attention decoder; `batch_score` / `score` / `select_state` are the
reference's per-step scorer interface on the same code (`em_lm_step`).  The torch.nn layers are parameter
containers only.
"""
import ctypes as C
from typing import Optional

import torch

from espnet_amd import lib as L
from espnet_amd.asr.decoder.transformer_decoder import abs_pos_table
from espnet_amd.asr.encoder.conformer_encoder import LayerNorm, _PositionwiseFeedForward
from espnet_amd.nets.scorer_interface import BatchScorerInterface


class _MultiHeadedAttention(torch.nn.Module):
    def __init__(self, n_feat):
        super().__init__()
        self.linear_q = torch.nn.Linear(n_feat, n_feat)
        self.linear_k = torch.nn.Linear(n_feat, n_feat)
        self.linear_v = torch.nn.Linear(n_feat, n_feat)
        self.linear_out = torch.nn.Linear(n_feat, n_feat)


class _EncoderLayer(torch.nn.Module):
    """Parameters of transformer/encoder_layer.py:17-63."""

    def __init__(self, size, ff):
        super().__init__()
        self.self_attn = _MultiHeadedAttention(size)
        self.feed_forward = _PositionwiseFeedForward(size, ff)
        self.norm1 = LayerNorm(size)
        self.norm2 = LayerNorm(size)


class _Encoder(torch.nn.Module):
    """Parameters of transformer/encoder.py Encoder(input_layer="linear") (:132-139, :200-330)."""

    def __init__(self, idim, d, ff, layers):
        super().__init__()
        # Linear, LayerNorm(eps 1e-5), Dropout, ReLU, pos-enc: indices 0 and 1 carry parameters
        self.embed = torch.nn.Sequential(torch.nn.Linear(idim, d), torch.nn.LayerNorm(d),
                                         torch.nn.Identity(), torch.nn.ReLU(), torch.nn.Identity())
        self.encoders = torch.nn.ModuleList([_EncoderLayer(d, ff) for _ in range(layers)])
        self.after_norm = LayerNorm(d)


class TransformerLM(torch.nn.Module, BatchScorerInterface):
    def __init__(self, vocab_size: int, pos_enc: Optional[str] = None, embed_unit: int = 128,
                 att_unit: int = 256, head: int = 2, unit: int = 1024, layer: int = 4,
                 dropout_rate: float = 0.1, positional_dropout_rate: float = 0.1,
                 attention_dropout_rate: float = 0.1, compute_dtype: str = "bfloat16"):
        super().__init__()
        if pos_enc not in (None, "sinusoidal"):
            raise ValueError(f"unknown pos-enc option: {pos_enc}")
        if att_unit % 64 or att_unit // head not in (32, 64) or unit % 64:
            raise NotImplementedError("MI355X TransformerLM fast path: att_unit % 64 == 0, d_k in {32, 64}, unit % 64 == 0")
        self.vocab_size, self.pos_enc = vocab_size, pos_enc
        self.embed_unit, self.att_unit, self.head, self.unit, self.layer = embed_unit, att_unit, head, unit, layer
        self.compute_dtype = compute_dtype
        self.embed = torch.nn.Embedding(vocab_size, embed_unit)
        self.encoder = _Encoder(embed_unit, att_unit, unit, layer)
        self.decoder = torch.nn.Linear(att_unit, vocab_size)
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

    def search_key(self):
        return ("transformer", self.att_unit, self.unit, self.layer, self.embed_unit)

    def search_buffers(self, n, V, Lmax, B, cap):
        """name -> shape of the EmSearchBuffers entries this scorer needs (nets/batch_beam_search.py)."""
        dl = self.att_unit
        return dict(lm_e=(n, self.embed_unit), lm_xn=(n, dl), lm_qkv=(n, 3 * dl), lm_ctx=(n, dl), lm_h=(n, self.unit),
                    lm_x=(n, dl), lm_logp=(n, V), lm_k=(self.layer, Lmax, n, dl), lm_v=(self.layer, Lmax, n, dl),
                    run_slm=(n,), end_slm=(B, cap))

    def pack(self, device, pe_len: int = 1024):
        dev = torch.device(device)
        act = torch.bfloat16 if self.em_dtype == L.EM_BF16 else torch.float32
        kmult = 64 if self.em_dtype == L.EM_BF16 else 32
        if self.embed_unit % kmult:
            raise NotImplementedError(f"embed_unit must be a multiple of {kmult} in {self.compute_dtype} mode")
        keep = []

        def A(t):
            t = t.detach().to(torch.float32).contiguous().to(act).to(dev)
            keep.append(t)
            return t

        def F(t):
            t = t.detach().to(torch.float32).contiguous().to(dev)
            keep.append(t)
            return t

        w = L.EmLmWeights()
        w.d, w.heads, w.ff, w.num_blocks = self.att_unit, self.head, self.unit, self.layer
        w.vocab, w.embed_unit = self.vocab_size, self.embed_unit
        e = self.encoder
        top = dict(embed=F(self.embed.weight), in_w=A(e.embed[0].weight), in_b=F(e.embed[0].bias),
                   in_ln_g=F(e.embed[1].weight), in_ln_b=F(e.embed[1].bias),
                   after_norm_g=F(e.after_norm.weight), after_norm_b=F(e.after_norm.bias),
                   out_w=A(self.decoder.weight), out_b=F(self.decoder.bias))
        if self.pos_enc == "sinusoidal":
            top["pe"] = F(abs_pos_table(pe_len, self.att_unit))
        for k, v in top.items():
            setattr(w, k, v.data_ptr())
        layers = (L.EmLmLayer * self.layer)()
        for i, l in enumerate(e.encoders):
            sa, ff = l.self_attn, l.feed_forward
            lt = dict(norm1_g=F(l.norm1.weight), norm1_b=F(l.norm1.bias), norm2_g=F(l.norm2.weight),
                      norm2_b=F(l.norm2.bias),
                      wqkv=A(torch.cat([sa.linear_q.weight, sa.linear_k.weight, sa.linear_v.weight], 0)),
                      bqkv=F(torch.cat([sa.linear_q.bias, sa.linear_k.bias, sa.linear_v.bias], 0)),
                      wout=A(sa.linear_out.weight), bout=F(sa.linear_out.bias),
                      w1=A(ff.w_1.weight), b1=F(ff.w_1.bias), w2=A(ff.w_2.weight), b2=F(ff.w_2.bias))
            for k, v in lt.items():
                setattr(layers[i], k, v.data_ptr())
        w.layers = C.cast(layers, C.POINTER(L.EmLmLayer))
        self._packed = dict(w=w, layers=layers, keep=keep, device=dev, dtype=self.em_dtype, pe_len=pe_len)
        return self._packed

    def ensure_packed(self, device, pe_len: int):
        p = self._packed
        if p is None or p["device"] != device or p["dtype"] != self.em_dtype or p["pe_len"] < pe_len:
            p = self.pack(device, max(1024, pe_len))
        return p

    # ------------------------------------------------------------------ scorer interface (one call per step)
    @torch.no_grad()
    def batch_score(self, ys: torch.Tensor, states, xs: torch.Tensor):
        """transformer_lm.py:103-137.  ys (n, L) int64 prefixes; states list[n] of None or this class's opaque
        cache (self-attention K/V of the prefix, (2, layers, L-1, att_unit) in the compute dtype); xs is only
        used for its device.  Returns (log-probs (n, V) f32, states list[n])."""
        from espnet_amd.lm.step import lm_step

        L.require_gpu(xs, "xs")
        dev = xs.device
        n, Lc = ys.shape
        pos, Lmax = Lc - 1, Lc + 1
        act = torch.bfloat16 if self.em_dtype == L.EM_BF16 else torch.float32
        kv = torch.zeros(2, self.layer, Lmax, n, self.att_unit, dtype=act, device=dev)
        if pos > 0:
            if states is None or any(s is None for s in states):
                raise ValueError("batch_score: a prefix longer than <sos> needs the state of its previous step")
            kv[:, :, :pos] = torch.stack(list(states), 0).permute(1, 2, 3, 0, 4)
        tok = torch.zeros(Lmax, n, dtype=torch.int32, device=dev)
        tok[:Lc] = ys.t().to(device=dev, dtype=torch.int32)
        logp, _ = lm_step(self, dev, n, Lmax, pos, tok, dict(lm_k=kv[0], lm_v=kv[1]))
        out = kv[:, :, :Lc].permute(3, 0, 1, 2, 4)
        return logp, [out[r] for r in range(n)]

    @torch.no_grad()
    def forward(self, input: torch.Tensor, hidden=None):
        """transformer_lm.py:59-75 (AbsLM.forward): input (B, L) int64 -> (logits (B, L, V) f32, None).  Causal
        self-attention makes position j depend on tokens <= j only, so the sequence is fed position by position
        through the step kernels over one K/V cache (inference-only class: no autograd)."""
        from espnet_amd.lm.step import lm_step

        L.require_gpu(input, "input")
        dev = input.device
        n, Lc = input.shape
        act = torch.bfloat16 if self.em_dtype == L.EM_BF16 else torch.float32
        kv = torch.zeros(2, self.layer, Lc + 1, n, self.att_unit, dtype=act, device=dev)
        tok = torch.zeros(Lc + 1, n, dtype=torch.int32, device=dev)
        tok[:Lc] = input.t().to(torch.int32)
        out = torch.empty(n, Lc, self.vocab_size, dtype=torch.float32, device=dev)
        for pos in range(Lc):
            out[:, pos] = lm_step(self, dev, n, Lc + 1, pos, tok, dict(lm_k=kv[0], lm_v=kv[1]), log_softmax=False)[0]
        return out, None

    def score(self, y: torch.Tensor, state, x: torch.Tensor):
        """transformer_lm.py:77-101: one hypothesis."""
        logp, st = self.batch_score(y.unsqueeze(0), [state], x.unsqueeze(0))
        return logp[0], st[0]


class ESPnetLanguageModel(torch.nn.Module):
    """espnet2/lm/espnet_model.py:13-22 (inference attributes only)."""

    def __init__(self, lm: TransformerLM, vocab_size: int, ignore_id: int = 0):
        super().__init__()
        self.lm = lm
        self.sos = self.eos = vocab_size - 1
        self.ignore_id = ignore_id
