"""TransformerDecoder (attention decoder used as a beam-search scorer) on the MI355X.

Mirrors espnet2/asr/decoder/transformer_decoder.py:393-468 (constructor) and :191-311
(`forward_one_step`, `score`, `batch_score`), with the reference's state-dict keys
(`embed.0`, `decoders.N.{self_attn,src_attn,feed_forward,norm1,norm2,norm3}`, `after_norm`,
`output_layer`).  The torch.nn layers are parameter containers only.

The arithmetic is csrc/decoder.hip + csrc/gemm.hip + csrc/norm.hip, driven per search step by
csrc/search.hip (`em_search_steps`); `pack()` repacks the reference-layout parameters once
(q|k|v rows concatenated for self-attention, k|v for source attention, absolute sinusoid table).
The scorer-interface methods (`batch_score`, `score`, `select_state`) of the reference are
fulfilled inside the fused device search (espnet_amd/nets/batch_beam_search.py) rather than
through per-step Python calls.
"""
import ctypes as C
import math
from typing import List

import torch

from espnet_amd import lib as L
from espnet_amd.asr.encoder.conformer_encoder import LayerNorm, _PositionwiseFeedForward


def abs_pos_table(length: int, d: int) -> torch.Tensor:
    """PositionalEncoding.extend_pe (transformer/embedding.py:56-79), same fp32 torch ops."""
    pe = torch.zeros(length, d)
    position = torch.arange(0, length, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


class _MultiHeadedAttention(torch.nn.Module):
    def __init__(self, n_head, n_feat):
        super().__init__()
        assert n_feat % n_head == 0
        self.d_k, self.h = n_feat // n_head, n_head
        self.linear_q = torch.nn.Linear(n_feat, n_feat)
        self.linear_k = torch.nn.Linear(n_feat, n_feat)
        self.linear_v = torch.nn.Linear(n_feat, n_feat)
        self.linear_out = torch.nn.Linear(n_feat, n_feat)


class _DecoderLayer(torch.nn.Module):
    """Parameters of transformer/decoder_layer.py:44-71."""

    def __init__(self, size, heads, ff):
        super().__init__()
        self.self_attn = _MultiHeadedAttention(heads, size)
        self.src_attn = _MultiHeadedAttention(heads, size)
        self.feed_forward = _PositionwiseFeedForward(size, ff)
        self.norm1 = LayerNorm(size)
        self.norm2 = LayerNorm(size)
        self.norm3 = LayerNorm(size)


class _PosEncPlaceholder(torch.nn.Module):
    """`embed.1` of the reference (PositionalEncoding, no parameters/buffers)."""


class TransformerDecoder(torch.nn.Module):
    def __init__(self, vocab_size: int, encoder_output_size: int, attention_heads: int = 4,
                 linear_units: int = 2048, num_blocks: int = 6, dropout_rate: float = 0.1,
                 positional_dropout_rate: float = 0.1, self_attention_dropout_rate: float = 0.0,
                 src_attention_dropout_rate: float = 0.0, input_layer: str = "embed",
                 use_output_layer: bool = True, pos_enc_class=None, normalize_before: bool = True,
                 concat_after: bool = False, layer_drop_rate: float = 0.0, qk_norm: bool = False,
                 use_flash_attn: bool = True, gradient_checkpoint_layers: List[int] = [],
                 compute_dtype: str = "bfloat16"):
        super().__init__()
        if input_layer != "embed" or not use_output_layer or not normalize_before or concat_after or qk_norm:
            raise NotImplementedError("outside the MI355X TransformerDecoder fast path")
        d = encoder_output_size
        self.vocab_size, self.d, self.heads = vocab_size, d, attention_heads
        self.linear_units, self.num_blocks = linear_units, num_blocks
        self.compute_dtype = compute_dtype
        self.embed = torch.nn.Sequential(torch.nn.Embedding(vocab_size, d), _PosEncPlaceholder())
        self.normalize_before = normalize_before
        self.after_norm = LayerNorm(d)
        self.output_layer = torch.nn.Linear(d, vocab_size)
        self.decoders = torch.nn.ModuleList(
            [_DecoderLayer(d, attention_heads, linear_units) for _ in range(num_blocks)])
        self._packed = None

    def invalidate(self):
        self._packed = None

    @property
    def em_dtype(self) -> int:
        return L.DTYPES[self.compute_dtype]

    @property
    def act_dtype(self) -> torch.dtype:
        return torch.bfloat16 if self.em_dtype == L.EM_BF16 else torch.float32

    def pack(self, device, pe_len: int = 1024):
        dev = torch.device(device)
        act = self.act_dtype
        keep = []

        def A(t):
            t = t.detach().to(torch.float32).contiguous().to(act).to(dev)
            keep.append(t)
            return t

        def F(t):
            t = t.detach().to(torch.float32).contiguous().to(dev)
            keep.append(t)
            return t

        w = L.EmDecoderWeights()
        w.d, w.heads, w.ff, w.num_blocks = self.d, self.heads, self.linear_units, self.num_blocks
        w.vocab, w.pe_len = self.vocab_size, pe_len
        top = dict(embed=F(self.embed[0].weight), pe=F(abs_pos_table(pe_len, self.d)),
                   after_norm_g=F(self.after_norm.weight), after_norm_b=F(self.after_norm.bias),
                   out_w=A(self.output_layer.weight), out_b=F(self.output_layer.bias))
        for k, v in top.items():
            setattr(w, k, v.data_ptr())
        layers = (L.EmDecoderLayer * self.num_blocks)()
        for i, l in enumerate(self.decoders):
            sa, ca, ff = l.self_attn, l.src_attn, l.feed_forward
            lt = dict(
                norm1_g=F(l.norm1.weight), norm1_b=F(l.norm1.bias),
                norm2_g=F(l.norm2.weight), norm2_b=F(l.norm2.bias),
                norm3_g=F(l.norm3.weight), norm3_b=F(l.norm3.bias),
                self_wqkv=A(torch.cat([sa.linear_q.weight, sa.linear_k.weight, sa.linear_v.weight], 0)),
                self_bqkv=F(torch.cat([sa.linear_q.bias, sa.linear_k.bias, sa.linear_v.bias], 0)),
                self_wout=A(sa.linear_out.weight), self_bout=F(sa.linear_out.bias),
                src_wq=A(ca.linear_q.weight), src_bq=F(ca.linear_q.bias),
                src_wkv=A(torch.cat([ca.linear_k.weight, ca.linear_v.weight], 0)),
                src_bkv=F(torch.cat([ca.linear_k.bias, ca.linear_v.bias], 0)),
                src_wout=A(ca.linear_out.weight), src_bout=F(ca.linear_out.bias),
                w1=A(ff.w_1.weight), b1=F(ff.w_1.bias), w2=A(ff.w_2.weight), b2=F(ff.w_2.bias))
            for k, v in lt.items():
                setattr(layers[i], k, v.data_ptr())
        w.layers = C.cast(layers, C.POINTER(L.EmDecoderLayer))
        self._packed = dict(w=w, layers=layers, keep=keep, device=dev, dtype=self.em_dtype,
                            pe_len=pe_len)
        return self._packed

    def ensure_packed(self, device, pe_len: int):
        p = self._packed
        if p is None or p["device"] != device or p["dtype"] != self.em_dtype or p["pe_len"] < pe_len:
            p = self.pack(device, max(1024, pe_len))
        return p
