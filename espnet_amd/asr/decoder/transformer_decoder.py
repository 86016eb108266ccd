"""TransformerDecoder (attention decoder used as a beam-search scorer) on the MI355X.

Mirrors espnet2/asr/decoder/transformer_decoder.py:393-468 (constructor) and :191-311
(`forward_one_step`, `score`, `batch_score`), with the reference's state-dict keys
(`embed.0`, `decoders.N.{self_attn,src_attn,feed_forward,norm1,norm2,norm3}`, `after_norm`,
`output_layer`).  The torch.nn layers are parameter containers only.

Round-1 status: parameter tree + checkpoint compatibility; the K/V-cached decoder-step kernels
 land with the beam-search row.
"""
from typing import List

import torch

from espnet_amd.asr.encoder.conformer_encoder import LayerNorm, _PositionwiseFeedForward


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
