"""ORACLE (test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this; never shipped, never on the product path, never measured as the GPU result).

CPU fp32 restatement of the reference's streaming (contextual block) Conformer encoder step,
written functionally over the reference's flat `state_dict` (keys `embed.*`, `encoders.N.*`,
`after_norm.*` of the encoder module):

  * `CBEncoderOracle.forward_infer`  ContextualBlockConformerEncoder.forward_infer
        (espnet2/asr/encoder/contextual_block_conformer_encoder.py:386-600): waveform-feature
        buffering before / after the 4x subsampling, block assembly with context slots,
        output stitching, carried state
  * `cb_layer_infer`                 ContextualBlockEncoderLayer.forward_infer
        (espnet2/legacy/nets/pytorch_backend/conformer/contextual_block_encoder_layer.py:197-310)
  * `subsampling_wo_posenc`          Conv2dSubsamplingWOPosEnc.forward
        (espnet2/legacy/nets/pytorch_backend/transformer/subsampling_without_posenc.py:44-62)
  * `stream_pos_enc`                 StreamPositionalEncoding.forward
        (espnet2/legacy/nets/pytorch_backend/transformer/embedding.py:376-389)

Quirks kept: the feed-forward blocks use ReLU (the encoder does not pass `activation` to
PositionwiseFeedForward, contextual_block_conformer_encoder.py:148-154) while the convolution
module uses Swish; the depthwise convolution runs over all block_size+2 slots including the two
context slots; row 0 of the attention mask is fully masked (its output is the all-zero context,
later overwritten by the context propagation).

Pinned against the reference class itself: tests/golden/stream_*.npz (tests/golden/make_golden.py
feeds the reference encoder chunk by chunk) via tests/test_oracle_golden.py.
"""
import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from oracle.conformer import LN_EPS, conv_module

Tensor = torch.Tensor


def _ln(x, sd, pre):
    return F.layer_norm(x, (x.size(-1),), sd[pre + "weight"], sd[pre + "bias"], LN_EPS)


def _lin(x, sd, pre):
    return F.linear(x, sd[pre + "weight"], sd[pre + "bias"])


def pos_table(length: int, d: int) -> Tensor:
    pe = torch.zeros(length, d)
    position = torch.arange(0, length, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def stream_pos_enc(x: Tensor, start: int, pe: Tensor) -> Tensor:
    """embedding.py:376-389: x * sqrt(d) + pe[start : start + L]."""
    return x * math.sqrt(x.size(-1)) + pe[start : start + x.size(-2)]


def subsampling_wo_posenc(sd, x: Tensor, pre: str = "embed.") -> Tensor:
    """subsampling_without_posenc.py:44-62 (kernels [3,3], strides [2,2]); x (1, t, idim)."""
    x = x.unsqueeze(1)
    x = F.relu(F.conv2d(x, sd[pre + "conv.0.weight"], sd[pre + "conv.0.bias"], stride=2))
    x = F.relu(F.conv2d(x, sd[pre + "conv.2.weight"], sd[pre + "conv.2.bias"], stride=2))
    b, c, t, f = x.size()
    return _lin(x.transpose(1, 2).contiguous().view(b, t, c * f), sd, pre + "out.")


def plain_mha(sd, x: Tensor, mask: Optional[Tensor], pre: str, h: int) -> Tensor:
    """MultiHeadedAttention.forward (transformer/attention.py:77-151,263-265); x (n, L, d),
    mask (n, L, L) 1 = attend."""
    n, L, d = x.shape
    dk = d // h
    q = _lin(x, sd, pre + "linear_q.").view(n, L, h, dk).transpose(1, 2)
    k = _lin(x, sd, pre + "linear_k.").view(n, L, h, dk).transpose(1, 2)
    v = _lin(x, sd, pre + "linear_v.").view(n, L, h, dk).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
    if mask is not None:
        m = mask.unsqueeze(1).eq(0)
        scores = scores.masked_fill(m, torch.finfo(scores.dtype).min)
        attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
    else:
        attn = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(attn, v).transpose(1, 2).contiguous().view(n, L, d)
    return _lin(ctx, sd, pre + "linear_out.")


def cb_layer_infer(sd, x: Tensor, mask: Optional[Tensor], pre: str, h: int) -> Tensor:
    """The arithmetic of ContextualBlockEncoderLayer.forward_infer (:229-283) on x (n_blk, L, d);
    the context propagation (:292-304) is done by the caller."""
    t = _ln(x, sd, pre + "norm_ff_macaron.")
    x = x + 0.5 * _lin(torch.relu(_lin(t, sd, pre + "feed_forward_macaron.w_1.")), sd,
                       pre + "feed_forward_macaron.w_2.")
    x = x + plain_mha(sd, _ln(x, sd, pre + "norm1."), mask, pre + "self_attn.", h)
    x = x + conv_module(sd, _ln(x, sd, pre + "norm_conv."), pre + "conv_module.")
    t = _ln(x, sd, pre + "norm2.")
    x = x + 0.5 * _lin(torch.relu(_lin(t, sd, pre + "feed_forward.w_1.")), sd, pre + "feed_forward.w_2.")
    return _ln(x, sd, pre + "norm_final.")


class CBEncoderOracle:
    def __init__(self, sd: Dict[str, Tensor], heads: int, num_blocks: int, block_size: int = 40,
                 hop_size: int = 16, look_ahead: int = 16, subsample: int = 4):
        self.sd, self.h, self.nl = sd, heads, num_blocks
        self.block_size, self.hop_size, self.look_ahead = block_size, hop_size, look_ahead
        self.subsample = subsample
        self.d = sd["after_norm.weight"].numel()
        self.pe = pos_table(5000, self.d)

    def _layers(self, x: Tensor, mask, past_ctx, short: bool):
        """self.encoders(xs_chunk, mask, True, past_ctx[, None, short]); x (n_blk, L, d).
        Returns (x, next_ctx (layers, d) or None)."""
        next_ctx = None if short else torch.zeros(self.nl, self.d)
        for l in range(self.nl):
            x = cb_layer_infer(self.sd, x, mask, f"encoders.{l}.", self.h)
            if not short:  # contextual_block_encoder_layer.py:292-304
                x = x.clone()
                x[0, 0] = x[0, -1] if past_ctx is None else past_ctx[l]
                if x.size(0) > 1:
                    x[1:, 0] = x[:-1, -1]
                next_ctx[l] = x[-1, -1]
        return x, next_ctx

    def forward_infer(self, xs: Tensor, state: Optional[dict], is_final: bool
                      ) -> Tuple[Tensor, Optional[dict]]:
        """xs (t, idim) new feature frames.  Returns (ys (t_out, d), next state)."""
        bs, hs, la, sub = self.block_size, self.hop_size, self.look_ahead, self.subsample
        st = state or dict(prev_addin=None, buf_before=None, buf_after=None, n_proc=0, past_ctx=None)
        prev_addin, buf_after, n_proc, past_ctx = st["prev_addin"], st["buf_after"], st["n_proc"], st["past_ctx"]
        if st["buf_before"] is not None:
            xs = torch.cat([st["buf_before"], xs], dim=0)
        if is_final:
            buf_before = None
        else:
            n_samples = xs.size(0) // sub - 1
            if n_samples < 2:  # :424-438
                return xs.new_zeros(0, self.d), dict(st, buf_before=xs)
            n_res = xs.size(0) % sub + sub * 2
            buf_before = xs[xs.size(0) - n_res:]
            xs = xs[: n_samples * sub]
        x = subsampling_wo_posenc(self.sd, xs.unsqueeze(0))[0]
        if buf_after is not None:
            x = torch.cat([buf_after, x], dim=0)
        total = x.size(0)
        if is_final:
            past_size = bs - hs - la
            block_num = math.ceil(float(total - past_size - la) / float(hs))
            buf_after = None
        else:
            if total <= bs:  # :474-487
                return x.new_zeros(0, self.d), dict(prev_addin=prev_addin, buf_before=buf_before,
                                                    buf_after=x, n_proc=n_proc, past_ctx=past_ctx)
            overlap = bs - hs
            block_num = max(0, total - overlap) // hs
            res = total - hs * block_num
            buf_after = x[total - res:]
            x = x[: block_num * hs + overlap]
        if n_proc == 0 and total <= bs and is_final:  # short utterance :496-505
            y, _ = self._layers(stream_pos_enc(x, 0, self.pe).unsqueeze(0), None, None, True)
            return _ln(y[0], self.sd, "after_norm."), None
        chunks = x.new_zeros(block_num, bs + 2, self.d)
        for i in range(block_num):  # :512-536
            cur = i * hs
            clen = min(bs, total - cur)
            addin = stream_pos_enc(x[cur : cur + clen].mean(0, keepdim=True), i + n_proc, self.pe)
            if prev_addin is None:
                prev_addin = addin
            chunks[i, 0] = prev_addin[0]
            chunks[i, -1] = addin[0]
            chunks[i, 1 : clen + 1] = stream_pos_enc(x[cur : cur + clen], cur + hs * n_proc, self.pe)
            prev_addin = addin
        mask = x.new_zeros(block_num, bs + 2, bs + 2)
        mask[:, 1:, : bs + 1] = 1  # :539-544
        ys_chunk, past_ctx = self._layers(chunks, mask, past_ctx, False)
        ys_chunk = ys_chunk[:, 1 : bs + 1]
        offset = bs - la - hs
        if is_final:
            y_len = x.size(0) if n_proc == 0 else x.size(0) - offset
        else:
            y_len = block_num * hs + (offset if n_proc == 0 else 0)
        ys = x.new_zeros(y_len, self.d)
        if n_proc == 0:
            ys[:offset] = ys_chunk[0, :offset]
        for i in range(block_num):  # :565-576
            cur = i * hs + (offset if n_proc == 0 else 0)
            clen = min(bs - offset, y_len - cur) if (i == block_num - 1 and is_final) else hs
            ys[cur : cur + clen] = ys_chunk[i, offset : offset + clen]
        ys = _ln(ys, self.sd, "after_norm.")
        if is_final:
            return ys, None
        return ys, dict(prev_addin=prev_addin, buf_before=buf_before, buf_after=buf_after,
                        n_proc=n_proc + block_num, past_ctx=past_ctx)
