"""ORACLE (test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this; it is never shipped, never on the product path and never the thing measured
as the GPU result).

CPU fp32 restatement of the reference's ASR encoder-side hot path, written functionally over a
flat reference `state_dict` (the reference's own checkpoint keys).  Arithmetic
primitives are PyTorch-CPU ATen ops (`torch.stft`, `F.conv2d`, `F.layer_norm`, `matmul`), i.e. the
same pinned third-party dependency (torch 2.10.0) the reference itself calls.

Pinned against the reference: tests/test_oracle_golden.py compares every function below with the
fixtures under tests/golden/ that tests/golden/make_golden.py produced by running the reference
classes themselves (features, every block output of a tiny model, encoder outputs of the small and
large models incl. a ragged batch, CTC argmax ids and G1 tokens).

Each function cites the reference file:line it follows (paths relative to /root/reference).
"""
import math
from itertools import groupby
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LN_EPS = 1e-12  # espnet2/legacy/nets/pytorch_backend/transformer/layer_norm.py:23
BN_EPS = 1e-5  # torch.nn.BatchNorm1d default (conformer/convolution.py:50)


def make_pad_mask(lengths: Tensor, maxlen: int) -> Tensor:
    """True at padded positions. espnet2/legacy/nets/pytorch_backend/nets_utils.py:65."""
    return torch.arange(maxlen)[None, :] >= lengths.to(torch.long)[:, None]


# --------------------------------------------------------------------------- frontend
def stft_power(speech: Tensor, n_fft: int, win_length: int, hop: int) -> Tensor:
    """espnet2/layers/stft.py:76-99 (+ default.py:110 power): hann(win_length) periodic window,
    center=True reflect pad n_fft//2, onesided, un-normalised.  Returns (B, T_f, n_fft//2+1)."""
    window = torch.hann_window(win_length, dtype=speech.dtype)
    spec = torch.stft(speech.float(), n_fft=n_fft, win_length=win_length, hop_length=hop,
                      center=True, window=window, normalized=False, onesided=True,
                      return_complex=True)
    spec = torch.view_as_real(spec).transpose(1, 2)  # (B, T_f, F, 2)
    return spec[..., 0] ** 2 + spec[..., 1] ** 2


def frontend_feats(speech: Tensor, speech_lengths: Tensor, melmat: Tensor, n_fft: int = 512,
                   win_length: Optional[int] = None, hop: int = 160) -> Tuple[Tensor, Tensor]:
    """`ESPnetASRModel._extract_feats` (espnet2/asr/espnet_model.py:450-467) ->
    `DefaultFrontend.forward` (espnet2/asr/frontend/default.py:82-117) ->
    `Stft.forward` (layers/stft.py:48-120) + `LogMel.forward` (layers/log_mel.py:57-84).

    Quirks kept: the STFT runs over the zero-PADDED batch (reflection happens at the padded end,
    short utterances see zeros, stft.py:94); frames >= olens are zeroed in the STFT (stft.py:116)
    -> power 0 -> clamp 1e-10 -> log, and then zeroed again after the log (log_mel.py:77-79).
    """
    win_length = n_fft if win_length is None else win_length
    speech = speech[:, : int(speech_lengths.max())]
    power = stft_power(speech, n_fft, win_length, hop)
    olens = torch.div(speech_lengths + 2 * (n_fft // 2) - n_fft, hop, rounding_mode="trunc") + 1
    pad = make_pad_mask(olens, power.size(1))
    power = power.masked_fill(pad[:, :, None], 0.0)
    mel = torch.matmul(power, melmat)
    mel = torch.clamp(mel, min=1e-10)
    logmel = mel.log().masked_fill(pad[:, :, None], 0.0)
    return logmel, olens


def utterance_mvn(x: Tensor, ilens: Tensor) -> Tensor:
    """espnet2/layers/utterance_mvn.py:45-88 with the task default norm_means=True,
    norm_vars=False: zero the padding, mean over VALID frames, subtract from ALL frames
    (padded frames become -mean: `x -= mean`, :73)."""
    x = x.masked_fill(make_pad_mask(ilens, x.size(1))[:, :, None], 0.0)
    mean = x.sum(dim=1, keepdim=True) / ilens.to(x.dtype).view(-1, 1, 1)
    return x - mean


def global_mvn(x: Tensor, ilens: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """espnet2/layers/global_mvn.py:71-100 (norm_means, norm_vars both true): (x-mean)/std with the
    padded frames forced to zero after each step."""
    pad = make_pad_mask(ilens, x.size(1))[:, :, None]
    x = (x - mean).masked_fill(pad, 0.0)
    return x / std


# --------------------------------------------------------------------------- encoder pieces
def rel_pos_emb(T: int, d: int) -> Tensor:
    """espnet2/legacy/nets/pytorch_backend/transformer/embedding.py:286-332: rows k=0..2T-2 hold
    the sinusoid of relative position T-1-k (even dims sin, odd dims cos)."""
    position = torch.arange(T - 1, -T, -1, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    # the reference builds +pos and -pos tables separately with sin(-1 * position * div_term);
    # fp32 sin/cos are odd/even exactly, and (-1*p)*div == -(p*div) exactly, so this is identical.
    pe = torch.zeros(2 * T - 1, d)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def subsampling_kind(sd: Dict[str, Tensor], pre: str = "encoder.embed.") -> str:
    """Which Conv2dSubsampling variant the state dict holds: "conv2d" (subsampling.py:386), "conv2d6" (:692,
    second conv 5x5 stride 3) or "conv2d8" (:785, a third 3x3 stride-2 conv)."""
    if pre + "conv.4.weight" in sd:
        return "conv2d8"
    return "conv2d6" if sd[pre + "conv.2.weight"].size(-1) == 5 else "conv2d"


def conv2d_subsampling(sd: Dict[str, Tensor], feats: Tensor, pre: str = "encoder.embed.") -> Tensor:
    """espnet2/legacy/nets/pytorch_backend/transformer/subsampling.py:432-447 / :752-756 / :846-850 (without
    pos-enc): Conv2d(1,d,3,2)+ReLU, Conv2d(d,d,3,2 | 5,3)+ReLU[, Conv2d(d,d,3,2)+ReLU], (b,c,t,f)->(b,t,c*f),
    Linear."""
    kind = subsampling_kind(sd, pre)
    x = feats.unsqueeze(1)
    x = F.relu(F.conv2d(x, sd[pre + "conv.0.weight"], sd[pre + "conv.0.bias"], stride=2))
    x = F.relu(F.conv2d(x, sd[pre + "conv.2.weight"], sd[pre + "conv.2.bias"], stride=3 if kind == "conv2d6" else 2))
    if kind == "conv2d8":
        x = F.relu(F.conv2d(x, sd[pre + "conv.4.weight"], sd[pre + "conv.4.bias"], stride=2))
    b, c, t, f = x.size()
    x = x.transpose(1, 2).contiguous().view(b, t, c * f)
    return F.linear(x, sd[pre + "out.weight"], sd[pre + "out.bias"])


def subsampled_lengths(flens: Tensor, tmax: int, kind: str = "conv2d") -> Tensor:
    """Valid-frame counts after `mask[:, :, :-2:2][:, :, :-2:2]` (subsampling.py:448-449; conv2d6:
    `[:-2:2][:-4:3]` :758; conv2d8: three `[:-2:2]` :851).  The slicing acts on the PADDED mask of length
    tmax, so the result depends on tmax as well."""
    mask = ~make_pad_mask(flens, tmax)
    if kind == "conv2d6":
        mask = mask[:, :-2:2][:, :-4:3]
    elif kind == "conv2d8":
        mask = mask[:, :-2:2][:, :-2:2][:, :-2:2]
    else:
        mask = mask[:, :-2:2][:, :-2:2]
    return mask.sum(1)


SHORT_LIMIT = {"conv2d": 7, "conv2d6": 11, "conv2d8": 15}  # check_short_utt, subsampling.py:31-49


def layer_norm(x: Tensor, sd, pre: str) -> Tensor:
    return F.layer_norm(x, (x.size(-1),), sd[pre + "weight"], sd[pre + "bias"], LN_EPS)


def swish(x: Tensor) -> Tensor:
    """conformer/swish.py:13-18."""
    return x * torch.sigmoid(x)


def feed_forward(sd, x: Tensor, pre: str, act=swish) -> Tensor:
    """transformer/positionwise_feed_forward.py:30-32."""
    return F.linear(act(F.linear(x, sd[pre + "w_1.weight"], sd[pre + "w_1.bias"])),
                    sd[pre + "w_2.weight"], sd[pre + "w_2.bias"])


def rel_shift(x: Tensor) -> Tensor:
    """transformer/attention.py:391-408, literally (pad/view trick)."""
    zero_pad = torch.zeros((*x.size()[:3], 1), dtype=x.dtype)
    x_padded = torch.cat([zero_pad, x], dim=-1)
    x_padded = x_padded.view(*x.size()[:2], x.size(3) + 1, x.size(2))
    return x_padded[:, :, 1:].view_as(x)[:, :, :, : x.size(-1) // 2 + 1]


def rel_pos_attention(sd, x: Tensor, pos_emb: Tensor, key_valid: Tensor, pre: str, h: int) -> Tensor:
    """RelPositionMultiHeadedAttention.forward (transformer/attention.py:416-459) +
    forward_qkv (:77-119) + forward_attention (:121-151).  key_valid: (B, T) bool."""
    B, T, d = x.shape
    dk = d // h
    q = F.linear(x, sd[pre + "linear_q.weight"], sd[pre + "linear_q.bias"]).view(B, T, h, dk)
    k = F.linear(x, sd[pre + "linear_k.weight"], sd[pre + "linear_k.bias"]).view(B, T, h, dk)
    v = F.linear(x, sd[pre + "linear_v.weight"], sd[pre + "linear_v.bias"]).view(B, T, h, dk)
    k = k.transpose(1, 2)
    v = v.transpose(1, 2)
    p = F.linear(pos_emb, sd[pre + "linear_pos.weight"]).view(1, -1, h, dk).transpose(1, 2)
    q_u = (q + sd[pre + "pos_bias_u"]).transpose(1, 2)
    q_v = (q + sd[pre + "pos_bias_v"]).transpose(1, 2)
    ac = torch.matmul(q_u, k.transpose(-2, -1))
    bd = rel_shift(torch.matmul(q_v, p.transpose(-2, -1)))
    scores = (ac + bd) / math.sqrt(dk)
    mask = ~key_valid[:, None, None, :]
    scores = scores.masked_fill(mask, torch.finfo(scores.dtype).min)
    attn = torch.softmax(scores, dim=-1).masked_fill(mask, 0.0)
    ctx = torch.matmul(attn, v).transpose(1, 2).contiguous().view(B, T, d)
    return F.linear(ctx, sd[pre + "linear_out.weight"], sd[pre + "linear_out.bias"])


def legacy_pos_emb(T: int, d: int, max_len: int = 5000) -> Tensor:
    """LegacyRelPositionalEncoding (transformer/embedding.py:223-262 over PositionalEncoding(reverse=True)
    :50-82): the table is built once for `max_len` with positions max_len-1 .. 0 and the first T rows are
    used, so row k is the sinusoid of position max_len-1-k."""
    position = torch.arange(max_len - 1, max_len - 1 - T, -1.0, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(T, d)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def legacy_rel_shift(x: Tensor) -> Tensor:
    """LegacyRelPositionMultiHeadedAttention.rel_shift (transformer/attention.py:296-316), zero_triu=False."""
    zero_pad = torch.zeros((*x.size()[:3], 1), dtype=x.dtype)
    x_padded = torch.cat([zero_pad, x], dim=-1)
    x_padded = x_padded.view(*x.size()[:2], x.size(3) + 1, x.size(2))
    return x_padded[:, :, 1:].view_as(x)


def legacy_rel_pos_attention(sd, x: Tensor, pos_emb: Tensor, key_valid: Tensor, pre: str, h: int) -> Tensor:
    """LegacyRelPositionMultiHeadedAttention.forward (transformer/attention.py:318-360)."""
    B, T, d = x.shape
    dk = d // h
    q = F.linear(x, sd[pre + "linear_q.weight"], sd[pre + "linear_q.bias"]).view(B, T, h, dk)
    k = F.linear(x, sd[pre + "linear_k.weight"], sd[pre + "linear_k.bias"]).view(B, T, h, dk).transpose(1, 2)
    v = F.linear(x, sd[pre + "linear_v.weight"], sd[pre + "linear_v.bias"]).view(B, T, h, dk).transpose(1, 2)
    p = F.linear(pos_emb, sd[pre + "linear_pos.weight"]).view(1, -1, h, dk).transpose(1, 2)
    q_u = (q + sd[pre + "pos_bias_u"]).transpose(1, 2)
    q_v = (q + sd[pre + "pos_bias_v"]).transpose(1, 2)
    ac = torch.matmul(q_u, k.transpose(-2, -1))
    bd = legacy_rel_shift(torch.matmul(q_v, p.transpose(-2, -1)))
    scores = (ac + bd) / math.sqrt(dk)
    mask = ~key_valid[:, None, None, :]
    scores = scores.masked_fill(mask, torch.finfo(scores.dtype).min)
    attn = torch.softmax(scores, dim=-1).masked_fill(mask, 0.0)
    ctx = torch.matmul(attn, v).transpose(1, 2).contiguous().view(B, T, d)
    return F.linear(ctx, sd[pre + "linear_out.weight"], sd[pre + "linear_out.bias"])


def conv_module(sd, x: Tensor, pre: str) -> Tensor:
    """ConvolutionModule.forward (conformer/convolution.py:56-79); eval-mode BatchNorm.
    NB: no padding mask inside (padded frames leak into their neighbours, as in the reference)."""
    d = x.size(-1)
    x = x.transpose(1, 2)
    x = F.conv1d(x, sd[pre + "pointwise_conv1.weight"], sd[pre + "pointwise_conv1.bias"])
    x = F.glu(x, dim=1)
    w = sd[pre + "depthwise_conv.weight"]
    x = F.conv1d(x, w, sd[pre + "depthwise_conv.bias"], padding=(w.size(-1) - 1) // 2, groups=d)
    x = F.batch_norm(x, sd[pre + "norm.running_mean"], sd[pre + "norm.running_var"],
                     sd[pre + "norm.weight"], sd[pre + "norm.bias"], False, 0.0, BN_EPS)
    x = swish(x)
    x = F.conv1d(x, sd[pre + "pointwise_conv2.weight"], sd[pre + "pointwise_conv2.bias"])
    return x.transpose(1, 2)


def conformer_block(sd, x: Tensor, pos_emb: Tensor, key_valid: Tensor, pre: str, h: int,
                    legacy: bool = False) -> Tensor:
    """EncoderLayer.forward, eval mode, normalize_before, macaron, cnn module
    (conformer/encoder_layer.py:79-179; ff_scale 0.5 :65)."""
    attn = legacy_rel_pos_attention if legacy else rel_pos_attention
    x = x + 0.5 * feed_forward(sd, layer_norm(x, sd, pre + "norm_ff_macaron."), pre + "feed_forward_macaron.")
    x = x + attn(sd, layer_norm(x, sd, pre + "norm_mha."), pos_emb, key_valid, pre + "self_attn.", h)
    x = x + conv_module(sd, layer_norm(x, sd, pre + "norm_conv."), pre + "conv_module.")
    x = x + 0.5 * feed_forward(sd, layer_norm(x, sd, pre + "norm_ff."), pre + "feed_forward.")
    return layer_norm(x, sd, pre + "norm_final.")


class TooShortUttError(Exception):
    """Mirror of subsampling.py:14-29 (raised for < 7 feature frames, :31-49)."""

    def __init__(self, message, actual_size, limit):
        super().__init__(message)
        self.actual_size = actual_size
        self.limit = limit


def conformer_encoder(sd, feats: Tensor, flens: Tensor, heads: int, num_blocks: int,
                      return_blocks: bool = False, rel_pos_type: str = "latest"):
    """ConformerEncoder.forward (espnet2/asr/encoder/conformer_encoder.py:327-429) for
    input_layer=conv2d, rel_pos/rel_selfattn (latest), macaron, cnn module, normalize_before."""
    kind = subsampling_kind(sd)
    if feats.size(1) < SHORT_LIMIT[kind]:
        raise TooShortUttError(
            f"has {feats.size(1)} frames and is too short for subsampling "
            f"(it needs more than {SHORT_LIMIT[kind]} frames), return empty results", feats.size(1), SHORT_LIMIT[kind])
    x = conv2d_subsampling(sd, feats)
    d = x.size(-1)
    T = x.size(1)
    x = x * math.sqrt(d)  # embedding.py:328
    legacy = rel_pos_type == "legacy"
    pos = (legacy_pos_emb(T, d) if legacy else rel_pos_emb(T, d)).unsqueeze(0)
    olens = subsampled_lengths(flens, feats.size(1), kind)
    key_valid = ~make_pad_mask(olens, T)
    blocks = []
    for i in range(num_blocks):
        x = conformer_block(sd, x, pos, key_valid, f"encoder.encoders.{i}.", heads, legacy)
        if return_blocks:
            blocks.append(x)
    x = layer_norm(x, sd, "encoder.after_norm.")
    if return_blocks:
        return x, olens, blocks
    return x, olens


def encode(sd, speech: Tensor, speech_lengths: Tensor, heads: int, num_blocks: int,
           n_fft: int = 512, win_length: Optional[int] = None, hop: int = 160, rel_pos_type: str = "latest"):
    """ESPnetASRModel.encode (espnet2/asr/espnet_model.py:380-448) with DefaultFrontend +
    UtteranceMVN + ConformerEncoder."""
    feats, flens = frontend_feats(speech, speech_lengths, sd["frontend.logmel.melmat"], n_fft,
                                  win_length, hop)
    feats = utterance_mvn(feats, flens)
    return conformer_encoder(sd, feats, flens, heads, num_blocks, rel_pos_type=rel_pos_type)


# --------------------------------------------------------------------------- CTC head
def ctc_logits(sd, enc: Tensor) -> Tensor:
    return F.linear(enc, sd["ctc.ctc_lo.weight"], sd["ctc.ctc_lo.bias"])


def ctc_log_softmax(sd, enc: Tensor) -> Tensor:
    """espnet2/asr/ctc.py:197-205."""
    return F.log_softmax(ctc_logits(sd, enc), dim=2)


def ctc_argmax(sd, enc: Tensor) -> Tensor:
    """espnet2/asr/ctc.py:207-215."""
    return torch.argmax(ctc_logits(sd, enc), dim=2)


def g1_collapse(ids: List[int], exclude=(0,)) -> List[int]:
    """Greedy CTC "G1": groupby + drop blank/sos/eos (espnet2/bin/asr_inference.py:574-575)."""
    return [int(x[0]) for x in groupby(ids) if int(x[0]) not in exclude]


def greedy_ctc(sd, enc: Tensor, olens: Tensor, blank: int = 0, sos_eos: Optional[int] = None):
    ids = ctc_argmax(sd, enc)
    excl = (blank,) if sos_eos is None else (blank, sos_eos)
    return [g1_collapse(ids[b, : int(olens[b])].tolist(), excl) for b in range(ids.size(0))]
