"""ConformerEncoder on the MI355X: parameter tree + weight packing + one C-ABI call per batch.

Mirrors espnet2/asr/encoder/conformer_encoder.py:53-429 (constructor keywords, `output_size()`,
`forward(xs_pad, ilens, prev_states=None) -> (ys, olens, None)`) and exposes the SAME state-dict
keys as the reference (`embed.conv.{0,2}`, `embed.out`, `encoders.N.{self_attn,feed_forward,
feed_forward_macaron,conv_module,norm_*}`, `after_norm`), so reference checkpoints load unchanged.

Only the BASELINE combination is accelerated (input_layer=conv2d, rel_pos / rel_selfattn with
rel_pos_type=latest, macaron_style, use_cnn_module, normalize_before, swish, d_k = 64); any other
combination raises NotImplementedError at construction (a maintainer integrating this class into
the reference registry keeps the stock `conformer` entry for those).

The torch.nn layers below are parameter CONTAINERS: their forward() is never called.  The
arithmetic is csrc/{frontend,gemm,norm,attention,conv,encoder}.hip.
"""
import ctypes as C
import math
import os
import threading
import weakref
from typing import List, Optional, Tuple

import torch

from espnet_amd import lib as L
from espnet_amd.nets_utils import (SUBSAMPLING_CONVS, SUBSAMPLING_MIN_FRAMES, conv2d_subsampled_lengths,
                                   conv_out_size)

_POS_PROJ_LOCK = threading.Lock()
# the packed weights struct carries the per-call `ctc_ids` pointer: setting it and launching the encoder is one critical
# section for host threads that encode concurrently; the ids a call produced are kept per THREAD
_ENC_CALL_LOCK = threading.Lock()
_TLS = threading.local()

LN_EPS = 1e-12  # transformer/layer_norm.py:23


class LayerNorm(torch.nn.LayerNorm):
    def __init__(self, nout):
        super().__init__(nout, eps=LN_EPS)


class _Conv2dSubsampling(torch.nn.Module):
    """Parameters of Conv2dSubsampling / Conv2dSubsampling6 / Conv2dSubsampling8
    (transformer/subsampling.py:386-409, 692-715, 785-808): `conv.{0,2[,4]}`, `out`."""

    def __init__(self, idim, odim, input_layer: str = "conv2d"):
        super().__init__()
        mods, cin = [], 1
        for k, s in SUBSAMPLING_CONVS[input_layer]:
            mods += [torch.nn.Conv2d(cin, odim, k, s), torch.nn.ReLU()]
            cin = odim
        self.conv = torch.nn.Sequential(*mods)
        self.out = torch.nn.Linear(odim * conv_out_size(idim, input_layer), odim)


class _RelPositionMultiHeadedAttention(torch.nn.Module):
    """Parameters of transformer/attention.py:362-389 (+ MultiHeadedAttention :24-75)."""

    def __init__(self, n_head, n_feat):
        super().__init__()
        self.d_k, self.h = n_feat // n_head, n_head
        self.linear_q = torch.nn.Linear(n_feat, n_feat)
        self.linear_k = torch.nn.Linear(n_feat, n_feat)
        self.linear_v = torch.nn.Linear(n_feat, n_feat)
        self.linear_out = torch.nn.Linear(n_feat, n_feat)
        self.linear_pos = torch.nn.Linear(n_feat, n_feat, bias=False)
        self.pos_bias_u = torch.nn.Parameter(torch.Tensor(self.h, self.d_k))
        self.pos_bias_v = torch.nn.Parameter(torch.Tensor(self.h, self.d_k))
        torch.nn.init.xavier_uniform_(self.pos_bias_u)
        torch.nn.init.xavier_uniform_(self.pos_bias_v)


class _PositionwiseFeedForward(torch.nn.Module):
    def __init__(self, idim, hidden):
        super().__init__()
        self.w_1 = torch.nn.Linear(idim, hidden)
        self.w_2 = torch.nn.Linear(hidden, idim)


class _ConvolutionModule(torch.nn.Module):
    """Parameters of conformer/convolution.py:22-54."""

    def __init__(self, channels, kernel_size):
        super().__init__()
        assert (kernel_size - 1) % 2 == 0
        self.pointwise_conv1 = torch.nn.Conv1d(channels, 2 * channels, 1)
        self.depthwise_conv = torch.nn.Conv1d(channels, channels, kernel_size,
                                              padding=(kernel_size - 1) // 2, groups=channels)
        self.norm = torch.nn.BatchNorm1d(channels)
        self.pointwise_conv2 = torch.nn.Conv1d(channels, channels, 1)


class _EncoderLayer(torch.nn.Module):
    """Parameters of conformer/encoder_layer.py:43-77."""

    def __init__(self, size, heads, ff, kernel):
        super().__init__()
        self.self_attn = _RelPositionMultiHeadedAttention(heads, size)
        self.feed_forward = _PositionwiseFeedForward(size, ff)
        self.feed_forward_macaron = _PositionwiseFeedForward(size, ff)
        self.conv_module = _ConvolutionModule(size, kernel)
        self.norm_ff = LayerNorm(size)
        self.norm_mha = LayerNorm(size)
        self.norm_ff_macaron = LayerNorm(size)
        self.norm_conv = LayerNorm(size)
        self.norm_final = LayerNorm(size)


def _fused_enabled(enc) -> bool:
    """The fused per-block kernels are the default where they apply; `enc.fused = False` or
    ESPNET_AMD_FUSED=0 keeps the one-operator-per-launch sequence (A/B measurements, bisecting)."""
    import os

    return bool(getattr(enc, "fused", True)) and os.environ.get("ESPNET_AMD_FUSED", "1") != "0"


def pack_k_units(w: torch.Tensor) -> torch.Tensor:
    """[N][256] -> the fragment-major "K units" of the fused block kernel (include/espnet_amd.h, EmBlockArgs): per
    64 rows one 32 KiB unit [nf][ks][lg][lr][e] = w[64 u + 16 nf + lr][32 ks + 8 lg + e], so that each MFMA weight
    operand of a wave is one contiguous 1 KiB line."""
    n, k = w.shape
    assert k == 256 and n % 64 == 0, (n, k)
    return w.detach().reshape(n // 64, 4, 16, 8, 4, 8).permute(0, 1, 3, 4, 2, 5).contiguous().reshape(n, k)


def _pad_chunks(ff: int) -> int:
    """FFN width as the fused kernel walks it: whole PAIRS of 64-wide chunks."""
    assert ff % 64 == 0, ff
    return (ff + 127) // 128 * 128


def pack_w1(w: torch.Tensor) -> torch.Tensor:
    """First matrix of an FFN, [ff][256] -> K units, zero-padded to an even number of 64-row chunks (the padded
    chunk's hidden activation is swish(0 + 0) = 0 and meets zero columns of w_2)."""
    ff = w.shape[0]
    return pack_k_units(torch.nn.functional.pad(w.detach(), (0, 0, 0, _pad_chunks(ff) - ff)))


def pack_w2(w: torch.Tensor) -> torch.Tensor:
    """Second matrix of an FFN, [256][ff] -> 64 KiB per pair of hidden chunks in the order the kernel's waves hold the
    hidden activation (include/espnet_amd.h, EmBlockArgs):
    pair[p][w][f][lg][lr][e] = W[16 f + lr][64 (2 p + (e >> 2)) + 16 w + 4 lg + (e & 3)]."""
    d, ff = w.shape
    assert d == 256 and ff % 64 == 0, (d, ff)
    ffp = _pad_chunks(ff)
    wp = torch.nn.functional.pad(w.detach(), (0, ffp - ff))
    # [o = 16 f + lr][hidden = 128 p + 64 ci + 16 wv + 4 lg + r] -> [p][wv][f][lg][lr][ci][r]
    u = wp.reshape(16, 16, ffp // 128, 2, 4, 4, 4).permute(2, 4, 0, 5, 1, 3, 6).contiguous()
    return u.reshape(ffp // 128, 4 * 16 * 64 * 8)


def pack_ffn_rows_w1(w: torch.Tensor) -> torch.Tensor:
    """First matrix of an FFN of the 512-wide model, [ff][512] -> the operand stream of csrc/ffn_rows.hip
    (include/espnet_amd.h, EmFfnRowsArgs): [c][ks][wv][lg][lr][e] = W1[128 c + 16 wv + lr][32 ks + 8 lg + e]."""
    ff, k = w.shape
    assert k == 512 and ff % 128 == 0, (ff, k)
    # [row = 128 c + 16 wv + lr][col = 32 ks + 8 lg + e] -> [c][ks][wv][lg][lr][e]
    u = w.detach().reshape(ff // 128, 8, 16, 16, 4, 8).permute(0, 3, 1, 4, 2, 5).contiguous()
    return u.reshape(ff, k)


def pack_ffn_rows_w2(w: torch.Tensor) -> torch.Tensor:
    """Second matrix, [512][ff] -> [c][s][cf][wv][lg][lr][e] = W2[64 wv + 16 cf + lr][128 c + 32 s + 16 (e >> 2) + 4 lg + (e & 3)]:
    the contraction index in the order the kernel's waves hold the hidden activation."""
    d, ff = w.shape
    assert d == 512 and ff % 128 == 0, (d, ff)
    # [o = 64 wv + 16 cf + lr][hidden = 128 c + 32 s + 16 eh + 4 lg + el] -> [c][s][cf][wv][lg][lr][eh][el]
    u = w.detach().reshape(8, 4, 16, ff // 128, 4, 2, 4, 4).permute(3, 4, 1, 0, 6, 2, 5, 7).contiguous()
    return u.reshape(ff // 128, 4 * 4 * 8 * 64 * 8)


def pack_rows_proj(w: torch.Tensor) -> torch.Tensor:
    """A [512][512] projection in front of a row-block feed-forward launch (EmFfnRowsArgs.pre_w):
    [ks][cf][wv][lg][lr][e] = W[64 wv + 16 cf + lr][32 ks + 8 lg + e]."""
    n, k = w.shape
    assert n == 512 and k == 512, (n, k)
    return w.detach().reshape(8, 4, 16, 16, 4, 8).permute(3, 1, 0, 4, 2, 5).contiguous().reshape(n, k)


def glu_chunk_order(d: int = 512) -> torch.Tensor:
    """Row order of pointwise_conv1 ([value rows 0 .. d) | gate rows d .. 2 d), F.glu(dim=1), convolution.py:66) for the
    row-block GLU launch: chunk 2 j = the value rows of output columns 128 j .. 128 j + 127, chunk 2 j + 1 their gate rows."""
    idx = torch.arange(2 * d).reshape(2, d // 128, 128)  # [value | gate][j][128]
    return idx.permute(1, 0, 2).reshape(-1)


def pack_rows_glu(w: torch.Tensor) -> torch.Tensor:
    """pointwise_conv1 [2 * 512][512] -> the w1p layout of EmFfnRowsArgs in value / gate chunk pairs (EM_ROWS_GLU)."""
    return pack_ffn_rows_w1(w.detach()[glu_chunk_order(w.shape[1])])


def pack_conv1_frags(w1: torch.Tensor, b1: torch.Tensor) -> torch.Tensor:
    """conv.0 weight [256][9] + bias [256] (f32) -> the MFMA operand of the fused conv1 + conv2 kernel (include/espnet_amd.h,
    em_conv2d_sub12_bf16): per channel 32 bf16 k-slots  hi(w) | hi(w) | lo(w) | hi(b), lo(b), 0, 0, 0  with hi(x) = bf16(x),
    lo(x) = bf16(x - hi(x)), laid out [chunk][fragment][lg][lr][e] for channel 32 cc + 16 f + lr, k = 8 lg + e."""
    w1 = w1.detach().to(torch.float32).cpu().reshape(-1, 9)
    b1 = b1.detach().to(torch.float32).cpu()
    d = w1.shape[0]
    assert d % 32 == 0, d

    def hi(x):
        return x.to(torch.bfloat16).to(torch.float32)

    wh, bh = hi(w1), hi(b1)
    wl, bl = hi(w1 - wh), hi(b1 - bh)
    k = torch.zeros(d, 32)
    k[:, 0:9], k[:, 9:18], k[:, 18:27], k[:, 27], k[:, 28] = wh, wh, wl, bh, bl
    return k.reshape(d // 32, 2, 16, 4, 8).permute(0, 1, 3, 2, 4).contiguous().reshape(-1)


def pack_conv2_frags(w2: torch.Tensor) -> torch.Tensor:
    """conv.2 weight as [256][9 * 256] with column (kt*3 + kf) * 256 + c_in -> fragment-major
    [chunk cc][tap][wave w][fragment j][lg][lr][e] = w2[64 w + 16 (lr // 4) + 4 j + lr % 4][tap * 256 + 32 cc + 8 lg + e]:
    MFMA row lr of a wave's fragment j is an output channel chosen so that a lane's 16 results of a position
    (4 fragments x 4 rows) are 16 CONSECUTIVE channels - two 16-byte stores instead of four 8-byte ones."""
    n, k = w2.shape
    assert n % 256 == 0 and k == 9 * n, (n, k)  # (d = 512: one block of this layout per 256 output channels, 16 chunks each)
    halves = []
    for h0 in range(0, n, 256):
        u = w2.detach()[h0 : h0 + 256].reshape(4, 4, 4, 4, 9, n // 32, 4, 8)  # [w][q = lr // 4][j][r = lr % 4][tap][cc][lg][e]
        halves.append(u.permute(5, 4, 0, 2, 6, 1, 3, 7).contiguous().reshape(-1))
    return torch.cat(halves)


def rel_pos_table(T: int, d: int) -> torch.Tensor:
    """RelPositionalEncoding rows for a length-T input (embedding.py:286-332): row k is the
    sinusoid of relative position T-1-k.  Built on the host with the same fp32 torch ops the
    reference uses (`extend_pe`), once per distinct T."""
    pe_positive = torch.zeros(T, d)
    pe_negative = torch.zeros(T, d)
    position = torch.arange(0, T, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe_positive[:, 0::2] = torch.sin(position * div_term)
    pe_positive[:, 1::2] = torch.cos(position * div_term)
    pe_negative[:, 0::2] = torch.sin(-1 * position * div_term)
    pe_negative[:, 1::2] = torch.cos(-1 * position * div_term)
    return torch.cat([torch.flip(pe_positive, [0]), pe_negative[1:]], dim=0)


class ConformerEncoder(torch.nn.Module):
    _WS_FN, _ENC_FN = "em_conformer_workspace_bytes", "em_conformer_encode"  # C-ABI entry points of forward_device

    @staticmethod
    def _option_check(*, input_layer, normalize_before, concat_after, positionwise_layer_type, macaron_style,
                      rel_pos_type, pos_enc_layer_type, selfattention_layer_type, activation_type, use_cnn_module,
                      zero_triu, interctc_layer_idx, interctc_use_conditioning, ctc_trim, qk_norm, output_size,
                      attention_heads, linear_units, cnn_module_kernel):
        """Options of espnet2/asr/encoder/conformer_encoder.py:89-121 that the MI355X kernels do not cover ->
        (list of "name=value" strings, legacy rel-pos flag).  An empty list = the fast path applies."""
        bad = []
        if input_layer not in SUBSAMPLING_CONVS: bad.append(f"input_layer={input_layer}")
        if not normalize_before: bad.append("normalize_before=False")
        if concat_after: bad.append("concat_after=True")
        if positionwise_layer_type != "linear": bad.append(f"positionwise_layer_type={positionwise_layer_type}")
        if not macaron_style: bad.append("macaron_style=False")
        # conformer_encoder.py:127-136: rel_pos_type "legacy" turns rel_pos / rel_selfattn into their legacy_ forms
        if rel_pos_type == "legacy":
            pos_enc_layer_type = "legacy_rel_pos" if pos_enc_layer_type == "rel_pos" else pos_enc_layer_type
            selfattention_layer_type = ("legacy_rel_selfattn" if selfattention_layer_type == "rel_selfattn"
                                        else selfattention_layer_type)
        elif rel_pos_type != "latest":
            raise ValueError("unknown rel_pos_type: " + rel_pos_type)
        legacy = pos_enc_layer_type == "legacy_rel_pos" and selfattention_layer_type == "legacy_rel_selfattn"
        if not legacy:
            if pos_enc_layer_type != "rel_pos": bad.append(f"pos_enc_layer_type={pos_enc_layer_type}")
            if selfattention_layer_type != "rel_selfattn": bad.append(f"selfattention_layer_type={selfattention_layer_type}")
        if activation_type != "swish": bad.append(f"activation_type={activation_type}")
        if not use_cnn_module: bad.append("use_cnn_module=False")
        if zero_triu: bad.append("zero_triu=True")
        if len(interctc_layer_idx) > 0 or interctc_use_conditioning or ctc_trim: bad.append("interctc/ctc_trim")
        if qk_norm: bad.append("qk_norm=True")
        if output_size % 64 or output_size // attention_heads != 64: bad.append("d_k != 64")
        if linear_units % 64: bad.append("linear_units % 64 != 0")
        if cnn_module_kernel not in (3, 7, 15, 31): bad.append(f"cnn_module_kernel={cnn_module_kernel}")
        return bad, legacy

    @classmethod
    def unsupported_options(cls, *args, **kwargs) -> List[str]:
        """The constructor arguments (positional or keyword, reference defaults applied) that fall outside the fast
        path, without building anything: what `espnet_amd.integration.espnet2_adapters` consults to hand such a
        configuration to the stock espnet2 class under the same yaml name."""
        import inspect

        ba = inspect.signature(cls.__init__).bind(None, *args, **kwargs)
        ba.apply_defaults()
        names = inspect.signature(cls._option_check).parameters
        try:
            return cls._option_check(**{k: ba.arguments[k] for k in names})[0]
        except ValueError as e:
            return [str(e)]

    def __init__(self, input_size: int, output_size: int = 256, attention_heads: int = 4,
                 linear_units: int = 2048, num_blocks: int = 6, dropout_rate: float = 0.1,
                 positional_dropout_rate: float = 0.1, attention_dropout_rate: float = 0.0,
                 input_layer: Optional[str] = "conv2d", normalize_before: bool = True,
                 concat_after: bool = False, positionwise_layer_type: str = "linear",
                 positionwise_conv_kernel_size: int = 3, macaron_style: bool = False,
                 rel_pos_type: str = "legacy", pos_enc_layer_type: str = "rel_pos",
                 selfattention_layer_type: str = "rel_selfattn", activation_type: str = "swish",
                 use_cnn_module: bool = True, zero_triu: bool = False, cnn_module_kernel: int = 31,
                 padding_idx: int = -1, interctc_layer_idx: List[int] = [],
                 interctc_use_conditioning: bool = False, ctc_trim: bool = False,
                 stochastic_depth_rate=0.0, layer_drop_rate: float = 0.0,
                 max_pos_emb_len: int = 5000, qk_norm: bool = False, use_flash_attn: bool = True,
                 compute_dtype: str = "bfloat16"):
        super().__init__()
        bad, legacy = self._option_check(
            input_layer=input_layer, normalize_before=normalize_before, concat_after=concat_after,
            positionwise_layer_type=positionwise_layer_type, macaron_style=macaron_style, rel_pos_type=rel_pos_type,
            pos_enc_layer_type=pos_enc_layer_type, selfattention_layer_type=selfattention_layer_type,
            activation_type=activation_type, use_cnn_module=use_cnn_module, zero_triu=zero_triu,
            interctc_layer_idx=interctc_layer_idx, interctc_use_conditioning=interctc_use_conditioning,
            ctc_trim=ctc_trim, qk_norm=qk_norm, output_size=output_size, attention_heads=attention_heads,
            linear_units=linear_units, cnn_module_kernel=cnn_module_kernel)
        if bad:
            raise NotImplementedError("outside the MI355X Conformer fast path: " + ", ".join(bad))
        self._output_size = output_size
        self._input_size = input_size
        self.heads, self.linear_units, self.num_blocks = attention_heads, linear_units, num_blocks
        self.cnn_module_kernel = cnn_module_kernel
        self.normalize_before = normalize_before
        self.interctc_layer_idx = list(interctc_layer_idx)
        self.interctc_use_conditioning = interctc_use_conditioning
        self.compute_dtype = compute_dtype
        self.input_layer = input_layer
        self.legacy_relpos, self.max_pos_emb_len = legacy, max_pos_emb_len
        self.embed = _Conv2dSubsampling(input_size, output_size, input_layer)
        self.encoders = torch.nn.ModuleList(
            [_EncoderLayer(output_size, attention_heads, linear_units, cnn_module_kernel)
             for _ in range(num_blocks)])
        self.after_norm = LayerNorm(output_size)
        self._packed = None
        self._pos_cache = {}
        self._ws = None
        self._olens_cache = {}
        self.last_ctc_ids = None
        object.__setattr__(self, "fused_ctc", None)  # set by the model: a CTC head to fuse (not a submodule of the encoder)

    def output_size(self) -> int:
        return self._output_size

    # ------------------------------------------------------------------ packing (load time)
    @property
    def em_dtype(self) -> int:
        return L.DTYPES[self.compute_dtype]

    @property
    def act_dtype(self) -> torch.dtype:
        return torch.bfloat16 if self.em_dtype == L.EM_BF16 else torch.float32

    def invalidate(self):
        self._packed = None
        self._pos_cache = {}

    def pack(self, device):
        """Repack the reference-layout parameters into the layouts the kernels consume (once)."""
        dev = torch.device(device)
        act = self.act_dtype
        d, ff, Lb = self._output_size, self.linear_units, self.num_blocks
        keep = []

        def A(t):  # act-dtype matrix on the device
            t = t.detach().to(torch.float32).contiguous().to(act).to(dev)
            keep.append(t)
            return t

        def F(t):  # f32 vector/table on the device
            t = t.detach().to(torch.float32).contiguous().to(dev)
            keep.append(t)
            return t

        e = self.embed
        F2 = e.out.in_features // d
        w = L.EmConformerWeights()
        w.d, w.heads, w.ff, w.num_blocks = d, self.heads, ff, Lb
        w.kernel, w.n_mels = self.cnn_module_kernel, self._input_size
        t = {}
        t["conv1_w"] = F(e.conv[0].weight.reshape(d, 9))
        t["conv1_b"] = F(e.conv[0].bias)
        self._pack_subsampling(w, t, A, F)
        t["embed_w"] = A(e.out.weight.reshape(d, d, F2).permute(0, 2, 1).reshape(d, F2 * d))
        t["embed_b"] = F(e.out.bias)
        t["wpos_all"] = A(torch.cat([l.self_attn.linear_pos.weight for l in self.encoders], dim=0))
        t["after_norm_g"], t["after_norm_b"] = F(self.after_norm.weight), F(self.after_norm.bias)
        for k, v in t.items():
            setattr(w, k, v.data_ptr())
        layers = (L.EmConformerLayer * Lb)()
        glu_perm = torch.arange(2 * d).reshape(2, d // 16, 16).permute(1, 0, 2).reshape(-1)
        for i, l in enumerate(self.encoders):
            sa, cm = l.self_attn, l.conv_module
            bn = cm.norm
            scale = (bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps))
            dw_w = cm.depthwise_conv.weight.double().reshape(d, -1) * scale[:, None]
            dw_b = (cm.depthwise_conv.bias.double() - bn.running_mean.double()) * scale + bn.bias.double()
            lt = dict(
                norm_ff_mac_g=F(l.norm_ff_macaron.weight), norm_ff_mac_b=F(l.norm_ff_macaron.bias),
                norm_mha_g=F(l.norm_mha.weight), norm_mha_b=F(l.norm_mha.bias),
                norm_conv_g=F(l.norm_conv.weight), norm_conv_b=F(l.norm_conv.bias),
                norm_ff_g=F(l.norm_ff.weight), norm_ff_b=F(l.norm_ff.bias),
                norm_final_g=F(l.norm_final.weight), norm_final_b=F(l.norm_final.bias),
                ffm_w1=A(l.feed_forward_macaron.w_1.weight), ffm_b1=F(l.feed_forward_macaron.w_1.bias),
                ffm_w2=A(l.feed_forward_macaron.w_2.weight), ffm_b2=F(l.feed_forward_macaron.w_2.bias),
                wqkv=A(torch.cat([sa.linear_q.weight, sa.linear_k.weight, sa.linear_v.weight], 0)),
                bqkv=F(torch.cat([sa.linear_q.bias, sa.linear_k.bias, sa.linear_v.bias], 0)),
                pos_u=F(sa.pos_bias_u), pos_v=F(sa.pos_bias_v),
                wout=A(sa.linear_out.weight), bout=F(sa.linear_out.bias),
                pw1=A(cm.pointwise_conv1.weight.reshape(2 * d, d)[glu_perm]),
                pw1_b=F(cm.pointwise_conv1.bias[glu_perm]),
                dw_w=F(dw_w.t()), dw_b=F(dw_b),  # [k][d]: tap-major, coalesced over channels
                pw2=A(cm.pointwise_conv2.weight.reshape(d, d)), pw2_b=F(cm.pointwise_conv2.bias),
                ff_w1=A(l.feed_forward.w_1.weight), ff_b1=F(l.feed_forward.w_1.bias),
                ff_w2=A(l.feed_forward.w_2.weight), ff_b2=F(l.feed_forward.w_2.bias),
            )
            for k, v in lt.items():
                setattr(layers[i], k, v.data_ptr())
        if self._fusable():
            self._pack_fused(layers, A, F)
            self._pack_fused_ctc(w, A, F)
        elif self.em_dtype == L.EM_BF16 and d == 512 and ff % 128 == 0 and ff >= 256 and _fused_enabled(self):
            # the 512-wide model: each feed-forward module as one row-block launch (csrc/ffn_rows.hip)
            for i, l in enumerate(self.encoders):
                lt = dict(ffm_w1p=A(pack_ffn_rows_w1(l.feed_forward_macaron.w_1.weight)),
                          ffm_w2p=A(pack_ffn_rows_w2(l.feed_forward_macaron.w_2.weight)),
                          ff_w1p=A(pack_ffn_rows_w1(l.feed_forward.w_1.weight)),
                          ff_w2p=A(pack_ffn_rows_w2(l.feed_forward.w_2.weight)),
                          pw2p=A(pack_rows_proj(l.conv_module.pointwise_conv2.weight.reshape(d, d))),
                          woutp=A(pack_rows_proj(l.self_attn.linear_out.weight)),
                          pw1f=A(pack_rows_glu(l.conv_module.pointwise_conv1.weight.reshape(2 * d, d))),
                          fp_c=F(l.conv_module.pointwise_conv1.bias[glu_chunk_order(d)]))
                if self.heads * 64 == d:  # round 6: q | k | v walked behind the macaron launch (EmFfnRowsArgs.post_q)
                    sa = l.self_attn
                    lt["wqkvp"] = A(pack_ffn_rows_w1(torch.cat([sa.linear_q.weight, sa.linear_k.weight, sa.linear_v.weight], 0)))
                for k, v in lt.items():
                    setattr(layers[i], k, v.data_ptr())
            self._pack_rows_ctc(w, A, F)
        w.layers = C.cast(layers, C.POINTER(L.EmConformerLayer))
        self._packed = dict(w=w, layers=layers, keep=keep, device=dev, dtype=self.em_dtype)
        self._pos_cache = {}
        return self._packed

    def _fusable(self) -> bool:
        """Shapes the row-block fused kernels cover (csrc/block.hip); everything else keeps the per-operator launches."""
        return (self.em_dtype == L.EM_BF16 and self._output_size == 256 and self.heads == 4
                and self.linear_units <= 1024 and self.cnn_module_kernel == 31
                and not getattr(self, "legacy_relpos", False))

    def _pack_fused(self, layers, A, F):
        """Operands of em_conformer_block_fused (include/espnet_amd.h): pointwise_conv1 in 64-row value / gate
        granules and the bias / LayerNorm vectors of every kernel as groups of EM_BLOCK_PARAM_GROUP floats in the
        order the kernel consumes them."""
        d, G = self._output_size, L.EM_BLOCK_PARAM_GROUP

        def group(*vecs):
            v = torch.cat([t.detach().to(torch.float32).reshape(-1).cpu() for t in vecs])
            assert v.numel() <= G
            return torch.nn.functional.pad(v, (0, G - v.numel()))

        def pad_ff(b):
            return torch.nn.functional.pad(b.detach().to(torch.float32).cpu(), (0, 1024 - b.numel()))

        perm = torch.cat([torch.cat([torch.arange(64 * j, 64 * j + 64), torch.arange(d + 64 * j, d + 64 * j + 64)])
                          for j in range(d // 64)])

        def a_groups(l):
            sa = l.self_attn
            return [group(l.norm_ff_macaron.weight, l.norm_ff_macaron.bias, pad_ff(l.feed_forward_macaron.w_1.bias),
                          l.feed_forward_macaron.w_2.bias),
                    group(l.norm_mha.weight, l.norm_mha.bias, sa.linear_q.bias, sa.linear_k.bias, sa.linear_v.bias)]

        n = len(self.encoders)
        for i, l in enumerate(self.encoders):
            cm = l.conv_module
            pw1 = cm.pointwise_conv1.weight.reshape(2 * d, d)
            d_groups = [group(cm.pointwise_conv2.bias, l.norm_ff.weight, l.norm_ff.bias),
                        group(pad_ff(l.feed_forward.w_1.bias), l.feed_forward.w_2.bias, l.norm_final.weight,
                              l.norm_final.bias)]
            tail = a_groups(self.encoders[i + 1]) if i + 1 < n else [group(self.after_norm.weight,
                                                                           self.after_norm.bias)]
            sa = l.self_attn
            lt = dict(pw1f=A(pack_k_units(pw1[perm])), ffm_w2p=A(pack_w2(l.feed_forward_macaron.w_2.weight)),
                      ff_w2p=A(pack_w2(l.feed_forward.w_2.weight)),
                      woutp=A(pack_k_units(sa.linear_out.weight)),
                      pw2p=A(pack_k_units(cm.pointwise_conv2.weight.reshape(d, d))),
                      ff_w1p=A(pack_w1(l.feed_forward.w_1.weight)),
                      ffm_w1p=A(pack_w1(l.feed_forward_macaron.w_1.weight)),
                      wqkvp=A(pack_k_units(torch.cat([sa.linear_q.weight, sa.linear_k.weight, sa.linear_v.weight], 0))),
                      fp_c=F(group(l.self_attn.linear_out.bias, l.norm_conv.weight, l.norm_conv.bias,
                                   cm.pointwise_conv1.bias[perm])),
                      fp_da=F(torch.cat(d_groups + tail)))
            if i == 0:
                lt["fp_a"] = F(torch.cat(a_groups(l)))
            for k, v in lt.items():
                setattr(layers[i], k, v.data_ptr())

    def _pack_fused_ctc(self, w, A, F):
        """The CTC head the model attached (`fused_ctc`, a espnet_amd.asr.ctc.CTC): its weight zero-padded to 64-row
        units and its bias padded with -3e38, for the arg-max stage of the last block kernel (EM_BLOCK_CTC)."""
        ctc = getattr(self, "fused_ctc", None)
        if ctc is None or ctc.eprojs != self._output_size:
            return
        V = ctc.odim
        units = (V + 63) // 64
        if units > L.EM_BLOCK_CTC_MAX_UNITS:
            return
        wt = torch.zeros(units * 64, self._output_size, dtype=torch.float32)
        wt[:V] = ctc.ctc_lo.weight.detach().to(torch.float32).cpu()
        bt = torch.full((units * 64,), -3.0e38, dtype=torch.float32)
        bt[:V] = ctc.ctc_lo.bias.detach().to(torch.float32).cpu()
        w.ctc_w, w.ctc_b, w.ctc_units = A(pack_k_units(wt)).data_ptr(), F(bt).data_ptr(), units
        self._ctc_stamp = self._ctc_version(ctc)

    def _pack_rows_ctc(self, w, A, F):
        """512-wide model: the attached CTC head for the arg-max walk behind the last row-block launch (EmFfnRowsArgs.post_*):
        weight zero-padded to whole 128-row chunks in the w1p layout, bias padded with -3e38."""
        ctc = getattr(self, "fused_ctc", None)
        if ctc is None or ctc.eprojs != self._output_size:
            return
        V = ctc.odim
        chunks = (V + 127) // 128
        wt = torch.zeros(chunks * 128, self._output_size, dtype=torch.float32)
        wt[:V] = ctc.ctc_lo.weight.detach().to(torch.float32).cpu()
        bt = torch.full((chunks * 128,), -3.0e38, dtype=torch.float32)
        bt[:V] = ctc.ctc_lo.bias.detach().to(torch.float32).cpu()
        w.ctc_w, w.ctc_b, w.ctc_units = A(pack_ffn_rows_w1(wt)).data_ptr(), F(bt).data_ptr(), chunks
        self._ctc_stamp = self._ctc_version(ctc)

    @staticmethod
    def _ctc_version(ctc):
        lo = ctc.ctc_lo
        return (lo.weight.data_ptr(), lo.weight._version, lo.bias.data_ptr(), lo.bias._version)

    def _pack_subsampling(self, w, t, A, F):
        """conv.2 (and conv.4): [d][k*k*d] with column (kt*k + kf)*d + c_in, the implicit GEMM's K order."""
        e, d = self.embed, self._output_size
        layer = getattr(self, "input_layer", "conv2d")
        w.subsample = {"conv2d": 4, "conv2d6": 6, "conv2d8": 8}[layer]
        w.legacy_relpos = int(getattr(self, "legacy_relpos", False))
        k2 = e.conv[2].weight.size(-1)
        t["conv2_w"] = A(e.conv[2].weight.permute(0, 2, 3, 1).reshape(d, k2 * k2 * d))
        t["conv2_b"] = F(e.conv[2].bias)
        if layer == "conv2d" and d in (256, 512) and self.em_dtype == L.EM_BF16:
            # operands of the fused conv1 + conv2 kernel (csrc/subsample2.hip): the conv1 map is never materialised
            t["conv1_wf"] = A(pack_conv1_frags(e.conv[0].weight, e.conv[0].bias))
            t["conv2_wf"] = A(pack_conv2_frags(e.conv[2].weight.permute(0, 2, 3, 1).reshape(d, 9 * d)))
        if layer == "conv2d8":
            t["conv3_w"] = A(e.conv[4].weight.permute(0, 2, 3, 1).reshape(d, 9 * d))
            t["conv3_b"] = F(e.conv[4].bias)

    def _ensure_packed(self, device):
        p = self._packed
        if p is None or p["device"] != device or p["dtype"] != self.em_dtype:
            p = self.pack(device)
        return p

    def _pos_projected(self, T: int, device, pk, pos: torch.Tensor) -> torch.Tensor:
        """pos (2T-1 | T, d) x wpos_all^T -> (rows, num_blocks * d) in the activation dtype, through the same em_gemm
        call the encoder entry point makes without EM_ENC_POS_PROJECTED (bit-identical); kept for the last few lengths
        of the current packing."""
        cache = self.__dict__.setdefault("_pos_proj_cache", {})
        lock = _POS_PROJ_LOCK  # (concurrent batches on host threads; module-level: a lock in __dict__ is not picklable)
        key = (T, str(device), self.em_dtype, id(pk["w"]))
        cur = torch.cuda.current_stream()
        with lock:
            ent = cache.get(key)
            if ent is not None and ent[3] is not pk["w"]:  # (an id reused by a later packing: not this weight block)
                ent = None
            if ent is not None:
                out, ev, sid, _ = ent
                if sid != cur.cuda_stream:  # produced on another stream (concurrent batches): order this one behind
                    cur.wait_event(ev)      # it, and keep the block from being recycled under this stream's readers
                    out.record_stream(cur)  # when the entry is evicted (a same-stream free is ordered anyway)
                return out
        out = None
        if out is None:
            pos = pos.contiguous()
            rows, d = pos.shape
            n = self.num_blocks * d
            # the fused 256-wide bf16 path also wants the rows fragment-major (block<ATT|C>): packed here, once per length,
            # behind the table (EM_ENC_POS_PACKED)
            packed = (self.em_dtype == L.EM_BF16 and d == 256 and rows == 2 * T - 1 and not getattr(pk["w"], "legacy_relpos", 0))
            es = 2 if self.em_dtype == L.EM_BF16 else 4
            tbytes = (rows * n * es + 255) // 256 * 256
            npg = L.load().em_relpos_pos_fragments(T) if packed else 0
            flat = torch.empty(tbytes + self.num_blocks * 4 * npg * 2048, dtype=torch.uint8, device=device)
            out = flat[:rows * n * es].view(self.act_dtype).view(rows, n)
            out._em_flat, out._em_packed = flat, packed  # (keeps the block alive; read back in forward_device)
            args = L.EmGemmArgs(A=pos.data_ptr(), W=pk["w"].wpos_all, C=out.data_ptr(), bias=None, M=rows, N=n, K=d,
                                lda=d, ldc=n, scale=1.0)
            L.check(L.load().em_gemm(self.em_dtype, L.EM_EPI_STORE, L.EM_A_PLAIN, C.byref(args),
                                     L.current_stream_ptr()), "em_gemm(linear_pos)")
            if packed:
                L.check(L.load().em_relpos_pack_pos_bf16(out.data_ptr(), n, T, self.num_blocks, flat.data_ptr() + tbytes,
                                                         L.current_stream_ptr()), "em_relpos_pack_pos_bf16")
            ev = torch.cuda.Event()
            ev.record(cur)
            with lock:
                while len(cache) >= 8:
                    cache.pop(next(iter(cache)))
                cache[key] = (out, ev, cur.cuda_stream, pk["w"])  # (the weight block is kept alive with its projection)
        return out

    def _pos_emb(self, T: int, device) -> torch.Tensor:
        """(2T-1, d) rows for a length-T input: a contiguous row slice of one table built for a maximum
        length, exactly as the reference slices its `pe` buffer (embedding.py:329-332).  Row k of the
        slice is the sinusoid of relative position T-1-k whatever the table length, so the values are
        bit-identical to a table built for T; the table grows by doubling (built once per size)."""
        if getattr(self, "legacy_relpos", False):
            return self._legacy_pos_emb(T, device)
        key = (str(device), self.em_dtype)
        tab = self._pos_cache.get(key)
        if tab is None or tab[0] < T:
            tmax = max(512, tab[0] * 2 if tab else 0)
            while tmax < T:
                tmax *= 2
            tab = (tmax, rel_pos_table(tmax, self._output_size).to(self.act_dtype).to(device))
            self._pos_cache[key] = tab
        tmax, table = tab
        return table[tmax - T : tmax + T - 1]

    def _legacy_pos_emb(self, T: int, device) -> torch.Tensor:
        """(T, d): LegacyRelPositionalEncoding rows (embedding.py:223-262 over PositionalEncoding(reverse=True)
        :50-82): the reference builds positions max_len-1 .. 0 once and uses the first T rows, so row k is the
        sinusoid of position max_len-1-k — a prefix of one table (rebuilt longer like `extend_pe` if needed)."""
        key = ("legacy", str(device), self.em_dtype)
        tab = self._pos_cache.get(key)
        if tab is None or tab.size(0) < T:
            n = max(T, self.max_pos_emb_len)
            d = self._output_size
            position = torch.arange(n - 1, -1, -1.0, dtype=torch.float32).unsqueeze(1)
            div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
            pe = torch.zeros(n, d)
            pe[:, 0::2] = torch.sin(position * div_term)
            pe[:, 1::2] = torch.cos(position * div_term)
            tab = pe.to(self.act_dtype).to(device)
            self._pos_cache[key] = tab
        return tab[:T]

    # ------------------------------------------------------------------ forward
    def output_frames(self, T_f: int) -> int:
        return conv_out_size(T_f, getattr(self, "input_layer", "conv2d"))

    def forward_device(self, feats: torch.Tensor, flens: List[int], flens_dev: torch.Tensor,
                       mvn_partial: Optional[torch.Tensor] = None, isolate: bool = False):
        """feats (B,T_f,D) f32 on the GPU.  Returns (enc_out f32 (B,T,d), enc_act (B,T,d) in the
        compute dtype, olens list, olens_dev i32)."""
        L.require_gpu(feats, "feats")
        B, T_f, D = feats.shape
        # check_short_utt (subsampling.py:31-49) via conformer_encoder.py:360-369; the reference sees one
        # utterance per call, so in a padded batch every row is held to the same limit
        layer = getattr(self, "input_layer", "conv2d")
        lim = SUBSAMPLING_MIN_FRAMES[layer]
        short = [b for b, n in enumerate(flens) if n < lim] if T_f >= lim else list(range(B))
        if short:
            n0 = min(T_f, int(flens[short[0]]))
            raise L.TooShortUttError(
                f"has {n0} frames and is too short for subsampling "
                f"(it needs more than {lim} frames), return empty results", n0, lim, indices=short)
        dev = feats.device
        pk = self._ensure_packed(dev)
        ctc = getattr(self, "fused_ctc", None)
        if ctc is not None and getattr(pk["w"], "ctc_units", 0) > 0 and self._ctc_stamp != self._ctc_version(ctc):
            pk = self.pack(dev)  # the head's parameters were replaced after packing
        lib = L.load()
        T = self.output_frames(T_f)
        if isolate:  # every utterance as if it were the whole batch: tmax = its own length
            olens = [conv2d_subsampled_lengths([n], int(n), layer)[0] for n in flens]
        else:  # the padded mask is sliced, so padded rows keep up to two frames more (subsampling.py:448)
            olens = conv2d_subsampled_lengths(flens, T_f, layer)
        okey = (tuple(olens), dev)
        olens_dev = self._olens_cache.get(okey)
        if olens_dev is None:  # kept on the device for repeating batch shapes (see encode_device)
            olens_dev = torch.tensor(olens, dtype=torch.int32).to(dev, non_blocking=True)
            if len(self._olens_cache) >= 8:
                self._olens_cache.pop(next(iter(self._olens_cache)))
            self._olens_cache[okey] = olens_dev
        need = getattr(lib, self._WS_FN)(self.em_dtype, C.byref(pk["w"]), B, T_f)
        # one workspace per stream: independent utterance batches may be encoded concurrently on
        # different HIP streams
        skey = torch.cuda.current_stream().cuda_stream
        if self._ws is None:
            self._ws = {}
        ws = self._ws.get(skey)
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self._ws[skey] = ws
        d = self._output_size
        enc_out = torch.empty(B, T, d, dtype=torch.float32, device=dev)
        enc_act = torch.empty(B, T, d, dtype=self.act_dtype, device=dev)
        # the fused path also hands back the CTC head's per-frame arg-max (EM_BLOCK_CTC): no logits, no second pass
        # (whether they WILL be written is the library's decision, asked through em_conformer_encode_plan: the host does
        # not re-derive the kernel's shape conditions)
        self.last_ctc_ids = None
        enc_flags = (L.EM_ENC_ISOLATE_UTTS if isolate else 0) | (0 if _fused_enabled(self) else L.EM_ENC_NO_FUSED)
        if getattr(self, "fold_c", None) if getattr(self, "fold_c", None) is not None else os.environ.get("ESPNET_AMD_FOLD") == "1":
            enc_flags |= L.EM_ENC_FOLD_C  # block<C|D|...>: two launches per block (opt-in: measured no faster)
        if getattr(self, "split_att", None) if getattr(self, "split_att", None) is not None else os.environ.get("ESPNET_AMD_SPLIT_ATT") == "1":
            enc_flags |= L.EM_ENC_SPLIT_ATT  # attention and block<C> as two launches (rounds 2-5) instead of block<ATT|C>: A/B switch
        # batches the caller keeps in flight on other HIP streams (bench.py's StepPipeline, the decode CLI's lanes set the
        # attribute): the library takes the 512-wide models' row-block launches from a smaller share of the chip then
        enc_flags |= L.EM_ENC_IN_FLIGHT(max(1, min(15, int(getattr(self, "batches_in_flight", 1) or 1))))
        pos = self._pos_emb(T, dev)
        if self._ENC_FN == "em_conformer_encode" and getattr(pk["w"], "wpos_all", None):
            # linear_pos of every block depends on T and the weights only: projected once per length, handed over ready
            pos = self._pos_projected(T, dev, pk, pos)
            enc_flags |= L.EM_ENC_POS_PROJECTED | (L.EM_ENC_POS_PACKED if getattr(pos, "_em_packed", False) else 0)
        with _ENC_CALL_LOCK:
            if hasattr(pk["w"], "ctc_ids"):
                pk["w"].ctc_ids = None
                if self._ENC_FN == "em_conformer_encode" and getattr(pk["w"], "ctc_units", 0) > 0:
                    ids = torch.empty(B, T, dtype=torch.int32, device=dev)
                    pk["w"].ctc_ids = ids.data_ptr()
                    plan = lib.em_conformer_encode_plan_for(self.em_dtype, C.byref(pk["w"]), enc_flags, B, T_f)
                    if plan < 0:
                        L.check(plan, "em_conformer_encode_plan")
                    if plan & L.EM_ENC_PLAN_CTC_IDS:
                        self.last_ctc_ids = ids
                    else:
                        pk["w"].ctc_ids = None
            rc = getattr(lib, self._ENC_FN)(
                self.em_dtype, C.byref(pk["w"]), L.ptr(feats), L.ptr(mvn_partial), L.ptr(flens_dev),
                L.ptr(olens_dev), B, T_f, L.ptr(pos), L.ptr(ws),
                ws.numel(), L.ptr(enc_out), L.ptr(enc_act),
                enc_flags, L.current_stream_ptr())
        L.check(rc, self._ENC_FN)
        return enc_out, enc_act, olens, olens_dev

    @property
    def last_ctc_ids(self):
        """Per-frame CTC arg-max ids of the calling thread's last `forward_device` (None when the launch sequence taken
        does not produce them)."""
        ids = getattr(_TLS, "ids", None)
        return ids.get(self) if ids is not None else None

    @last_ctc_ids.setter
    def last_ctc_ids(self, v):
        if not hasattr(_TLS, "ids"):
            _TLS.ids = weakref.WeakKeyDictionary()  # (keyed by the encoder itself: an entry dies with it, ids are never reused)
        _TLS.ids[self] = v

    def forward(self, xs_pad: torch.Tensor, ilens: torch.Tensor, prev_states: torch.Tensor = None
                ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
        flens = [int(v) for v in ilens.tolist()]
        flens_dev = torch.tensor(flens, dtype=torch.int32).to(xs_pad.device, non_blocking=True)
        enc_out, _, olens, _ = self.forward_device(xs_pad.to(torch.float32).contiguous(), flens,
                                                   flens_dev, None)
        return enc_out, torch.tensor(olens, dtype=torch.long, device=xs_pad.device), None
