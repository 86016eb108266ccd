"""EBranchformerEncoder on the MI355X: parameter tree + weight packing + one
C-ABI call per batch (`em_ebranchformer_encode`, csrc/ebranchformer.hip).

Mirrors espnet2/asr/encoder/e_branchformer_encoder.py:186-520 (constructor keywords, `output_size()`,
`forward(xs_pad, ilens, prev_states=None) -> (ys, olens, None)`) and exposes the SAME state-dict keys
as the reference (`embed.conv.{0,2}`, `embed.out`, `encoders.N.{attn, cgmlp.channel_proj1.0,
cgmlp.csgu.{norm,conv}, cgmlp.channel_proj2, feed_forward, feed_forward_macaron, norm_*,
depthwise_conv_fusion, merge_proj}`, `after_norm`), so reference checkpoints load unchanged.

Accelerated combination (the one the E-Branchformer recipes use): input_layer=conv2d, rel_pos /
rel_selfattn (latest), use_ffn + macaron_ffn with swish, cgMLP with identity gate and no linear after
the conv, d_k = 64; anything else raises NotImplementedError at construction.  Conv2dSubsampling,
rel-pos attention, FFN, LayerNorm are the Conformer kernels; the cgMLP adds a GELU GEMM epilogue, a
strided LayerNorm and a gated depthwise conv, the merge a depthwise conv with its residual fused.
The torch.nn layers below are parameter CONTAINERS: their forward() is never called.
"""
import ctypes as C
from typing import List, Optional

import torch

from espnet_amd import lib as L
from espnet_amd.asr.encoder.conformer_encoder import (ConformerEncoder, LayerNorm, _Conv2dSubsampling,
                                                      _PositionwiseFeedForward,
                                                      _RelPositionMultiHeadedAttention, pack_ffn_rows_w1,
                                                      pack_ffn_rows_w2)


class _CSGU(torch.nn.Module):
    """Parameters of ConvolutionalSpatialGatingUnit (espnet2/asr/layers/cgmlp.py:14-50)."""

    def __init__(self, size, kernel_size):
        super().__init__()
        n = size // 2
        self.norm = LayerNorm(n)
        self.conv = torch.nn.Conv1d(n, n, kernel_size, 1, (kernel_size - 1) // 2, groups=n)


class _CGMLP(torch.nn.Module):
    """Parameters of ConvolutionalGatingMLP (cgmlp.py:82-121)."""

    def __init__(self, size, linear_units, kernel_size):
        super().__init__()
        self.channel_proj1 = torch.nn.Sequential(torch.nn.Linear(size, linear_units), torch.nn.GELU())
        self.csgu = _CSGU(linear_units, kernel_size)
        self.channel_proj2 = torch.nn.Linear(linear_units // 2, size)


class _EBranchformerEncoderLayer(torch.nn.Module):
    """Parameters of EBranchformerEncoderLayer (e_branchformer_encoder.py:68-108); with ff = None and
    merge_kernel = None those of BranchformerEncoderLayer (branchformer_encoder.py:65-136), whose
    merge_method decides the merge parameters (:99-133)."""

    def __init__(self, size, heads, ff, cg, cg_kernel, merge_kernel, merge_method="concat"):
        super().__init__()
        self.attn = _RelPositionMultiHeadedAttention(heads, size)
        self.cgmlp = _CGMLP(size, cg, cg_kernel)
        if ff is not None:
            self.feed_forward = _PositionwiseFeedForward(size, ff)
            self.feed_forward_macaron = _PositionwiseFeedForward(size, ff)
            self.norm_ff = LayerNorm(size)
            self.norm_ff_macaron = LayerNorm(size)
        self.norm_mha = LayerNorm(size)
        self.norm_mlp = LayerNorm(size)
        self.norm_final = LayerNorm(size)
        if merge_kernel is not None:
            self.depthwise_conv_fusion = torch.nn.Conv1d(2 * size, 2 * size, merge_kernel, 1,
                                                         (merge_kernel - 1) // 2, groups=2 * size, bias=True)
        if merge_method == "learned_ave":
            self.pooling_proj1 = torch.nn.Linear(size, 1)
            self.pooling_proj2 = torch.nn.Linear(size, 1)
            self.weight_proj1 = torch.nn.Linear(size, 1)
            self.weight_proj2 = torch.nn.Linear(size, 1)
        self.merge_proj = torch.nn.Linear(2 * size if merge_method == "concat" else size, size)


class EBranchformerEncoder(ConformerEncoder):
    merge_method, cgmlp_weight = "concat", None
    _WS_FN, _ENC_FN = "em_ebranchformer_workspace_bytes", "em_ebranchformer_encode"

    def __init__(self, input_size: int, output_size: int = 256, attention_heads: int = 4,
                 attention_layer_type: str = "rel_selfattn", pos_enc_layer_type: str = "rel_pos",
                 rel_pos_type: str = "latest", cgmlp_linear_units: int = 2048, cgmlp_conv_kernel: int = 31,
                 use_linear_after_conv: bool = False, gate_activation: str = "identity", num_blocks: int = 12,
                 dropout_rate: float = 0.1, positional_dropout_rate: float = 0.1,
                 attention_dropout_rate: float = 0.0, input_layer: Optional[str] = "conv2d",
                 zero_triu: bool = False, padding_idx: int = -1, layer_drop_rate: float = 0.0,
                 max_pos_emb_len: int = 5000, use_ffn: bool = False, macaron_ffn: bool = False,
                 ffn_activation_type: str = "swish", linear_units: int = 2048,
                 positionwise_layer_type: str = "linear", merge_conv_kernel: int = 3, interctc_layer_idx=None,
                 interctc_use_conditioning: bool = False, qk_norm: bool = False, use_flash_attn: bool = True,
                 gradient_checkpoint_layers: List[int] = [], compute_dtype: str = "bfloat16"):
        torch.nn.Module.__init__(self)
        bad = []
        if input_layer not in ("conv2d", "conv2d6", "conv2d8"): bad.append(f"input_layer={input_layer}")
        # e_branchformer_encoder.py:226-235 / branchformer_encoder.py:330-339: "legacy" maps to the legacy_ classes
        if rel_pos_type == "legacy":
            pos_enc_layer_type = "legacy_rel_pos" if pos_enc_layer_type == "rel_pos" else pos_enc_layer_type
            attention_layer_type = "legacy_rel_selfattn" if attention_layer_type == "rel_selfattn" else attention_layer_type
        elif rel_pos_type != "latest":
            raise ValueError("unknown rel_pos_type: " + rel_pos_type)
        legacy = pos_enc_layer_type == "legacy_rel_pos" and attention_layer_type == "legacy_rel_selfattn"
        if not legacy:
            if attention_layer_type != "rel_selfattn": bad.append(f"attention_layer_type={attention_layer_type}")
            if pos_enc_layer_type != "rel_pos": bad.append(f"pos_enc_layer_type={pos_enc_layer_type}")
        if use_linear_after_conv: bad.append("use_linear_after_conv=True")
        if gate_activation != "identity": bad.append(f"gate_activation={gate_activation}")
        if not (use_ffn and macaron_ffn): bad.append("use_ffn/macaron_ffn must both be True")
        if ffn_activation_type != "swish": bad.append(f"ffn_activation_type={ffn_activation_type}")
        if positionwise_layer_type != "linear": bad.append(f"positionwise_layer_type={positionwise_layer_type}")
        if zero_triu: bad.append("zero_triu=True")
        if interctc_layer_idx or interctc_use_conditioning: bad.append("interctc")
        if qk_norm: bad.append("qk_norm=True")
        if output_size % 64 or output_size // attention_heads != 64: bad.append("d_k != 64")
        if linear_units % 64: bad.append("linear_units % 64 != 0")
        if cgmlp_linear_units % 128: bad.append("cgmlp_linear_units % 128 != 0")
        if cgmlp_conv_kernel not in (3, 7, 15, 31): bad.append(f"cgmlp_conv_kernel={cgmlp_conv_kernel}")
        if merge_conv_kernel not in (3, 7, 15, 31): bad.append(f"merge_conv_kernel={merge_conv_kernel}")
        if bad:
            raise NotImplementedError("outside the MI355X E-Branchformer fast path: " + ", ".join(bad))
        self._output_size, self._input_size = output_size, input_size
        self.heads, self.linear_units, self.num_blocks = attention_heads, linear_units, num_blocks
        self.cgmlp_linear_units, self.cgmlp_conv_kernel = cgmlp_linear_units, cgmlp_conv_kernel
        self.merge_conv_kernel = merge_conv_kernel
        self.interctc_layer_idx, self.interctc_use_conditioning = [], False
        self.compute_dtype = compute_dtype
        self.input_layer = input_layer
        self.legacy_relpos, self.max_pos_emb_len = legacy, max_pos_emb_len
        self.embed = _Conv2dSubsampling(input_size, output_size, input_layer)
        self.encoders = torch.nn.ModuleList(
            [_EBranchformerEncoderLayer(output_size, attention_heads, linear_units, cgmlp_linear_units,
                                        cgmlp_conv_kernel, merge_conv_kernel) for _ in range(num_blocks)])
        self.after_norm = LayerNorm(output_size)
        self._packed, self._pos_cache, self._ws, self._olens_cache = None, {}, None, {}

    def pack(self, device):
        dev = torch.device(device)
        act = self.act_dtype
        d, ff, Lb, cg = self._output_size, self.linear_units, self.num_blocks, self.cgmlp_linear_units
        if (cg // 2) % (64 if self.em_dtype == L.EM_BF16 else 32):
            raise NotImplementedError("cgmlp_linear_units / 2 must be a multiple of the GEMM K step")
        keep = []

        def A(t):
            t = t.detach().to(torch.float32).contiguous().to(act).to(dev)
            keep.append(t)
            return t

        def F(t):
            t = t.detach().to(torch.float32).contiguous().to(dev)
            keep.append(t)
            return t

        e = self.embed
        F2 = e.out.in_features // d
        w = L.EmEBranchformerWeights()
        w.d, w.heads, w.cg, w.num_blocks = d, self.heads, cg, Lb
        w.cg_kernel, w.merge_kernel, w.n_mels = self.cgmlp_conv_kernel, self.merge_conv_kernel or 0, self._input_size
        has_ffn, has_mconv = ff is not None, self.merge_conv_kernel is not None
        w.ff = ff or 0
        w.use_ffn, w.merge_conv = int(has_ffn), int(has_mconv)
        w.merge_method = L.EM_MERGE_LEARNED_AVE if self.merge_method == "learned_ave" else L.EM_MERGE_CONCAT
        t = dict(conv1_w=F(e.conv[0].weight.reshape(d, 9)), conv1_b=F(e.conv[0].bias),
                 embed_w=A(e.out.weight.reshape(d, d, F2).permute(0, 2, 1).reshape(d, F2 * d)),
                 embed_b=F(e.out.bias),
                 wpos_all=A(torch.cat([l.attn.linear_pos.weight for l in self.encoders], dim=0)),
                 after_norm_g=F(self.after_norm.weight), after_norm_b=F(self.after_norm.bias))
        self._pack_subsampling(w, t, A, F)
        for k, v in t.items():
            setattr(w, k, v.data_ptr())
        layers = (L.EmEBranchformerLayer * Lb)()
        for i, l in enumerate(self.encoders):
            sa, cm = l.attn, l.cgmlp
            lt = dict(
                norm_mha_g=F(l.norm_mha.weight), norm_mha_b=F(l.norm_mha.bias),
                norm_mlp_g=F(l.norm_mlp.weight), norm_mlp_b=F(l.norm_mlp.bias),
                norm_final_g=F(l.norm_final.weight), norm_final_b=F(l.norm_final.bias),
                wqkv=A(torch.cat([sa.linear_q.weight, sa.linear_k.weight, sa.linear_v.weight], 0)),
                bqkv=F(torch.cat([sa.linear_q.bias, sa.linear_k.bias, sa.linear_v.bias], 0)),
                pos_u=F(sa.pos_bias_u), pos_v=F(sa.pos_bias_v),
                wout=A(sa.linear_out.weight), bout=F(sa.linear_out.bias),
                proj1_w=A(cm.channel_proj1[0].weight), proj1_b=F(cm.channel_proj1[0].bias),
                csgu_norm_g=F(cm.csgu.norm.weight), csgu_norm_b=F(cm.csgu.norm.bias),
                csgu_conv_w=F(cm.csgu.conv.weight.reshape(cg // 2, -1).t()),  # [k][cg/2] tap-major
                csgu_conv_b=F(cm.csgu.conv.bias),
                proj2_w=A(cm.channel_proj2.weight), proj2_b=F(cm.channel_proj2.bias),
                merge_b=F(l.merge_proj.bias))
            mw = l.merge_proj.weight.detach().to(torch.float32)
            if self.merge_method == "fixed_ave":
                # merge_proj((1 - c) x1 + c x2) == [(1 - c) W | c W] [x1 | x2]: the concat launch sequence
                c = float(self.cgmlp_weight[i])
                mw = torch.cat([(1.0 - c) * mw, c * mw], dim=1)
            elif self.merge_method == "learned_ave":
                lt.update(pool_w=F(torch.cat([l.pooling_proj1.weight, l.pooling_proj2.weight], 0)),
                          pool_b=F(torch.cat([l.pooling_proj1.bias, l.pooling_proj2.bias], 0)),
                          wproj_w=F(torch.cat([l.weight_proj1.weight, l.weight_proj2.weight], 0)),
                          wproj_b=F(torch.cat([l.weight_proj1.bias, l.weight_proj2.bias], 0)))
            lt["merge_w"] = A(mw)
            if has_ffn:
                lt.update(
                    norm_ff_mac_g=F(l.norm_ff_macaron.weight), norm_ff_mac_b=F(l.norm_ff_macaron.bias),
                    norm_ff_g=F(l.norm_ff.weight), norm_ff_b=F(l.norm_ff.bias),
                    ffm_w1=A(l.feed_forward_macaron.w_1.weight), ffm_b1=F(l.feed_forward_macaron.w_1.bias),
                    ffm_w2=A(l.feed_forward_macaron.w_2.weight), ffm_b2=F(l.feed_forward_macaron.w_2.bias),
                    ff_w1=A(l.feed_forward.w_1.weight), ff_b1=F(l.feed_forward.w_1.bias),
                    ff_w2=A(l.feed_forward.w_2.weight), ff_b2=F(l.feed_forward.w_2.bias))
            if has_ffn and d == 512 and ff % 128 == 0 and ff >= 256 and act == torch.bfloat16:
                # operand streams of the row-block feed-forward launches (csrc/ffn_rows.hip), as the 512-wide Conformer's
                lt.update(ffm_w1p=A(pack_ffn_rows_w1(l.feed_forward_macaron.w_1.weight)),
                          ffm_w2p=A(pack_ffn_rows_w2(l.feed_forward_macaron.w_2.weight)),
                          ff_w1p=A(pack_ffn_rows_w1(l.feed_forward.w_1.weight)),
                          ff_w2p=A(pack_ffn_rows_w2(l.feed_forward.w_2.weight)),
                          wqkvp=A(pack_ffn_rows_w1(torch.cat([sa.linear_q.weight, sa.linear_k.weight, sa.linear_v.weight], 0))))
            if has_mconv:
                lt.update(merge_conv_w=F(l.depthwise_conv_fusion.weight.reshape(2 * d, -1).t()),
                          merge_conv_b=F(l.depthwise_conv_fusion.bias))
            for k, v in lt.items():
                setattr(layers[i], k, v.data_ptr())
        w.layers = C.cast(layers, C.POINTER(L.EmEBranchformerLayer))
        self._packed = dict(w=w, layers=layers, keep=keep, device=dev, dtype=self.em_dtype)
        self._pos_cache = {}
        return self._packed


class BranchformerEncoder(EBranchformerEncoder):
    """BranchformerEncoder (espnet2/asr/encoder/branchformer_encoder.py:293-620) with both branches: the
    E-Branchformer layer without feed-forward modules and without the depthwise conv in front of `merge_proj`
    (state-dict keys `encoders.N.{attn, cgmlp, norm_mha, norm_mlp, norm_final, merge_proj}`, plus
    `pooling_proj{1,2}` / `weight_proj{1,2}` for learned_ave); same kernels,
    `EmEBranchformerWeights.use_ffn = merge_conv = 0`.  merge_method: concat; fixed_ave (0 < cgmlp_weight < 1,
    a float or one per block — folded into the packed merge_proj weight); learned_ave (the pooled per-utterance
    branch weights are computed on the device, `em_branch_learned_ave`; attn_branch_drop_rate is training-only)."""

    def __init__(self, input_size: int, output_size: int = 256, use_attn: bool = True, attention_heads: int = 4,
                 attention_layer_type: str = "rel_selfattn", pos_enc_layer_type: str = "rel_pos",
                 rel_pos_type: str = "latest", use_cgmlp: bool = True, cgmlp_linear_units: int = 2048,
                 cgmlp_conv_kernel: int = 31, use_linear_after_conv: bool = False,
                 gate_activation: str = "identity", merge_method: str = "concat", cgmlp_weight=0.5,
                 attn_branch_drop_rate=0.0, num_blocks: int = 12, dropout_rate: float = 0.1,
                 positional_dropout_rate: float = 0.1, attention_dropout_rate: float = 0.0,
                 input_layer: Optional[str] = "conv2d", zero_triu: bool = False, padding_idx: int = -1,
                 stochastic_depth_rate=0.0, qk_norm: bool = False, use_flash_attn: bool = True,
                 compute_dtype: str = "bfloat16"):
        torch.nn.Module.__init__(self)
        bad = []
        if not (use_attn and use_cgmlp): bad.append("use_attn and use_cgmlp must both be True")
        if merge_method not in ("concat", "learned_ave", "fixed_ave"):
            raise ValueError(f"unknown merge method: {merge_method}")  # branchformer_encoder.py:133
        if isinstance(cgmlp_weight, (int, float)):  # :490-496
            cgmlp_weight = [float(cgmlp_weight)] * num_blocks
        if len(cgmlp_weight) != num_blocks:
            raise ValueError(f"Length of cgmlp_weight ({len(cgmlp_weight)}) should be equal to "
                             f"num_blocks ({num_blocks})")
        if merge_method == "fixed_ave":
            assert all(0.0 <= c <= 1.0 for c in cgmlp_weight), "cgmlp weight should be between 0.0 and 1.0"
            if any(c in (0.0, 1.0) for c in cgmlp_weight):  # :120-127 drops one branch and its parameters
                bad.append("fixed_ave with cgmlp_weight 0.0 / 1.0 (single-branch layers)")
        if input_layer not in ("conv2d", "conv2d6", "conv2d8"): bad.append(f"input_layer={input_layer}")
        # e_branchformer_encoder.py:226-235 / branchformer_encoder.py:330-339: "legacy" maps to the legacy_ classes
        if rel_pos_type == "legacy":
            pos_enc_layer_type = "legacy_rel_pos" if pos_enc_layer_type == "rel_pos" else pos_enc_layer_type
            attention_layer_type = "legacy_rel_selfattn" if attention_layer_type == "rel_selfattn" else attention_layer_type
        elif rel_pos_type != "latest":
            raise ValueError("unknown rel_pos_type: " + rel_pos_type)
        legacy = pos_enc_layer_type == "legacy_rel_pos" and attention_layer_type == "legacy_rel_selfattn"
        if not legacy:
            if attention_layer_type != "rel_selfattn": bad.append(f"attention_layer_type={attention_layer_type}")
            if pos_enc_layer_type != "rel_pos": bad.append(f"pos_enc_layer_type={pos_enc_layer_type}")
        if use_linear_after_conv: bad.append("use_linear_after_conv=True")
        if gate_activation != "identity": bad.append(f"gate_activation={gate_activation}")
        if zero_triu: bad.append("zero_triu=True")
        if qk_norm: bad.append("qk_norm=True")
        if output_size % 64 or output_size // attention_heads != 64: bad.append("d_k != 64")
        if cgmlp_linear_units % 128: bad.append("cgmlp_linear_units % 128 != 0")
        if cgmlp_conv_kernel not in (3, 7, 15, 31): bad.append(f"cgmlp_conv_kernel={cgmlp_conv_kernel}")
        if bad:
            raise NotImplementedError("outside the MI355X Branchformer fast path: " + ", ".join(bad))
        self._output_size, self._input_size = output_size, input_size
        self.heads, self.linear_units, self.num_blocks = attention_heads, None, num_blocks
        self.cgmlp_linear_units, self.cgmlp_conv_kernel = cgmlp_linear_units, cgmlp_conv_kernel
        self.merge_conv_kernel = None
        self.merge_method, self.cgmlp_weight = merge_method, [float(c) for c in cgmlp_weight]
        self.interctc_layer_idx, self.interctc_use_conditioning = [], False
        self.compute_dtype = compute_dtype
        self.input_layer = input_layer
        self.legacy_relpos, self.max_pos_emb_len = legacy, 5000  # BranchformerEncoder has no max_pos_emb_len argument (pos_enc_class default)
        self.embed = _Conv2dSubsampling(input_size, output_size, input_layer)
        self.encoders = torch.nn.ModuleList(
            [_EBranchformerEncoderLayer(output_size, attention_heads, None, cgmlp_linear_units, cgmlp_conv_kernel,
                                        None, merge_method) for _ in range(num_blocks)])
        self.after_norm = LayerNorm(output_size)
        self._packed, self._pos_cache, self._ws, self._olens_cache = None, {}, None, {}
