"""ContextualBlockConformerEncoder (streaming, BASELINE config 5) on the MI355X.

Mirrors espnet2/asr/encoder/contextual_block_conformer_encoder.py:39-600 for streaming inference:
the constructor keywords, `output_size()`, `forward(xs_pad, ilens, prev_states, is_final=...,
infer_mode=True)` and `forward_infer(xs_pad, ilens, prev_states, is_final)` with the reference's
state dictionary (`prev_addin`, `buffer_before_downsampling`, `ilens_buffer`,
`buffer_after_downsampling`, `n_processed_blocks`, `past_encoder_ctx`), and the reference's
state-dict keys (`embed.conv.{0,2}`, `embed.out`, `encoders.N.{self_attn,feed_forward,
feed_forward_macaron,conv_module,norm1,norm2,norm_ff_macaron,norm_conv,norm_final}`, `after_norm`).

Split of work:
  * host (this file, integers + tensor slicing only): the buffering before / after the 4x
    subsampling, block counting, output stitching — exactly the reference's control flow
    (:386-600), because it decides WHICH frames a call processes;
  * device (csrc/streaming.hip, gemm.hip, norm.hip, conv.hip, frontend.hip): every arithmetic op —
    subsampling convs, block assembly with stream positional encoding and block means, all
    encoder layers, context hand-over, after_norm.

`step_graph(...)` captures one steady-state call (fixed chunk -> fixed block count) into a hipGraph
(torch.cuda.CUDAGraph on ROCm) so a chunk costs one graph launch instead of ~200 kernel launches.
The torch.nn layers are parameter containers only.
"""
import ctypes as C
import math
from typing import Optional, Tuple

import torch

from espnet_amd import lib as L
from espnet_amd.asr.decoder.transformer_decoder import abs_pos_table
from espnet_amd.asr.encoder.conformer_encoder import (LayerNorm, _ConvolutionModule,
                                                      _PositionwiseFeedForward)

LN_EPS = 1e-12


class _Conv2dSubsamplingWOPosEnc(torch.nn.Module):
    """Parameters of transformer/subsampling_without_posenc.py:11-42 (kernels [3,3], strides [2,2])."""

    def __init__(self, idim, odim):
        super().__init__()
        self.conv = torch.nn.Sequential(torch.nn.Conv2d(1, odim, 3, 2), torch.nn.ReLU(),
                                        torch.nn.Conv2d(odim, odim, 3, 2), torch.nn.ReLU())
        self.out = torch.nn.Linear(odim * (((idim - 3) // 2 + 1 - 3) // 2 + 1), odim)


class _MultiHeadedAttention(torch.nn.Module):
    def __init__(self, n_head, n_feat):
        super().__init__()
        self.linear_q = torch.nn.Linear(n_feat, n_feat)
        self.linear_k = torch.nn.Linear(n_feat, n_feat)
        self.linear_v = torch.nn.Linear(n_feat, n_feat)
        self.linear_out = torch.nn.Linear(n_feat, n_feat)


class _ContextualBlockEncoderLayer(torch.nn.Module):
    """Parameters of conformer/contextual_block_encoder_layer.py:46-77."""

    def __init__(self, size, heads, ff, kernel):
        super().__init__()
        self.self_attn = _MultiHeadedAttention(heads, size)
        self.feed_forward = _PositionwiseFeedForward(size, ff)
        self.feed_forward_macaron = _PositionwiseFeedForward(size, ff)
        self.conv_module = _ConvolutionModule(size, kernel)
        self.norm1 = LayerNorm(size)
        self.norm2 = LayerNorm(size)
        self.norm_ff_macaron = LayerNorm(size)
        self.norm_conv = LayerNorm(size)
        self.norm_final = LayerNorm(size)


class ContextualBlockConformerEncoder(torch.nn.Module):
    def __init__(self, input_size: int, output_size: int = 256, attention_heads: int = 4,
                 linear_units: int = 2048, num_blocks: int = 6, dropout_rate: float = 0.1,
                 positional_dropout_rate: float = 0.1, attention_dropout_rate: float = 0.0,
                 input_layer: Optional[str] = "conv2d", normalize_before: bool = True,
                 concat_after: bool = False, positionwise_layer_type: str = "linear",
                 positionwise_conv_kernel_size: int = 3, macaron_style: bool = False,
                 pos_enc_class=None, selfattention_layer_type: str = "rel_selfattn",
                 activation_type: str = "swish", use_cnn_module: bool = True,
                 cnn_module_kernel: int = 31, padding_idx: int = -1, block_size: int = 40,
                 hop_size: int = 16, look_ahead: int = 16, init_average: bool = True,
                 ctx_pos_enc: bool = True, compute_dtype: str = "bfloat16"):
        super().__init__()
        bad = []
        if input_layer != "conv2d": bad.append(f"input_layer={input_layer}")
        if not normalize_before: bad.append("normalize_before=False")
        if concat_after: bad.append("concat_after=True")
        if positionwise_layer_type != "linear": bad.append(f"positionwise_layer_type={positionwise_layer_type}")
        if not macaron_style: bad.append("macaron_style=False")
        if pos_enc_class is not None: bad.append("pos_enc_class")
        if activation_type != "swish": bad.append(f"activation_type={activation_type}")
        if not use_cnn_module: bad.append("use_cnn_module=False")
        if not init_average: bad.append("init_average=False")
        if not ctx_pos_enc: bad.append("ctx_pos_enc=False")
        if output_size % 64 or output_size // attention_heads not in (32, 64): bad.append("d_k not in {32,64}")
        if linear_units % 64: bad.append("linear_units % 64 != 0")
        if cnn_module_kernel not in (3, 7, 15, 31): bad.append(f"cnn_module_kernel={cnn_module_kernel}")
        if block_size <= 0 or block_size + 2 > 64: bad.append(f"block_size={block_size}")
        if bad:
            raise NotImplementedError("outside the MI355X streaming-Conformer fast path: " + ", ".join(bad))
        self._output_size, self._input_size = output_size, input_size
        self.heads, self.linear_units, self.num_blocks = attention_heads, linear_units, num_blocks
        self.cnn_module_kernel = cnn_module_kernel
        self.normalize_before = normalize_before
        self.block_size, self.hop_size, self.look_ahead = block_size, hop_size, look_ahead
        self.init_average, self.ctx_pos_enc = init_average, ctx_pos_enc
        self.subsample = 4
        self.compute_dtype = compute_dtype
        self.embed = _Conv2dSubsamplingWOPosEnc(input_size, output_size)
        self.encoders = torch.nn.ModuleList(
            [_ContextualBlockEncoderLayer(output_size, attention_heads, linear_units, cnn_module_kernel)
             for _ in range(num_blocks)])
        self.after_norm = LayerNorm(output_size)
        self._packed = None
        self._ws = {}
        self._flen_cache = {}  # (streams, frames, device) -> per-stream frame counts on the device (_embed_device_batch)

    def output_size(self) -> int:
        return self._output_size

    @property
    def em_dtype(self) -> int:
        return L.DTYPES[self.compute_dtype]

    @property
    def act_dtype(self) -> torch.dtype:
        return torch.bfloat16 if self.em_dtype == L.EM_BF16 else torch.float32

    def invalidate(self):
        self._packed = None

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self.invalidate()
        return r

    # ------------------------------------------------------------------ packing (load time)
    def pack(self, device):
        dev = torch.device(device)
        act = self.act_dtype
        d, ff, NL = self._output_size, self.linear_units, self.num_blocks
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
        w = L.EmConformerWeights()
        w.d, w.heads, w.ff, w.num_blocks = d, self.heads, ff, NL
        w.kernel, w.n_mels = self.cnn_module_kernel, self._input_size
        top = dict(conv1_w=F(e.conv[0].weight.reshape(d, 9)), conv1_b=F(e.conv[0].bias),
                   conv2_w=A(e.conv[2].weight.permute(0, 2, 3, 1).reshape(d, 9 * d)),
                   conv2_b=F(e.conv[2].bias),
                   embed_w=A(e.out.weight.reshape(d, d, F2).permute(0, 2, 1).reshape(d, F2 * d)),
                   embed_b=F(e.out.bias), after_norm_g=F(self.after_norm.weight),
                   after_norm_b=F(self.after_norm.bias))
        for k, v in top.items():
            setattr(w, k, v.data_ptr())
        layers = (L.EmConformerLayer * NL)()
        glu_perm = torch.arange(2 * d).reshape(2, d // 16, 16).permute(1, 0, 2).reshape(-1)
        for i, l in enumerate(self.encoders):
            sa, cm = l.self_attn, l.conv_module
            bn = cm.norm
            scale = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
            dw_w = cm.depthwise_conv.weight.double().reshape(d, -1) * scale[:, None]
            dw_b = (cm.depthwise_conv.bias.double() - bn.running_mean.double()) * scale + bn.bias.double()
            lt = dict(
                norm_ff_mac_g=F(l.norm_ff_macaron.weight), norm_ff_mac_b=F(l.norm_ff_macaron.bias),
                norm_mha_g=F(l.norm1.weight), norm_mha_b=F(l.norm1.bias),
                norm_conv_g=F(l.norm_conv.weight), norm_conv_b=F(l.norm_conv.bias),
                norm_ff_g=F(l.norm2.weight), norm_ff_b=F(l.norm2.bias),
                norm_final_g=F(l.norm_final.weight), norm_final_b=F(l.norm_final.bias),
                ffm_w1=A(l.feed_forward_macaron.w_1.weight), ffm_b1=F(l.feed_forward_macaron.w_1.bias),
                ffm_w2=A(l.feed_forward_macaron.w_2.weight), ffm_b2=F(l.feed_forward_macaron.w_2.bias),
                wqkv=A(torch.cat([sa.linear_q.weight, sa.linear_k.weight, sa.linear_v.weight], 0)),
                bqkv=F(torch.cat([sa.linear_q.bias, sa.linear_k.bias, sa.linear_v.bias], 0)),
                wout=A(sa.linear_out.weight), bout=F(sa.linear_out.bias),
                pw1=A(cm.pointwise_conv1.weight.reshape(2 * d, d)[glu_perm]),
                pw1_b=F(cm.pointwise_conv1.bias[glu_perm]),
                dw_w=F(dw_w.t()), dw_b=F(dw_b),
                pw2=A(cm.pointwise_conv2.weight.reshape(d, d)), pw2_b=F(cm.pointwise_conv2.bias),
                ff_w1=A(l.feed_forward.w_1.weight), ff_b1=F(l.feed_forward.w_1.bias),
                ff_w2=A(l.feed_forward.w_2.weight), ff_b2=F(l.feed_forward.w_2.bias))
            if self._fusable():
                lt.update(self._fused_layer(l, A, F))
            for k, v in lt.items():
                setattr(layers[i], k, v.data_ptr())
        w.layers = C.cast(layers, C.POINTER(L.EmConformerLayer))
        pe = abs_pos_table(5000, d).to(dev)  # StreamPositionalEncoding.extend_pe (embedding.py:357-374)
        self._packed = dict(w=w, layers=layers, keep=keep, device=dev, dtype=self.em_dtype, pe=pe, F2=F2)
        return self._packed

    def _fusable(self) -> bool:
        """Shapes the fused streaming layer covers (csrc/streaming.hip `cb_fusable`, csrc/block.hip with EM_BLOCK_RELU):
        bf16, 256 wide, 4 heads, depthwise conv width 15, feed-forward width a multiple of 128 up to 4096 - the
        aishell streaming recipe.  Everything else keeps the thirteen launches per layer."""
        return (bool(getattr(self, "fused", True))  # (`enc.fused = False` + `invalidate()`: A/B and bisecting)
                and self.em_dtype == L.EM_BF16 and self._output_size == 256 and self.heads == 4
                and self.cnn_module_kernel == 15 and self.linear_units % 128 == 0 and self.linear_units <= 4096)

    def _fused_layer(self, l, A, F):
        """Operands of the row-block kernels for one contextual-block layer (include/espnet_amd.h, EmBlockArgs): weights
        in fragment-major units (`pack_k_units` / `pack_w1` / `pack_w2` of the Conformer encoder), bias / LayerNorm
        vectors as groups of EM_BLOCK_PARAM_GROUP floats in the order the kernels consume them.  The first FFN bias
        does not fit a group at ff = 2048: the kernels read it from the row-major `ffm_b1` / `ff_b1` (EmBlockArgs.ffm_b1g /
        ff_b1g) and its slot in the group stays zero."""
        from espnet_amd.asr.encoder.conformer_encoder import pack_k_units, pack_w1, pack_w2

        d, G = self._output_size, L.EM_BLOCK_PARAM_GROUP
        sa, cm = l.self_attn, l.conv_module

        def group(*vecs):
            v = torch.cat([t.detach().to(torch.float32).reshape(-1).cpu() for t in vecs])
            assert v.numel() <= G
            return torch.nn.functional.pad(v, (0, G - v.numel()))

        b1_slot = torch.zeros(1024)
        perm = torch.cat([torch.cat([torch.arange(64 * j, 64 * j + 64), torch.arange(d + 64 * j, d + 64 * j + 64)])
                          for j in range(d // 64)])
        pw1 = cm.pointwise_conv1.weight.reshape(2 * d, d)
        fp_a = torch.cat([group(l.norm_ff_macaron.weight, l.norm_ff_macaron.bias, b1_slot, l.feed_forward_macaron.w_2.bias),
                          group(l.norm1.weight, l.norm1.bias, sa.linear_q.bias, sa.linear_k.bias, sa.linear_v.bias)])
        fp_d = torch.cat([group(cm.pointwise_conv2.bias, l.norm2.weight, l.norm2.bias),
                          group(b1_slot, l.feed_forward.w_2.bias, l.norm_final.weight, l.norm_final.bias),
                          torch.zeros(G)])
        return dict(
            pw1f=A(pack_k_units(pw1[perm])), ffm_w2p=A(pack_w2(l.feed_forward_macaron.w_2.weight)),
            ff_w2p=A(pack_w2(l.feed_forward.w_2.weight)), woutp=A(pack_k_units(sa.linear_out.weight)),
            pw2p=A(pack_k_units(cm.pointwise_conv2.weight.reshape(d, d))),
            ff_w1p=A(pack_w1(l.feed_forward.w_1.weight)), ffm_w1p=A(pack_w1(l.feed_forward_macaron.w_1.weight)),
            wqkvp=A(pack_k_units(torch.cat([sa.linear_q.weight, sa.linear_k.weight, sa.linear_v.weight], 0))),
            fp_c=F(group(sa.linear_out.bias, l.norm_conv.weight, l.norm_conv.bias, cm.pointwise_conv1.bias[perm])),
            fp_da=F(fp_d), fp_a=F(fp_a))

    def _ensure_packed(self, device):
        p = self._packed
        if p is None or p["device"] != device or p["dtype"] != self.em_dtype:
            p = self.pack(device)
        return p

    # ------------------------------------------------------------------ device pieces
    def _embed_device(self, xs: torch.Tensor) -> torch.Tensor:
        """Conv2dSubsamplingWOPosEnc.forward (subsampling_without_posenc.py:44-62); xs (t, idim) f32
        on the GPU -> (t', d) f32."""
        pk = self._ensure_packed(xs.device)
        lib, w = L.load(), pk["w"]
        t, nm = xs.shape
        d = self._output_size
        T1, F1 = (t - 3) // 2 + 1, (nm - 3) // 2 + 1
        T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
        dev, act, st = xs.device, self.act_dtype, L.current_stream_ptr()
        flen = torch.full((1,), t, dtype=torch.int32, device=dev)
        c1 = torch.empty(T1 * F1 * d, dtype=act, device=dev)
        L.check(lib.em_conv2d_sub1(self.em_dtype, L.ptr(xs), None, L.ptr(flen), 1, t, nm, w.conv1_w,
                                   w.conv1_b, d, L.ptr(c1), st), "em_conv2d_sub1")
        c2 = torch.empty(T2 * F2 * d, dtype=act, device=dev)
        a = L.EmGemmArgs(A=c1.data_ptr(), W=w.conv2_w, C=c2.data_ptr(), bias=w.conv2_b, M=T2 * F2, N=d,
                         K=9 * d, lda=0, ldc=d, scale=1.0, T1=T1, F1=F1, T2=T2, F2=F2, d=d)
        L.check(lib.em_gemm(self.em_dtype, L.EM_EPI_RELU, L.EM_A_CONV2, a, st), "em_gemm(conv2)")
        out = torch.empty(T2, d, dtype=torch.float32, device=dev)
        a = L.EmGemmArgs(A=c2.data_ptr(), W=w.embed_w, C=out.data_ptr(), bias=w.embed_b, M=T2, N=d,
                         K=F2 * d, lda=F2 * d, ldc=d, scale=1.0)
        L.check(lib.em_gemm(self.em_dtype, L.EM_EPI_SCALE_F32, L.EM_A_PLAIN, a, st), "em_gemm(embed.out)")
        return out

    def _workspace(self, dev, n_blk, Lb):
        pk = self._ensure_packed(dev)
        need = L.load().em_cb_workspace_bytes(self.em_dtype, C.byref(pk["w"]), n_blk, Lb)
        key = (torch.cuda.current_stream().cuda_stream, n_blk, Lb)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self._ws[key] = ws
        return ws

    def _encode_blocks(self, x: torch.Tensor, mask_mode: int, past_ctx, next_ctx):
        """x (n_blk, L, d) f32 in place."""
        pk = self._ensure_packed(x.device)
        n_blk, Lb, _ = x.shape
        ws = self._workspace(x.device, n_blk, Lb)
        L.check(L.load().em_cb_encode_blocks(self.em_dtype, C.byref(pk["w"]), L.ptr(x), n_blk, Lb,
                                             mask_mode, L.ptr(past_ctx), L.ptr(next_ctx), L.ptr(ws),
                                             ws.numel(), L.current_stream_ptr()), "em_cb_encode_blocks")

    def _after_norm(self, ys: torch.Tensor) -> torch.Tensor:
        pk = self._ensure_packed(ys.device)
        w = pk["w"]
        L.check(L.load().em_layernorm_inplace_f32(L.ptr(ys), w.after_norm_g, w.after_norm_b,
                                                  ys.size(0), ys.size(1), LN_EPS,
                                                  L.current_stream_ptr()), "after_norm")
        return ys

    # ------------------------------------------------------------------ reference entry points
    def forward(self, xs_pad, ilens, prev_states=None, is_final=True, infer_mode=False):
        if not infer_mode:
            raise NotImplementedError("forward_train (full-utterance block processing used in "
                                      "training) is outside the inference hot path")
        return self.forward_infer(xs_pad, ilens, prev_states, is_final)

    @torch.no_grad()
    def _empty_out(self, dev):
        e = self.__dict__.get("_empty_cache")
        if e is None or e[0].device != dev:
            e = self._empty_cache = (torch.zeros(1, 0, self._output_size, device=dev), torch.zeros(1, device=dev))
        return e

    def _olen_out(self, dev, n):
        c = self.__dict__.setdefault("_olen_cache", {})
        t = c.get((dev, n))
        if t is None:
            if len(c) > 256:
                c.clear()
            t = c[(dev, n)] = torch.full((1,), float(n), device=dev)
            torch.cuda.current_stream().synchronize()  # (filled before any stream may read it)
        return t

    def forward_infer(self, xs_pad: torch.Tensor, ilens: torch.Tensor, prev_states=None,
                      is_final: bool = True) -> Tuple[torch.Tensor, torch.Tensor, Optional[dict]]:
        """contextual_block_conformer_encoder.py:386-600.  xs_pad (1, t, idim) f32 ON THE GPU."""
        L.require_gpu(xs_pad, "xs_pad")
        assert xs_pad.size(0) == 1
        dev = xs_pad.device
        pk = self._ensure_packed(dev)
        lib = L.load()
        d, bs, hs, la, sub = self._output_size, self.block_size, self.hop_size, self.look_ahead, self.subsample
        st = prev_states or dict(prev_addin=None, buffer_before_downsampling=None, ilens_buffer=None,
                                 buffer_after_downsampling=None, n_processed_blocks=0,
                                 past_encoder_ctx=None)
        prev_addin, buf_after = st["prev_addin"], st["buffer_after_downsampling"]
        n_proc, past_ctx = st["n_processed_blocks"], st["past_encoder_ctx"]
        xs = xs_pad[0].to(torch.float32)
        if st["buffer_before_downsampling"] is not None:
            xs = torch.cat([st["buffer_before_downsampling"], xs], dim=0)
        empty = self._empty_out(dev)  # (cached: two fills per call otherwise)
        if is_final:
            buf_before = None
        else:
            n_samples = xs.size(0) // sub - 1
            if n_samples < 2:  # :424-438
                return (*empty, dict(st, buffer_before_downsampling=xs,
                                     ilens_buffer=torch.tensor([xs.size(0)])))
            n_res = xs.size(0) % sub + sub * 2
            buf_before = xs[xs.size(0) - n_res:].contiguous()
            xs = xs[: n_samples * sub]
        x = self._embed_device(xs.contiguous())
        if buf_after is not None:
            x = torch.cat([buf_after, x], dim=0)
        total = x.size(0)
        if is_final:
            block_num = math.ceil(float(total - (bs - hs - la) - la) / float(hs))
            buf_after = None
        else:
            if total <= bs:  # :474-487
                return (*empty, dict(prev_addin=prev_addin, buffer_before_downsampling=buf_before,
                                     ilens_buffer=torch.tensor([buf_before.size(0)]),
                                     buffer_after_downsampling=x, n_processed_blocks=n_proc,
                                     past_encoder_ctx=past_ctx))
            overlap = bs - hs
            block_num = max(0, total - overlap) // hs
            res = total - hs * block_num
            buf_after = x[total - res:].contiguous()
            x = x[: block_num * hs + overlap]
        x = x.contiguous()
        stream = L.current_stream_ptr()
        if n_proc == 0 and total <= bs and is_final:  # short utterance (:496-505)
            xc = torch.empty(1, total, d, dtype=torch.float32, device=dev)
            L.check(lib.em_stream_pos_enc_f32(L.ptr(x), L.ptr(pk["pe"]), 0, total, d, L.ptr(xc), stream),
                    "em_stream_pos_enc_f32")
            self._encode_blocks(xc, 0, None, None)
            return self._after_norm(xc[0]).unsqueeze(0), self._olen_out(dev, 0), None
        chunks = torch.empty(block_num, bs + 2, d, dtype=torch.float32, device=dev)
        addin = torch.empty(d, dtype=torch.float32, device=dev)
        n_proc_dev = st.get("n_processed_blocks_dev")  # set by StreamingStepGraph only
        L.check(lib.em_cb_build_blocks_f32(L.ptr(x), L.ptr(pk["pe"]), L.ptr(prev_addin), n_proc,
                                           L.ptr(n_proc_dev), block_num, x.size(0), bs, hs, d,
                                           L.ptr(chunks), L.ptr(addin), stream),
                "em_cb_build_blocks_f32")
        next_ctx = torch.empty(self.num_blocks, d, dtype=torch.float32, device=dev)
        self._encode_blocks(chunks, 1, past_ctx, next_ctx)
        ys_chunk = chunks[:, 1 : bs + 1]
        offset = bs - la - hs
        if is_final:
            y_len = x.size(0) if n_proc == 0 else x.size(0) - offset
        else:
            y_len = block_num * hs + (offset if n_proc == 0 else 0)
        # (not final: the head piece and the blocks' hops below tile [0, y_len) exactly - nothing to clear)
        ys = (torch.zeros if is_final else torch.empty)(y_len, d, dtype=torch.float32, device=dev)
        if n_proc == 0:
            ys[:offset] = ys_chunk[0, :offset]
        for i in range(block_num):  # :565-576 (slicing only)
            cur = i * hs + (offset if n_proc == 0 else 0)
            clen = min(bs - offset, y_len - cur) if (i == block_num - 1 and is_final) else hs
            ys[cur : cur + clen] = ys_chunk[i, offset : offset + clen]
        ys = self._after_norm(ys).unsqueeze(0)
        olen = self._olen_out(dev, y_len)  # (f32 (1,) on the device as the reference returns it; cached per value: read-only)
        if is_final:
            return ys, olen, None
        return ys, olen, dict(prev_addin=addin, buffer_before_downsampling=buf_before,
                              ilens_buffer=torch.tensor([buf_before.size(0)]),
                              buffer_after_downsampling=buf_after,
                              n_processed_blocks=n_proc + block_num, past_encoder_ctx=next_ctx)


    # ------------------------------------------------------------------ a batch of lock-step streams
    def _embed_device_batch(self, xs: torch.Tensor) -> torch.Tensor:
        """Conv2dSubsamplingWOPosEnc.forward for S streams at once: xs (S, t, idim) f32 on the GPU -> (S, t', d)."""
        pk = self._ensure_packed(xs.device)
        lib, w = L.load(), pk["w"]
        S, t, nm = xs.shape
        d = self._output_size
        T1, F1 = (t - 3) // 2 + 1, (nm - 3) // 2 + 1
        T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
        dev, act, st = xs.device, self.act_dtype, L.current_stream_ptr()
        flen = self._flen_cache.get((S, t, dev))  # (a batch's ticks repeat their shapes: filled once, not per tick)
        if flen is None:
            if len(self._flen_cache) > 64:
                self._flen_cache.clear()
            flen = self._flen_cache[(S, t, dev)] = torch.full((S,), t, dtype=torch.int32, device=dev)
            torch.cuda.current_stream().synchronize()  # (filled before another stream's tick may read it)
        c1 = torch.empty(S * T1 * F1 * d, dtype=act, device=dev)
        L.check(lib.em_conv2d_sub1(self.em_dtype, L.ptr(xs), None, L.ptr(flen), S, t, nm, w.conv1_w,
                                   w.conv1_b, d, L.ptr(c1), st), "em_conv2d_sub1")
        c2 = torch.empty(S * T2 * F2 * d, dtype=act, device=dev)
        a = L.EmGemmArgs(A=c1.data_ptr(), W=w.conv2_w, C=c2.data_ptr(), bias=w.conv2_b, M=S * T2 * F2, N=d,
                         K=9 * d, lda=0, ldc=d, scale=1.0, T1=T1, F1=F1, T2=T2, F2=F2, d=d)
        L.check(lib.em_gemm(self.em_dtype, L.EM_EPI_RELU, L.EM_A_CONV2, a, st), "em_gemm(conv2)")
        out = torch.empty(S, T2, d, dtype=torch.float32, device=dev)
        a = L.EmGemmArgs(A=c2.data_ptr(), W=w.embed_w, C=out.data_ptr(), bias=w.embed_b, M=S * T2, N=d,
                         K=F2 * d, lda=F2 * d, ldc=d, scale=1.0)
        L.check(lib.em_gemm(self.em_dtype, L.EM_EPI_SCALE_F32, L.EM_A_PLAIN, a, st), "em_gemm(embed.out)")
        return out

    @torch.no_grad()
    def forward_infer_batch(self, xs_pad: torch.Tensor, prev_states=None, is_final: bool = False):
        """`forward_infer` (contextual_block_conformer_encoder.py:386-600) for S streams whose carried buffers have the
        same SHAPES: every stream is fed a chunk of the same length at this call (a server batching its live
        connections) and they agree on the buffer lengths and on whether they have processed a block yet; the NUMBER
        of blocks a stream has processed may differ per stream (`n_processed_blocks` a list of S ints: streams that
        joined at different times - it only moves the positional-encoding offsets, em_cb_build_blocks_rows_f32).
        Streams in different phases are grouped by `espnet_amd.bin.asr_inference_streaming.StreamPool`.
        xs_pad (S, t, idim) f32 ON THE GPU.  Returns (ys (S, t_out, d), t_out, state); row s equals what
        `forward_infer` returns for stream s alone (tests/test_gpu_streaming.py::test_batch_of_streams).  The dense
        operators of a call see S * n_blk independent blocks - one launch sequence for all streams."""
        L.require_gpu(xs_pad, "xs_pad")
        dev = xs_pad.device
        pk = self._ensure_packed(dev)
        lib = L.load()
        S = xs_pad.size(0)
        d, bs, hs, la, sub = self._output_size, self.block_size, self.hop_size, self.look_ahead, self.subsample
        st = prev_states or dict(prev_addin=None, buffer_before_downsampling=None, buffer_after_downsampling=None,
                                 n_processed_blocks=0, past_encoder_ctx=None)
        prev_addin, buf_after = st["prev_addin"], st["buffer_after_downsampling"]
        n_proc, past_ctx = st["n_processed_blocks"], st["past_encoder_ctx"]
        n_rows = None  # per-stream block counts (all zero or all positive: the callers group streams that way)
        if isinstance(n_proc, (list, tuple)):
            n_rows = [int(v) for v in n_proc]
            if len(n_rows) != S or (min(n_rows) == 0) != (max(n_rows) == 0):
                raise ValueError("n_processed_blocks: one count per stream, all zero or all positive")
            n_proc = n_rows[0] if len(set(n_rows)) == 1 else (1 if n_rows[0] > 0 else 0)
            if len(set(n_rows)) == 1:
                n_rows = None
        xs = xs_pad.to(torch.float32)
        if st["buffer_before_downsampling"] is not None:
            xs = torch.cat([st["buffer_before_downsampling"], xs], dim=1)
        empty = xs.new_zeros(S, 0, d)
        if is_final:
            buf_before = None
        else:
            n_samples = xs.size(1) // sub - 1
            if n_samples < 2:  # :424-438
                return empty, 0, dict(st, buffer_before_downsampling=xs)
            n_res = xs.size(1) % sub + sub * 2
            buf_before = xs[:, xs.size(1) - n_res:].contiguous()
            xs = xs[:, : n_samples * sub]
        x = self._embed_device_batch(xs.contiguous())
        if buf_after is not None:
            x = torch.cat([buf_after, x], dim=1)
        total = x.size(1)
        if is_final:
            block_num = math.ceil(float(total - (bs - hs - la) - la) / float(hs))
            buf_after = None
        else:
            if total <= bs:  # :474-487
                return empty, 0, dict(prev_addin=prev_addin, buffer_before_downsampling=buf_before,
                                      buffer_after_downsampling=x, n_processed_blocks=st["n_processed_blocks"],
                                      past_encoder_ctx=past_ctx)
            overlap = bs - hs
            block_num = max(0, total - overlap) // hs
            res = total - hs * block_num
            buf_after = x[:, total - res:].contiguous()
            x = x[:, : block_num * hs + overlap]
        x = x.contiguous()
        stream = L.current_stream_ptr()
        if n_proc == 0 and total <= bs and is_final:  # short utterances (:496-505): no context slots
            xc = torch.empty(S, total, d, dtype=torch.float32, device=dev)
            for s_ in range(S):  # (rare path: one launch per stream)
                L.check(lib.em_stream_pos_enc_f32(L.ptr(x[s_]), L.ptr(pk["pe"]), 0, total, d, L.ptr(xc[s_]), stream),
                        "em_stream_pos_enc_f32")
            ws = self._workspace(dev, S, total)
            L.check(lib.em_cb_encode_blocks(self.em_dtype, C.byref(pk["w"]), L.ptr(xc), S, total, 0, None, None,
                                            L.ptr(ws), ws.numel(), stream), "em_cb_encode_blocks")
            return self._after_norm(xc.view(S * total, d)).view(S, total, d), total, None
        Lb = bs + 2
        chunks = torch.empty(S, block_num, Lb, d, dtype=torch.float32, device=dev)
        addin = torch.empty(S, d, dtype=torch.float32, device=dev)
        rows_static = st.get("n_processed_blocks_dev")  # (S,) int32 on the device: set by a captured tick only (BatchTickGraph)
        if rows_static is not None:
            L.check(lib.em_cb_build_blocks_rows_f32(L.ptr(x), L.ptr(pk["pe"]), L.ptr(prev_addin), L.ptr(rows_static), S,
                                                    block_num, x.size(1), bs, hs, d, L.ptr(chunks), L.ptr(addin), stream),
                    "em_cb_build_blocks_rows_f32")
        elif n_rows is None:
            L.check(lib.em_cb_build_blocks_batch_f32(L.ptr(x), L.ptr(pk["pe"]), L.ptr(prev_addin), n_proc, S, block_num,
                                                     x.size(1), bs, hs, d, L.ptr(chunks), L.ptr(addin), stream),
                    "em_cb_build_blocks_batch_f32")
        else:
            rows_dev = torch.tensor(n_rows, dtype=torch.int32).to(dev, non_blocking=True)
            L.check(lib.em_cb_build_blocks_rows_f32(L.ptr(x), L.ptr(pk["pe"]), L.ptr(prev_addin), L.ptr(rows_dev), S,
                                                    block_num, x.size(1), bs, hs, d, L.ptr(chunks), L.ptr(addin), stream),
                    "em_cb_build_blocks_rows_f32")
        next_ctx = torch.empty(S, self.num_blocks, d, dtype=torch.float32, device=dev)
        ws = self._workspace(dev, S * block_num, Lb)
        L.check(lib.em_cb_encode_blocks_batch(self.em_dtype, C.byref(pk["w"]), L.ptr(chunks), S, block_num, Lb, 1,
                                              L.ptr(past_ctx), L.ptr(next_ctx), L.ptr(ws), ws.numel(), stream),
                "em_cb_encode_blocks_batch")
        ys_chunk = chunks[:, :, 1 : bs + 1]
        offset = bs - la - hs
        if is_final:
            y_len = x.size(1) if n_proc == 0 else x.size(1) - offset
        else:
            y_len = block_num * hs + (offset if n_proc == 0 else 0)
        # (not final: the head piece and the blocks' hops below tile [0, y_len) exactly - nothing to clear)
        ys = (torch.zeros if is_final else torch.empty)(S, y_len, d, dtype=torch.float32, device=dev)
        if n_proc == 0:
            ys[:, :offset] = ys_chunk[:, 0, :offset]
        for i in range(block_num):  # :565-576 (slicing only)
            cur = i * hs + (offset if n_proc == 0 else 0)
            clen = min(bs - offset, y_len - cur) if (i == block_num - 1 and is_final) else hs
            ys[:, cur : cur + clen] = ys_chunk[:, i, offset : offset + clen]
        ys = self._after_norm(ys.view(S * y_len, d)).view(S, y_len, d)
        if is_final:
            return ys, y_len, None
        n_next = n_proc + block_num if n_rows is None else [v + block_num for v in n_rows]
        return ys, y_len, dict(prev_addin=addin, buffer_before_downsampling=buf_before,
                               buffer_after_downsampling=buf_after, n_processed_blocks=n_next,
                               past_encoder_ctx=next_ctx)


class StreamingStepGraph:
    """hipGraph replay of the steady-state streaming step (BASELINE config 5).

    Feeding fixed-size chunks, `forward_infer` reaches a steady state after a few calls: the carried
    buffers keep their shapes and every call processes the same number of blocks, so the ~200
    kernel launches of a call are identical except for the positional-encoding offset (read from
    device memory).  This wrapper runs the encoder eagerly until two consecutive calls have the same
    signature, captures the next call into a hipGraph (torch.cuda.CUDAGraph = hipGraph on ROCm)
    over static input / state buffers, and from then on replays it: one graph launch per chunk.
    `is_final` calls and any call whose chunk size differs fall back to the eager path.
    """

    def __init__(self, encoder: ContextualBlockConformerEncoder, chunk_frames: int = 0):
        # chunk_frames is only a hint: the graph is captured for whatever chunk size repeats
        self.enc, self.chunk, self.last_size = encoder, chunk_frames, -1
        self.state, self.graph, self.graph_sig = None, None, None
        self.in_graph_state = False
        self.n_replays = 0

    @staticmethod
    def _signature(st):
        return (tuple(st["buffer_before_downsampling"].shape), tuple(st["buffer_after_downsampling"].shape),
                st["prev_addin"] is not None, st["past_encoder_ctx"] is not None)

    def reset(self):
        self.state = None  # the captured graph stays valid for the next utterance
        self.in_graph_state = False

    def _capture(self, feats):
        st = self.state
        dev = feats.device
        self.s_in = feats.clone()
        self.s_state = dict(
            prev_addin=st["prev_addin"].clone(),
            buffer_before_downsampling=st["buffer_before_downsampling"].clone(),
            ilens_buffer=st["ilens_buffer"],
            buffer_after_downsampling=st["buffer_after_downsampling"].clone(),
            n_processed_blocks=1,  # > 0: steady state; the real count lives on the device
            n_processed_blocks_dev=torch.tensor([st["n_processed_blocks"]], dtype=torch.int32, device=dev),
            past_encoder_ctx=st["past_encoder_ctx"].clone())
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up on the capture stream (allocator, workspaces)
            self.enc.forward_infer(self.s_in[None], torch.tensor([self.chunk]), dict(self.s_state), False)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ys, _, nst = self.enc.forward_infer(self.s_in[None], torch.tensor([self.chunk]),
                                                dict(self.s_state), False)
            # carry the state forward inside the graph (static buffers, same shapes)
            self.s_state["prev_addin"].copy_(nst["prev_addin"])
            self.s_state["buffer_before_downsampling"].copy_(nst["buffer_before_downsampling"])
            self.s_state["buffer_after_downsampling"].copy_(nst["buffer_after_downsampling"])
            self.s_state["past_encoder_ctx"].copy_(nst["past_encoder_ctx"])
            self.s_state["n_processed_blocks_dev"].add_(nst["n_processed_blocks"] - 1)
        self.graph, self.s_out = g, ys
        self.blocks_per_call = nst["n_processed_blocks"] - 1

    def _load_static_state(self):
        st = self.state
        for k in ("prev_addin", "buffer_before_downsampling", "buffer_after_downsampling", "past_encoder_ctx"):
            self.s_state[k].copy_(st[k])
        self.s_state["n_processed_blocks_dev"].fill_(st["n_processed_blocks"])

    @torch.no_grad()
    def __call__(self, feats: torch.Tensor, is_final: bool = False):
        """feats (t, idim) f32 on the GPU.  Returns ys (t_out, d) f32 (a view of a static buffer
        when replayed: consume or clone it before the next call)."""
        if self.graph is None and feats.size(0) == self.last_size:
            self.chunk = feats.size(0)  # a repeating chunk size: this is the one worth capturing
        self.last_size = feats.size(0)
        steady = (not is_final and feats.size(0) == self.chunk and self.state is not None
                  and self.state["past_encoder_ctx"] is not None
                  and self.state["buffer_after_downsampling"] is not None)
        if steady and self.graph is not None and self._signature(self.state) == self.graph_sig:
            if not self.in_graph_state:
                self._load_static_state()
                self.in_graph_state = True
            self.s_in.copy_(feats)
            self.graph.replay()
            self.n_replays += 1
            self.state["n_processed_blocks"] += self.blocks_per_call
            return self.s_out[0]
        if self.in_graph_state:  # leave graph mode: pull the state back out
            for k in ("prev_addin", "buffer_before_downsampling", "buffer_after_downsampling", "past_encoder_ctx"):
                self.state[k] = self.s_state[k].clone()
            self.in_graph_state = False
        prev_sig = self._signature(self.state) if steady else None
        ys, _, nst = self.enc.forward_infer(feats[None], torch.tensor([feats.size(0)]), self.state, is_final)
        self.state = nst
        if (steady and self.graph is None and nst is not None and prev_sig == self._signature(nst)):
            self.graph_sig = prev_sig
            self._capture(feats)
            self.in_graph_state = False
        return ys[0]
