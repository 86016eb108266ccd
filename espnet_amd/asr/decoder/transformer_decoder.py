"""TransformerDecoder (attention decoder used as a beam-search scorer) on the MI355X.

Mirrors espnet2/asr/decoder/transformer_decoder.py:393-468 (constructor) and :191-311
(`forward_one_step`, `score`, `batch_score`), with the reference's state-dict keys
(`embed.0`, `decoders.N.{self_attn,src_attn,feed_forward,norm1,norm2,norm3}`, `after_norm`,
`output_layer`).  The torch.nn layers are parameter containers only.

The arithmetic is csrc/decoder.hip + csrc/gemm.hip + csrc/norm.hip, driven per search step by
csrc/search.hip (`em_search_steps`); `pack()` repacks the reference-layout parameters once
(q|k|v rows concatenated for self-attention, k|v for source attention, absolute sinusoid table).
Inside the fused device search (espnet_amd/nets/batch_beam_search.py) the decoder step never returns to
Python.  The reference's scorer interface (`init_state`, `batch_init_state`, `select_state`, `score`,
`batch_score`, `final_score`; legacy/nets/scorer_interface.py:29-122) is implemented on the same device code
through `em_decoder_memory` + `em_decoder_step`, one call per search step, so the reference's own
BatchBeamSearch can be driven by this class.
"""
import ctypes as C
import math
from typing import List

import torch

from espnet_amd import lib as L
from espnet_amd.asr.encoder.conformer_encoder import LayerNorm, _PositionwiseFeedForward
from espnet_amd.nets.scorer_interface import BatchScorerInterface


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


class TransformerDecoder(torch.nn.Module, BatchScorerInterface):
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
        self._mem = None

    def invalidate(self):
        self._packed = None
        self._mem = None

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
        if act == torch.bfloat16 and self.d % 32 == 0:
            top["out_w_frag"] = A(L.pack_frag16(self.output_layer.weight.detach(), pad_rows=512))
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
            if act == torch.bfloat16 and self.d % 32 == 0 and self.linear_units % 32 == 0:
                # fragment-major copies for the one-launch feed-forward of the label step (csrc/dec_ffn.hip)
                lt["w1_frag"] = A(L.pack_frag16(ff.w_1.weight.detach()))
                lt["w2_frag"] = A(L.pack_frag16(ff.w_2.weight.detach()))
                lt["self_wqkv_frag"] = A(L.pack_frag16(torch.cat([sa.linear_q.weight, sa.linear_k.weight,
                                                                  sa.linear_v.weight], 0).detach(), pad_rows=512))
                lt["self_wout_frag"] = A(L.pack_frag16(sa.linear_out.weight.detach()))
                lt["src_wout_frag"] = A(L.pack_frag16(ca.linear_out.weight.detach()))
                lt["src_wq_frag"] = A(L.pack_frag16(ca.linear_q.weight.detach()))
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

    # ------------------------------------------------------------------ scorer interface (one call per step)
    def init_state(self, x: torch.Tensor):
        """transformer_decoder.py:249-251 (BatchScorerInterface default): no state before the first token."""
        self._mem = None
        return None

    def batch_init_state(self, x: torch.Tensor):
        self._mem = None  # a new utterance: drop the projected memory of the previous one
        return None

    def select_state(self, state, i: int, new_id: int = None):
        """scorer_interface.py:41-52: the state of hypothesis i (full scorers ignore new_id)."""
        return None if state is None else state[i]

    def final_score(self, state) -> float:
        return 0.0

    def _memory(self, xs: torch.Tensor, pk):
        """Source-attention K | V and V^T of the memory, once per memory tensor (the reference recomputes
        `linear_k/linear_v(memory)` for every hypothesis, layer and step).  xs (n, T, d): hypotheses of ONE
        utterance arrive as an expanded view (stride 0, batch_beam_search.py:283-285) and share one memory."""
        n, T, d = xs.shape
        shared = n == 1 or xs.stride(0) == 0
        base = xs[0:1] if shared else xs
        key = (base.data_ptr(), tuple(base.shape), base._version, base.dtype, pk["dtype"], n if not shared else 0)
        if self._mem is not None and self._mem["key"] == key:
            return self._mem
        dev, act = xs.device, self.act_dtype
        B = base.size(0)
        Tpad = (T + 31) // 32 * 32
        src = base.to(torch.float32).contiguous()
        enc = torch.empty(src.shape, dtype=act, device=dev)
        lib = L.load()
        L.check(lib.em_cast_f32(self.em_dtype, L.ptr(src), src.numel(), L.ptr(enc), L.current_stream_ptr()),
                "em_cast_f32")
        mem_kv = torch.empty(self.num_blocks, B * T, 2 * d, dtype=act, device=dev)
        mem_vT = torch.zeros(self.num_blocks, B, d, Tpad, dtype=act, device=dev)
        L.check(lib.em_decoder_memory(self.em_dtype, C.byref(pk["w"]), L.ptr(enc), B, T, Tpad, L.ptr(mem_kv),
                                      L.ptr(mem_vT), L.current_stream_ptr()), "em_decoder_memory")
        # `base` is kept alive with the entry: its address cannot be handed to another tensor meanwhile
        self._mem = dict(key=key, base=base, B=B, T=T, Tpad=Tpad, mem_kv=mem_kv, mem_vT=mem_vT, shared=shared,
                         xlens=torch.full((B,), T, dtype=torch.int32, device=dev))
        return self._mem

    @torch.no_grad()
    def batch_score(self, ys: torch.Tensor, states, xs: torch.Tensor):
        """transformer_decoder.py:270-311.  ys (n, L) int64 prefixes incl. <sos>; states list[n] of None or
        this class's opaque per-hypothesis cache (self-attention K/V of the prefix, (2, layers, L-1, d) in the
        compute dtype); xs (n, T, d) encoder output on the GPU.  Returns (log-probs (n, V) f32, states list[n])."""
        L.require_gpu(xs, "xs")
        dev = xs.device
        n, Lc = ys.shape
        pos = Lc - 1
        pk = self.ensure_packed(dev, pos + 2)
        mem = self._memory(xs, pk)
        B, W = (1, n) if mem["shared"] else (n, 1)
        Lmax = pos + 1
        nl, d, ff, V, act = self.num_blocks, self.d, self.linear_units, self.vocab_size, self.act_dtype
        kv = torch.empty(2, nl, Lmax, n, d, dtype=act, device=dev)
        if pos > 0:
            if states is None or any(s is None for s in states):
                raise ValueError("batch_score: a prefix longer than <sos> needs the state of its previous step")
            kv[:, :, :pos] = torch.stack(list(states), 0).permute(1, 2, 3, 0, 4)
        tok = ys.t().to(device=dev, dtype=torch.int32).contiguous()
        anc = torch.arange(n, dtype=torch.int32, device=dev).unsqueeze(1).expand(n, Lmax).contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        x = torch.empty(n, d, **f32)
        xn, qs, ctx = (torch.empty(n, d, dtype=act, device=dev) for _ in range(3))
        qkv = torch.empty(n, 3 * d, dtype=act, device=dev)
        hbuf = torch.empty(n, ff, dtype=act, device=dev)
        logits = torch.empty(n, V, **f32)
        a = L.EmDecoderStepArgs(B=B, W=W, T=mem["T"], Tpad=mem["Tpad"], Lmax=Lmax, pos=pos, tok=L.ptr(tok),
                                anc=L.ptr(anc), xlens=L.ptr(mem["xlens"]), self_k=kv[0].data_ptr(),
                                self_v=kv[1].data_ptr(), mem_kv=L.ptr(mem["mem_kv"]), mem_vT=L.ptr(mem["mem_vT"]),
                                x=L.ptr(x), xn=L.ptr(xn), qkv=L.ptr(qkv), qs=L.ptr(qs), ctx=L.ptr(ctx),
                                hbuf=L.ptr(hbuf), logits=L.ptr(logits))
        lib = L.load()
        L.check(lib.em_decoder_step(self.em_dtype, C.byref(pk["w"]), C.byref(a), L.current_stream_ptr()),
                "em_decoder_step")
        L.check(lib.em_log_softmax_rows_f32(L.ptr(logits), n, V, L.current_stream_ptr()), "em_log_softmax_rows_f32")
        out = kv.permute(3, 0, 1, 2, 4)  # (n, 2, layers, L, d) views of the one cache tensor
        return logits, [out[r] for r in range(n)]

    @torch.no_grad()
    def forward(self, hs_pad: torch.Tensor, hlens: torch.Tensor, ys_in_pad: torch.Tensor, ys_in_lens: torch.Tensor):
        """transformer_decoder.py:100-189 (AbsDecoder.forward, inference only: no autograd): hs_pad (B, T, d),
        hlens (B,), ys_in_pad (B, L) int64 starting with <sos> -> (scores before softmax (B, L, V) f32,
        ys_in_lens).  The causal mask makes position j depend on tokens <= j only, so the sequence is fed
        position by position through the step kernels over one K/V cache."""
        L.require_gpu(hs_pad, "hs_pad")
        dev = hs_pad.device
        B, T, d = hs_pad.shape
        Lc = ys_in_pad.size(1)
        pk = self.ensure_packed(dev, Lc + 1)
        act, nl, ff, V = self.act_dtype, self.num_blocks, self.linear_units, self.vocab_size
        self._mem = None
        mem = self._memory(hs_pad if B > 1 else hs_pad[0:1], pk)
        mem["xlens"] = torch.as_tensor(hlens).to(device=dev, dtype=torch.int32).contiguous()
        kv = torch.empty(2, nl, Lc, B, d, dtype=act, device=dev)
        tok = ys_in_pad.t().to(device=dev, dtype=torch.int32).clamp_(0, V - 1).contiguous()
        anc = torch.arange(B, dtype=torch.int32, device=dev).unsqueeze(1).expand(B, Lc).contiguous()
        x = torch.empty(B, d, dtype=torch.float32, device=dev)
        xn, qs, ctx = (torch.empty(B, d, dtype=act, device=dev) for _ in range(3))
        qkv = torch.empty(B, 3 * d, dtype=act, device=dev)
        hbuf = torch.empty(B, ff, dtype=act, device=dev)
        out = torch.empty(B, Lc, V, dtype=torch.float32, device=dev)
        logits = torch.empty(B, V, dtype=torch.float32, device=dev)
        lib = L.load()
        for pos in range(Lc):
            a = L.EmDecoderStepArgs(B=B, W=1, T=T, Tpad=mem["Tpad"], Lmax=Lc, pos=pos, tok=L.ptr(tok), anc=L.ptr(anc),
                                    xlens=L.ptr(mem["xlens"]), self_k=kv[0].data_ptr(), self_v=kv[1].data_ptr(),
                                    mem_kv=L.ptr(mem["mem_kv"]), mem_vT=L.ptr(mem["mem_vT"]), x=L.ptr(x),
                                    xn=L.ptr(xn), qkv=L.ptr(qkv), qs=L.ptr(qs), ctx=L.ptr(ctx), hbuf=L.ptr(hbuf),
                                    logits=L.ptr(logits))
            L.check(lib.em_decoder_step(self.em_dtype, C.byref(pk["w"]), C.byref(a), L.current_stream_ptr()),
                    "em_decoder_step")
            out[:, pos] = logits
        self._mem = None
        return out, ys_in_lens

    def score(self, ys: torch.Tensor, state, x: torch.Tensor):
        """transformer_decoder.py:253-268: one hypothesis.  ys (L,), x (T, d)."""
        logp, st = self.batch_score(ys.unsqueeze(0), [state], x.unsqueeze(0))
        return logp[0], st[0]
