"""CTC head on the MI355X.  Mirrors espnet2/asr/ctc.py:9-215 for inference (`ctc_lo`,
`softmax`/`log_softmax`/`argmax`); state-dict keys `ctc_lo.{weight,bias}`.  The loss is training
only and out of scope."""
import os

import torch

from espnet_amd import lib as L


class CTC(torch.nn.Module):
    def __init__(self, odim: int, encoder_output_size: int, dropout_rate: float = 0.0,
                 ctc_type: str = "builtin", reduce: bool = True, ignore_nan_grad: bool = None,
                 zero_infinity: bool = True, brctc_risk_strategy: str = "exp",
                 brctc_group_strategy: str = "end", brctc_risk_factor: float = 0.0,
                 compute_dtype: str = "bfloat16"):
        super().__init__()
        self.odim, self.eprojs = odim, encoder_output_size
        self.dropout_rate = dropout_rate
        self.ctc_lo = torch.nn.Linear(encoder_output_size, odim)  # parameter container
        self.ctc_type = ctc_type
        self.compute_dtype = compute_dtype
        self._packed = None

    @property
    def em_dtype(self):
        return L.DTYPES[self.compute_dtype]

    @property
    def act_dtype(self):
        return torch.bfloat16 if self.em_dtype == L.EM_BF16 else torch.float32

    def invalidate(self):
        self._packed = None

    def _pack(self, device):
        p = self._packed
        if p is None or p["device"] != device or p["dtype"] != self.em_dtype:
            p = dict(w=self.ctc_lo.weight.detach().float().contiguous().to(self.act_dtype).to(device),
                     b=self.ctc_lo.bias.detach().float().contiguous().to(device), device=device,
                     dtype=self.em_dtype)
            self._packed = p
        return p

    def _to_act(self, hs_pad: torch.Tensor) -> torch.Tensor:
        L.require_gpu(hs_pad, "hs_pad")
        if hs_pad.dtype == self.act_dtype:
            return hs_pad.contiguous()
        src = hs_pad.to(torch.float32).contiguous()
        dst = torch.empty(src.shape, dtype=self.act_dtype, device=src.device)
        L.check(L.load().em_cast_f32(self.em_dtype, L.ptr(src), src.numel(), L.ptr(dst),
                                     L.current_stream_ptr()), "em_cast_f32")
        return dst

    def logits_device(self, enc_act: torch.Tensor) -> torch.Tensor:
        """enc_act (B,T,d) in the compute dtype -> logits (B,T,V) f32."""
        p = self._pack(enc_act.device)
        B, T, d = enc_act.shape
        out = torch.empty(B, T, self.odim, dtype=torch.float32, device=enc_act.device)
        a = L.EmGemmArgs(A=enc_act.data_ptr(), W=p["w"].data_ptr(), C=out.data_ptr(),
                         bias=p["b"].data_ptr(), M=B * T, N=self.odim, K=d, lda=d, ldc=self.odim,
                         scale=1.0)
        L.check(L.load().em_gemm(self.em_dtype, L.EM_EPI_STORE_F32, L.EM_A_PLAIN, a,
                                 L.current_stream_ptr()), "em_gemm(ctc_lo)")
        return out

    def log_softmax(self, hs_pad: torch.Tensor) -> torch.Tensor:
        """asr/ctc.py:197-205."""
        logits = self.logits_device(self._to_act(hs_pad))
        B, T, V = logits.shape
        L.check(L.load().em_log_softmax_rows_f32(L.ptr(logits), B * T, V, L.current_stream_ptr()),
                "em_log_softmax_rows_f32")
        return logits

    def argmax(self, hs_pad: torch.Tensor, as_int32: bool = False) -> torch.Tensor:
        """asr/ctc.py:207-215.  Returns (B, T) int64 like the reference (`as_int32`: the kernel's own int32 ids, for
        callers that read them back to the host anyway - one conversion launch less per streaming tick)."""
        act = self._to_act(hs_pad)
        B, T, d = act.shape
        V, dev, st = self.odim, act.device, L.current_stream_ptr()
        ids = torch.empty(B, T, dtype=torch.int32, device=dev)
        if d % 64 != 0 or os.environ.get("ESPNET_AMD_CTC_ARGMAX_LOGITS"):
            # the (B, T, V) f32 logits written and read back (rounds 1-4; developer A/B switch; widths the arg-max
            # epilogue's GEMM does not take)
            logits = self.logits_device(act)
            L.check(L.load().em_argmax_rows_f32(L.ptr(logits), B * T, V, L.ptr(ids), st), "em_argmax_rows_f32")
            return ids if as_int32 else ids.to(torch.int64)
        # round 5: arg-max in the epilogue of the ctc_lo GEMM, as em_ctc_greedy has it - the logits never exist (10 MB per
        # tick of a 32-stream batch, 26.5 + 26.1 us of its 1.25 ms: profiles/r05x_stream_batch32_kernel_stats.csv); only
        # (value, column) pairs per 64 columns are written and reduced.  Same values compared, ties to the lowest column.
        p = self._pack(dev)
        G = 2 * ((V + 127) // 128)
        part = torch.empty(B * T * G * 2, dtype=torch.float32, device=dev)
        a = L.EmGemmArgs(A=act.data_ptr(), W=p["w"].data_ptr(), C=part.data_ptr(), bias=p["b"].data_ptr(), M=B * T, N=V,
                         K=d, lda=d, ldc=G, scale=1.0)
        L.check(L.load().em_gemm(self.em_dtype, L.EM_EPI_ARGMAX_PART, L.EM_A_PLAIN, a, st), "em_gemm(ctc_lo, arg-max)")
        L.check(L.load().em_argmax_partials(L.ptr(part), B * T, G, L.ptr(ids), st), "em_argmax_partials")
        return ids if as_int32 else ids.to(torch.int64)

    @staticmethod
    def _token_outputs(B, T, dev, out):
        """(tokens (B,T) i32, token_lens (B,) i32): fresh tensors, or the caller's (`out`: e.g. the current slot of an
        `espnet_amd.distributed.RecordRing`, so the records are written where the collation reads them)."""
        if out is None:
            return torch.empty(B, T, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)
        tokens, tlens = out
        if (tuple(tokens.shape) != (B, T) or tuple(tlens.shape) != (B,) or tokens.dtype != torch.int32
                or tlens.dtype != torch.int32 or not tokens.is_contiguous() or not tlens.is_contiguous()):
            raise ValueError(f"out must be contiguous int32 ({B}, {T}) and ({B},) tensors")
        L.require_gpu(tokens, "out tokens")
        return tokens, tlens

    def greedy_device(self, enc_act: torch.Tensor, olens_dev: torch.Tensor, blank: int, sos_eos: int, out=None):
        """Fused G1 path (bin/asr_inference.py:574-575): returns (ids (B,T) i32, tokens (B,T) i32
        padded with -1, token_lens (B,) i32), all on the device, no host sync."""
        p = self._pack(enc_act.device)
        B, T, d = enc_act.shape
        dev = enc_act.device
        logits = torch.empty(B * T, self.odim, dtype=torch.float32, device=dev)
        ids = torch.empty(B, T, dtype=torch.int32, device=dev)
        tokens, tlens = self._token_outputs(B, T, dev, out)
        L.check(L.load().em_ctc_greedy(self.em_dtype, L.ptr(enc_act), L.ptr(p["w"]), L.ptr(p["b"]),
                                       B, T, d, self.odim, L.ptr(olens_dev), blank, sos_eos,
                                       L.ptr(logits), L.ptr(ids), L.ptr(tokens), L.ptr(tlens),
                                       L.current_stream_ptr()), "em_ctc_greedy")
        return ids, tokens, tlens

    def collapse_device(self, ids: torch.Tensor, olens_dev: torch.Tensor, blank: int, sos_eos: int, out=None):
        """groupby + drop blank / <sos/eos> (bin/asr_inference.py:574-575) over given per-frame ids (B, T) i32:
        returns (ids, tokens padded with -1, token_lens), no host sync."""
        B, T = ids.shape
        tokens, tlens = self._token_outputs(B, T, ids.device, out)
        L.check(L.load().em_ctc_collapse(L.ptr(ids), L.ptr(olens_dev), B, T, blank, sos_eos, L.ptr(tokens),
                                         L.ptr(tlens), L.current_stream_ptr()), "em_ctc_collapse")
        return ids, tokens, tlens
