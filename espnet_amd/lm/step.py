"""One language-model step per call: the reference's `batch_score(ys, states, xs)` of the LM scorers
(espnet2/lm/transformer_lm.py:103-137, espnet2/lm/seq_rnn_lm.py:140-177) on the device code of the fused
search (`em_lm_step`, csrc/search.hip `lm_step` / `rnn_lm_step`).  The caller describes the n hypotheses as a
search state of one utterance with n independent rows (identity ancestor / parent tables)."""
import ctypes as C

import torch

from espnet_amd import lib as L

_ACT = {"lm_e", "lm_xn", "lm_qkv", "lm_ctx", "lm_h", "lm_k", "lm_v", "rnn_hs", "rnn_hin"}


def lm_step(lm, dev, n: int, Lmax: int, i: int, tok: torch.Tensor, preset: dict, log_softmax: bool = True):
    """tok [Lmax][n] int32 token table (row i = the tokens consumed now); preset = already filled buffers
    (lm_k / lm_v or rnn_hs / rnn_cs).  Returns (log-probs (n, V) f32, dict of all buffers)."""
    pk = lm.ensure_packed(dev, Lmax + 1)
    V = lm.vocab_size
    act = torch.bfloat16 if lm.em_dtype == L.EM_BF16 else torch.float32
    t = dict(preset)
    for name, shp in lm.search_buffers(n, V, Lmax, 1, 1).items():
        if name in t or name in ("run_slm", "end_slm"):
            continue
        t[name] = torch.zeros(shp, dtype=act if name in _ACT else torch.float32, device=dev)
    ident = torch.arange(n, dtype=torch.int32, device=dev)
    t["tok"] = tok
    t["anc_a"] = t["anc_b"] = ident.unsqueeze(1).expand(n, Lmax).contiguous()
    t["parent"] = ident.unsqueeze(0).expand(Lmax, n).contiguous()
    p = L.EmSearchParams(B=1, W=n, V=V, T=1, Tpad=32, S=V, NC=V, Lmax=Lmax, end_cap=1, sos=V - 1, eos=V - 1,
                         blank=0, use_end_detect=0, w_dec=0.0, w_ctc=0.0, w_len=0.0, w_lm=1.0, ldT=0)
    bs = L.EmSearchBuffers()
    for name, v in t.items():
        setattr(bs, name, v.data_ptr())
    bs.lm = C.addressof(pk["w"])
    lib = L.load()
    L.check(lib.em_lm_step(lm.em_dtype, C.byref(p), C.byref(bs), i, L.current_stream_ptr()), "em_lm_step")
    logp = t["lm_logp"]
    if log_softmax:
        L.check(lib.em_log_softmax_rows_f32(L.ptr(logp), n, V, L.current_stream_ptr()),
                "em_log_softmax_rows_f32")
    return logp, t
