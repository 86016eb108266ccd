#!/usr/bin/env python3
"""em_ln_gemm vs em_layernorm + em_gemm on the decoder-step shapes (developer tool; run on the GPU box)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from espnet_amd import lib as L

lib = L.load()


def t(fn, iters=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for M, N, K, epi, name in [(160, 1536, 512, L.EM_EPI_STORE, "self qkv"), (160, 512, 512, L.EM_EPI_STORE, "src q"),
                           (160, 2048, 512, L.EM_EPI_RELU, "ffn w1"), (160, 5000, 512, L.EM_EPI_STORE_F32, "vocab"),
                           (640, 1536, 512, L.EM_EPI_STORE, "self qkv B=64"), (10, 1536, 512, L.EM_EPI_STORE, "qkv stream")]:
    x = torch.randn(M, K, device="cuda")
    g, b = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
    w = torch.randn(N, K, device="cuda").bfloat16()
    bias = torch.zeros(N, device="cuda")
    xn = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == L.EM_EPI_STORE_F32 else torch.bfloat16)
    st = L.current_stream_ptr()
    args = L.EmGemmArgs(A=xn.data_ptr(), W=w.data_ptr(), C=out.data_ptr(), bias=bias.data_ptr(), M=M, N=N, K=K,
                        lda=K, ldc=N, scale=1.0)

    def pair():
        lib.em_layernorm(L.EM_BF16, x.data_ptr(), g.data_ptr(), b.data_ptr(), M, K, 1e-12, xn.data_ptr(), None, st)
        lib.em_gemm(L.EM_BF16, epi, L.EM_A_PLAIN, args, st)

    def fused():
        lib.em_ln_gemm(L.EM_BF16, epi, x.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-12, w.data_ptr(), bias.data_ptr(),
                       out.data_ptr(), M, N, K, N, st)

    print(f"{name:16s} M={M:4d} N={N:5d} K={K}: LN+GEMM {t(pair):6.2f} us   ln_gemm {t(fused):6.2f} us")
