#!/usr/bin/env python3
"""Micro-benchmark of em_gemm on the encoder's GEMM shapes (developer tool; run on the GPU box).
Prints us/launch and TFLOP/s, next to torch.matmul (hipBLASLt) as a yardstick for what the chip
reaches at the same shape -- the yardstick is not part of the product path."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from espnet_amd import lib as L

lib = L.load()
dev = "cuda"
SHAPES = [  # (name, M, N, K, epilogue)
    ("ffn_w1 small", 7968, 1024, 256, L.EM_EPI_SWISH),
    ("ffn_w2 small", 7968, 256, 1024, L.EM_EPI_RESID_F32),
    ("qkv small", 7968, 768, 256, L.EM_EPI_STORE),
    ("out small", 7968, 256, 256, L.EM_EPI_RESID_F32),
    ("glu small", 7968, 512, 256, L.EM_EPI_GLU),
    ("ctc small", 7968, 5000, 256, L.EM_EPI_STORE_F32),
    ("ffn_w1 large", 3984, 2048, 512, L.EM_EPI_SWISH),
    ("ffn_w2 large", 3984, 512, 2048, L.EM_EPI_RESID_F32),
    ("square 4096", 4096, 4096, 4096, L.EM_EPI_STORE),
    ("embed small", 7968, 256, 4864, L.EM_EPI_SCALE_F32),
    ("embed large", 3984, 512, 9728, L.EM_EPI_SCALE_F32),
    ("qkv large", 3984, 1536, 512, L.EM_EPI_STORE),
    ("out large", 3984, 512, 512, L.EM_EPI_RESID_F32),
    # E-Branchformer (17 x 512d, ff 2048, cgMLP 3072), B = 32: 7 968 rows
    ("ebf ffn_w1", 7968, 2048, 512, L.EM_EPI_SWISH),
    ("ebf ffn_w2", 7968, 512, 2048, L.EM_EPI_RESID_F32),
    ("ebf cgmlp_p1", 7968, 3072, 512, L.EM_EPI_GELU),
    ("ebf cgmlp_p2", 7968, 512, 1536, L.EM_EPI_STORE),
    ("ebf qkv", 7968, 1536, 512, L.EM_EPI_STORE),
    ("ebf merge", 7968, 512, 1024, L.EM_EPI_RESID_F32),
    # Conformer-large, B = 64: 15 936 rows
    ("large64 ffn_w1", 15936, 2048, 512, L.EM_EPI_SWISH),
    ("large64 qkv", 15936, 1536, 512, L.EM_EPI_STORE),
]


def run(name, M, N, K, epi, iters=50):
    a = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    nout = N // 2 if epi == L.EM_EPI_GLU else N
    f32out = epi in (L.EM_EPI_RESID_F32, L.EM_EPI_SCALE_F32, L.EM_EPI_STORE_F32)
    c = torch.zeros(M, nout, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
    bias = torch.randn(N, device=dev)
    args = L.EmGemmArgs(A=a.data_ptr(), W=w.data_ptr(), C=c.data_ptr(), bias=bias.data_ptr(), M=M,
                        N=N, K=K, lda=K, ldc=nout, scale=0.5)
    st = L.current_stream_ptr()
    for _ in range(5):
        L.check(lib.em_gemm(L.EM_BF16, epi, L.EM_A_PLAIN, args, st), name)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        lib.em_gemm(L.EM_BF16, epi, L.EM_A_PLAIN, args, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    wt = w.t().contiguous()
    for _ in range(5):
        torch.matmul(a, wt)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        torch.matmul(a, wt)
    e1.record()
    torch.cuda.synchronize()
    us_ref = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * M * N * K
    print(f"{name:16s} M={M:5d} N={N:5d} K={K:5d}  em_gemm {us:8.2f} us {fl/us/1e6:8.1f} TF | "
          f"torch.matmul {us_ref:8.2f} us {fl/us_ref/1e6:8.1f} TF")


if __name__ == "__main__":
    for s in SHAPES:
        run(*s)
