#!/usr/bin/env python3
"""Microbenchmark of em_ffn_rows_fused (csrc/ffn_rows.hip) at the large model's shape (M = 64 x 249 rows, d = 512,
ff = 2048): microseconds per launch and TFLOP/s for both LayerNorm modes, and the error against torch (f32 matmuls
on the same bf16-rounded operands).  ESPNET_AMD_LIB=espnet_amd/lib/dbg/lib_ffn<d>.so times a developer build
(tools/build_block_variants.sh ffn<d>; wrong results by design).   usage: python tools/ffn_rows_bench.py [M] [ff]"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from espnet_amd import lib as L  # noqa: E402
from espnet_amd.asr.encoder.conformer_encoder import pack_ffn_rows_w1, pack_ffn_rows_w2  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 15936
ff = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
d = 512
lib = L.load()
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()  # noqa: E731
x = r(M, d) * 2 + 0.3
xn = F.layer_norm(x, (d,)).to(torch.bfloat16)
w1, w2 = r(ff, d, sc=d ** -0.5).to(torch.bfloat16), r(d, ff, sc=ff ** -0.5).to(torch.bfloat16)
b1, b2 = r(ff, sc=0.1), r(d, sc=0.1)
g1, be1, g2, be2 = 1 + r(d, sc=0.1), r(d, sc=0.1), 1 + r(d, sc=0.1), r(d, sc=0.1)
w1p, w2p = pack_ffn_rows_w1(w1.cpu()).cuda(), pack_ffn_rows_w2(w2.cpu()).cuda()
h = F.silu(xn.float() @ w1.float().t() + b1).to(torch.bfloat16).float()
x1 = x + 0.5 * (h @ w2.float().t() + b2)
for mode in (1, 2):
    xd, out, of = x.clone(), torch.zeros(M, d, dtype=torch.bfloat16, device="cuda"), torch.zeros(M, d, device="cuda")
    a = L.EmFfnRowsArgs(xn_in=xn.data_ptr(), x=xd.data_ptr(), w1p=w1p.data_ptr(), w2p=w2p.data_ptr(), b1=b1.data_ptr(),
                        b2=b2.data_ptr(), g1=g1.data_ptr(), be1=be1.data_ptr(), g2=g2.data_ptr(), be2=be2.data_ptr(),
                        xn_out=out.data_ptr(), out_f32=of.data_ptr() if mode == 2 else 0, M=M, d=d, ff=ff, ln_mode=mode,
                        scale=0.5, eps=1e-12)
    L.check(lib.em_ffn_rows_fused(a, L.current_stream_ptr()))
    torch.cuda.synchronize()
    if mode == 1:
        rx, rn = x1, F.layer_norm(x1, (d,), g1, be1, 1e-12)
    else:
        rx = F.layer_norm(x1, (d,), g1, be1, 1e-12)
        rn = F.layer_norm(rx, (d,), g2, be2, 1e-12)
    ex, en = (xd - rx).abs().max().item(), (out.float() - rn).abs().max().item()
    for _ in range(5):
        lib.em_ffn_rows_fused(a, L.current_stream_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        lib.em_ffn_rows_fused(a, L.current_stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"ffn_rows mode {mode}: M {M} ff {ff}: {us:7.1f} us per launch, {4.0 * M * d * ff / us * 1e-6:7.1f} TFLOP/s, "
          f"max err x {ex:.2e} LN {en:.2e}", flush=True)
