#!/usr/bin/env python3
"""Micro-benchmark of em_ffn_fused_bf16 (developer tool, GPU box)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from espnet_amd import lib as L

lib = L.load()
M, d, ff = 7968, 256, 1024
x = torch.randn(M, d, device="cuda")
g = torch.ones(d, device="cuda"); b = torch.zeros(d, device="cuda")
w1 = (torch.randn(ff, d, device="cuda") * d ** -0.5).bfloat16()
w2 = (torch.randn(d, ff, device="cuda") * ff ** -0.5).bfloat16()
b1 = torch.zeros(ff, device="cuda"); b2 = torch.zeros(d, device="cuda")
st = L.current_stream_ptr()


def call():
    return lib.em_ffn_fused_bf16(L.ptr(x), L.ptr(g), L.ptr(b), 1e-12, L.ptr(w1), L.ptr(b1), L.ptr(w2),
                                 L.ptr(b2), M, d, ff, 0.0, st)


for _ in range(5):
    L.check(call(), "ffn")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(50):
    call()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
print(f"ffn fused M={M} ff={ff}: {us:.2f} us  {4.0*M*d*ff/us/1e6:.1f} TF")
