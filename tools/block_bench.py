#!/usr/bin/env python3
"""Micro-benchmark of the fused per-block kernels (csrc/block.hip) and the LDS-resident attention
(csrc/attention2.hip) at the bench shape (B = 32 utterances x T = 249 frames, d = 256, ff = 1024), next to the
kernels they replace.  Prints microseconds per launch (HIP events over a loop on the current stream) and, for
the block kernels, the per-CU weight ingest they sustain (every workgroup streams all weights of its stages).

    python tools/block_bench.py [--B 32] [--T 249] [--iters 100]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from espnet_amd import lib as L  # noqa: E402
from tests.test_gpu_block import BF, D, G, H, Layer, block_args, group, rnd, tpad  # noqa: E402


_COLD = None  # --cold: a buffer streamed through the L2s between the timed launches


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    if _COLD is not None:
        # in the encoder a block kernel finds its weights in no L2 (36 MB of weights pass through 8 x 4 MB); back to
        # back in this tool they stay resident.  Cold mode evicts them before every launch and times the launches
        # one by one (events on torch's stream = the launch stream).
        tot = 0.0
        for _ in range(iters):
            _COLD.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot * 1e3 / iters
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=249)
    ap.add_argument("--ff", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--cold-mb", type=int, default=48, help="size of the eviction buffer: 48 MB clears the L2s and leaves the "
                    "weights in the 256 MB Infinity Cache, 384 MB clears that too (the encoder streams > 1 GB per step)")
    ap.add_argument("--cold", action="store_true", help="evict the L2s before every timed launch, one launch per measurement (NOT a proxy for the "
                                                         "encoder's launch sequence: profiles/r02y vs r02m)")
    args = ap.parse_args()
    if args.cold:
        global _COLD
        _COLD = torch.zeros(args.cold_mb * 256 * 1024, device="cuda")  # read + written before every launch
    B, T, ff = args.B, args.T, args.ff
    lib = L.load()
    sp = L.current_stream_ptr
    l0, l1 = Layer(400, ff), Layer(500, ff)
    M, Tp = B * T, tpad(T)
    c = lambda t: t.contiguous().cuda()
    x = c(rnd(M, D, seed=1))
    glu = c(rnd(M, D, seed=2).to(BF))
    ctx = c(rnd(M, D, seed=3).to(BF))
    qh, kh = (torch.zeros(B, H, Tp, 64, dtype=BF, device="cuda") for _ in range(2))
    vt = torch.zeros(B, H, 64, Tp, dtype=BF, device="cuda")
    out = torch.zeros(M, D, device="cuda")
    act = torch.zeros(M, D, dtype=BF, device="cuda")
    from espnet_amd.asr.encoder.conformer_encoder import pack_k_units, pack_w1, pack_w2

    w = {k: c((pack_w1 if k.endswith("w1") else pack_k_units)(getattr(l0, k)).to(BF)) for k in ("pw2", "ff_w1", "wout", "ffm_w1", "wqkv")}
    w["ff_w2"], w["ffm_w2"] = c(pack_w2(l0.ff_w2).to(BF)), c(pack_w2(l0.ffm_w2).to(BF))
    w["pw1f"] = c(pack_k_units(l0.pw1[l0.perm()]).to(BF))
    w["dw_w"], w["dw_b"] = c(l0.dw_w), c(l0.dw_b)
    keep = []

    def P(t):
        t = c(t)
        keep.append(t)
        return t

    a_da = block_args(B, T, ff, x=x, glu=glu, qh=qh, kh=kh, vt=vt, pw2=w["pw2"], ff_w1=w["ff_w1"], ff_w2=w["ff_w2"],
                      dw_w=w["dw_w"], dw_b=w["dw_b"], ffm_w1=w["ffm_w1"], ffm_w2=w["ffm_w2"], wqkv=w["wqkv"],
                      params=P(torch.cat(l0.d_groups() + l1.a_groups())))
    a_a = block_args(B, T, ff, x=x, qh=qh, kh=kh, vt=vt, ffm_w1=w["ffm_w1"], ffm_w2=w["ffm_w2"], wqkv=w["wqkv"],
                     params=P(torch.cat(l1.a_groups())))
    a_c = block_args(B, T, ff, x=x, ctx=ctx, glu=glu, wout=w["wout"], pw1f=w["pw1f"], params=P(l0.c_group()))
    a_df = block_args(B, T, ff, x=x, glu=glu, enc_out=out, enc_act=act, pw2=w["pw2"], ff_w1=w["ff_w1"],
                      ff_w2=w["ff_w2"], dw_w=w["dw_w"], dw_b=w["dw_b"],
                      params=P(torch.cat(l0.d_groups() + [group(torch.ones(D), torch.zeros(D)), torch.zeros(G)])))
    kib = {"C": 384, "A": 2 * ff * 512 / 1024 + 384, "D|A": 128 + 4 * ff * 512 / 1024 + 384,
           "D|FINAL": 128 + 2 * ff * 512 / 1024}
    for name, mode, a in (("C", L.EM_BLOCK_C, a_c), ("A", L.EM_BLOCK_A, a_a), ("D|A", L.EM_BLOCK_D | L.EM_BLOCK_A, a_da),
                          ("D|FINAL", L.EM_BLOCK_D | L.EM_BLOCK_FINAL, a_df)):
        rc = lib.em_conformer_block_fused(mode, a, sp())
        L.check(rc, name)
        us = timeit(lambda: lib.em_conformer_block_fused(mode, a, sp()), args.iters)
        print(f"block<{name:8s}> {us:8.2f} us   weights/workgroup {kib[name]:7.0f} KiB -> "
              f"{kib[name] * 1024 / us / 1e3:6.1f} GB/s per CU", flush=True)
        x.copy_(rnd(M, D, seed=1).cuda())  # keep the residual stream bounded across iterations

    # attention: new per-head kernel vs the tiled one
    p = c(rnd(2 * T - 1, 12 * D, seed=7).to(BF))
    u, v = c(rnd(H, 64, seed=8, scale=0.3)), c(rnd(H, 64, seed=9, scale=0.3))
    kl = torch.full((B,), T, dtype=torch.int32, device="cuda")
    qh.copy_(torch.randn_like(qh.float()).to(BF))
    kh.copy_(torch.randn_like(kh.float()).to(BF))
    vt.copy_(torch.randn_like(vt.float()).to(BF))
    us2 = timeit(lambda: lib.em_relpos_attention2_bf16(L.ptr(qh), L.ptr(kh), L.ptr(vt), L.ptr(p), 12 * D, L.ptr(u),
                                                       L.ptr(v), L.ptr(kl), B, T, Tp, H, L.ptr(ctx), sp()), args.iters)
    qkv = c(rnd(M, 3 * D, seed=10).to(BF))
    us1 = timeit(lambda: lib.em_relpos_attention(L.EM_BF16, L.ptr(qkv), L.ptr(p), 12 * D, L.ptr(u), L.ptr(v),
                                                 L.ptr(kl), B, T, H, 64, L.ptr(ctx), sp()), args.iters)
    fl = 6.0 * B * H * T * T * 64
    print(f"relpos_attn2    {us2:8.2f} us  ({fl / us2 / 1e6:6.1f} TFLOP/s)   relpos_attn (tiled) {us1:8.2f} us", flush=True)


if __name__ == "__main__":
    main()
