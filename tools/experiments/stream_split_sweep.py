"""Split FFN of the streaming row-block launches (EmBlockArgs.ffn_split, round 6): the encoder step of n lock-step streams
(one block of 42 slots each, 12 x 256-wide layers, ff 2048) on the fused launches, shares per row block S = 1 / 2 / 4 / 8,
and the per-operator sequence beside it.  us per call, eager launches, median of 5 x 40 calls; |d|: one call on the
same rows against the unsplit fused launches.
`python tools/experiments/stream_split_sweep.py`"""
import ctypes as C
import os
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
from espnet_amd import lib as L  # noqa: E402
from espnet_amd.asr.encoder.contextual_block_conformer_encoder import ContextualBlockConformerEncoder  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    enc = ContextualBlockConformerEncoder(
        input_size=80, compute_dtype="bfloat16", output_size=256, attention_heads=4, linear_units=2048, num_blocks=12,
        input_layer="conv2d", normalize_before=True, activation_type="swish", macaron_style=True, use_cnn_module=True,
        cnn_module_kernel=15, block_size=40, hop_size=16, look_ahead=16, init_average=True, ctx_pos_enc=True).to(dev).eval()
    pk = enc._ensure_packed(dev)
    lib = L.load()
    Lb, d = 42, 256

    def time_calls(n):
        torch.manual_seed(n)  # (the same rows under every setting)
        x0 = torch.randn(n, 1, Lb, d, device=dev)
        x = x0.clone()
        past = torch.randn(n, 12, d, device=dev) * 0.1
        nxt = torch.empty_like(past)
        ws = enc._workspace(dev, n, Lb)
        st = L.current_stream_ptr()

        def call():
            L.check(lib.em_cb_encode_blocks_batch(enc.em_dtype, C.byref(pk["w"]), L.ptr(x), n, 1, Lb, 1, L.ptr(past), L.ptr(nxt),
                                                  L.ptr(ws), ws.numel(), st), "em_cb_encode_blocks_batch")

        for _ in range(5):
            x.copy_(x0)
            call()
        ts = []
        for _ in range(5):
            x.copy_(x0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                call()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 40 * 1e3)
        ts.sort()
        x.copy_(x0)  # ONE call on the same rows: what the layers leave behind (40 calls in place amplify any round-off)
        call()
        return ts[2], x.clone()

    if len(sys.argv) > 2 and sys.argv[1] == "--one":  # `--one n`: a few calls of n streams under the environment's switches (stamps, profiles)
        print(f"{int(sys.argv[2])} streams: {time_calls(int(sys.argv[2]))[0]:.1f} us per call", flush=True)
        return
    for n in (1, 2, 4, 8, 16, 32, 64):
        row = []
        ref = None
        for fused_min, split in ((10 ** 6, 1), (1, 1), (1, 2), (1, 4), (1, 8)):
            os.environ["ESPNET_AMD_STREAM_FUSED_MIN"] = str(fused_min)
            os.environ["ESPNET_AMD_STREAM_FFN_SPLIT"] = str(split)
            lib.em_dev_switches_reload()
            t, y = time_calls(n)
            if fused_min == 1 and split == 1:
                ref = y
            dmax = float((y - ref).abs().max()) if ref is not None and fused_min == 1 else float("nan")
            row.append(f"{'per-op' if fused_min > 1 else f'S={split}'} {t:7.1f} us" + (f" (|d| {dmax:.1e})" if split > 1 else ""))
        print(f"{n:3d} streams: " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
