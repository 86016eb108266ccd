#!/usr/bin/env python3
"""Probe (round 4): does the greedy step gain from running the batch as TWO half batches on two HIP streams?  At B = 32
every fused kernel is exactly one workgroup per CU and all of them walk through their memory and compute phases in lock
step; two half-sized launch sequences on disjoint CUs drift apart and could smooth the bursts.
Prints ms per step for one stream x B and two streams x B/2 (same total work, same collation-free loop)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
from espnet_amd.tasks.asr import ASRTask  # noqa: E402


def main():
    B, steps = 32, 400
    torch.manual_seed(0)
    model = ASRTask.build_model(bench.model_config("small", "bfloat16")).cuda().eval()
    wav = bench.synth_batch(0, B).cuda()
    lens = [bench.N_SAMPLES] * B
    halves = [wav[: B // 2].contiguous(), wav[B // 2:].contiguous()]
    hl = lens[: B // 2]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def one():
        model.greedy_ctc_device(model.encode_device(wav, lens))

    def two():
        for s, w in zip(streams, halves):
            with torch.cuda.stream(s):
                model.greedy_ctc_device(model.encode_device(w, hl))

    # round 6: two FULL batches in flight on two streams (each stream has its own workspace in the encoder): what one stream
    # leaves idle at a kernel boundary (1.5 - 1.9 us between dependent launches, ~27 per step) the other stream's launch can
    # fill.  Counts per batch: one call = two batches.
    def two_full():
        for s in streams:
            with torch.cuda.stream(s):
                model.greedy_ctc_device(model.encode_device(wav, lens))

    with torch.no_grad():
        for name, fn, per in (("one stream x 32", one, 1), ("two streams x 16", two, 1), ("two streams x 32 (two batches in flight)", two_full, 2),
                              ("one stream x 32", one, 1), ("two streams x 32 (two batches in flight)", two_full, 2)):
            for _ in range(30):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            print(f"{name}: {(time.perf_counter() - t0) / steps / per * 1e3:.3f} ms per batch of {B} utterances", flush=True)


if __name__ == "__main__":
    main()
