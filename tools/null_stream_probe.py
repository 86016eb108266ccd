#!/usr/bin/env python3
"""Does a process that has created many HIP streams launch more slowly?  (round 6: the bench's late legs ran slower inside
the default line than alone - beam_cfg3_per_gpu 0.58 against 0.515 ms per label step - while nothing but the process' history
differed.)  Times back-to-back tiny launches on the default (null) stream and on one side stream, before and after the process
has touched all 32 streams of torch's pool, and again with events recorded on them.  Measured (profiles/r06ak_null_stream_probe.txt):
4.57 -> 4.71 us per launch on the null stream, 4.65 -> 4.74 on a side stream: NOT the cause.

    python tools/null_stream_probe.py"""
import time

import torch


def rate(stream, n=4000):
    x = torch.zeros(64, device="cuda")
    with torch.cuda.stream(stream):
        for _ in range(200):
            x.add_(1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            x.add_(1.0)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
    return t_issue / n * 1e6, t_all / n * 1e6


def main():
    dev = torch.device("cuda:0")
    null = torch.cuda.default_stream(dev)
    side = torch.cuda.Stream(dev)
    print("fresh process         null: issue %.2f us, done %.2f us per launch | side: issue %.2f, done %.2f" % (*rate(null), *rate(side)))
    pool = [torch.cuda.Stream(dev) for _ in range(40)]
    for s in pool:  # touch them (a stream of the pool exists once it was used)
        with torch.cuda.stream(s):
            torch.zeros(8, device="cuda").add_(1.0)
    torch.cuda.synchronize()
    print("after 40 Stream()s    null: issue %.2f us, done %.2f us per launch | side: issue %.2f, done %.2f" % (*rate(null), *rate(side)))
    ev = [torch.cuda.Event() for _ in pool]
    for s, e in zip(pool, ev):
        with torch.cuda.stream(s):
            e.record()
    print("... + events recorded null: issue %.2f us, done %.2f us per launch | side: issue %.2f, done %.2f" % (*rate(null), *rate(side)))


if __name__ == "__main__":
    main()
