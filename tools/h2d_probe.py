"""Where does the host -> device copy of the waveforms leak into the greedy step?  

The bench's `pcie_inclusive` leg double-buffers the 20.5 MB of a batch on a copy stream under the previous step's compute and
still cost 0.18 ms per 1.03 ms step.  This probe times, on one box in one call:
  a. the copy alone (pinned host -> device, `copy_(non_blocking=True)`): GB/s, as one piece and as 2 / 4 pieces on as many streams;
  b. the step with resident inputs;
  c. the step under the bench's HostFeeder;
  d. the same with the copy split over 2 / 4 streams, with a high-priority copy stream, and with the copy issued ONE STEP
     EARLIER (two steps of slack instead of one: triple buffering);
  e. the step with a copy running that nobody waits for (pure interference: does the copy slow the kernels?).
Prints one line per variant.  `python tools/h2d_probe.py [--steps 300]`
"""
import argparse
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import bench  # noqa: E402


class Feeder:
    """Waveform batches from pinned host memory: `depth` device buffers, the copy of batch k + depth - 1 issued when batch k
    is released, split over `nsplit` copy streams."""

    def __init__(self, wav_host, dev, depth=2, nsplit=1, priority=0):
        self.host = wav_host.pin_memory()
        self.depth, self.nsplit = depth, nsplit
        self.bufs = [torch.empty_like(wav_host, device=dev) for _ in range(depth)]
        self.streams = [torch.cuda.Stream(device=dev, priority=priority) for _ in range(nsplit)]
        self.copied = [[torch.cuda.Event() for _ in range(nsplit)] for _ in range(depth)]
        self.consumed = [None] * depth
        self.k = 0
        rows = wav_host.shape[0]
        self.cuts = [(i * rows // nsplit, (i + 1) * rows // nsplit) for i in range(nsplit)]
        for s in range(depth - 1):
            self._prefetch(s)

    def _prefetch(self, slot):
        for i, st in enumerate(self.streams):
            with torch.cuda.stream(st):
                if self.consumed[slot] is not None:
                    st.wait_event(self.consumed[slot])
                a, b = self.cuts[i]
                self.bufs[slot][a:b].copy_(self.host[a:b], non_blocking=True)
                self.copied[slot][i].record(st)

    def acquire(self):
        slot = self.k % self.depth
        for ev in self.copied[slot]:
            torch.cuda.current_stream().wait_event(ev)
        return self.bufs[slot]

    def release(self):
        slot = self.k % self.depth
        ev = torch.cuda.Event()
        ev.record()
        self.consumed[slot] = ev
        self.k += 1
        self._prefetch((self.k + self.depth - 2) % self.depth)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    from espnet_amd.tasks.asr import ASRTask
    from espnet_amd import distributed as D

    torch.manual_seed(0)
    model = ASRTask.build_model(bench.model_config("small", "bfloat16")).to(dev).eval()
    B = 32
    wav_host = bench.synth_batch(0, B)
    wav = wav_host.to(dev)
    lens = [bench.N_SAMPLES] * B
    T = model.encoder.output_frames(1 + bench.N_SAMPLES // 160)

    def barrier():
        torch.cuda.synchronize()

    # ---- a. the copy alone
    pinned = wav_host.pin_memory()
    mb = pinned.numel() * 4 / 1e6
    for nsplit in (1, 2, 4):
        streams = [torch.cuda.Stream(device=dev) for _ in range(nsplit)]
        dst = torch.empty_like(wav)
        rows = B
        cuts = [(i * rows // nsplit, (i + 1) * rows // nsplit) for i in range(nsplit)]

        def go():
            for (x, y), st in zip(cuts, streams):
                with torch.cuda.stream(st):
                    dst[x:y].copy_(pinned[x:y], non_blocking=True)

        for _ in range(3):
            go()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            go()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / n
        print(f"copy alone, {nsplit} stream(s): {t * 1e3:.3f} ms for {mb:.1f} MB = {mb / t / 1e3:.1f} GB/s", flush=True)

    def run(name, make_feeder, interfere=False):
        sink = D.RecordRing(0, 1, B, T, bench.RING_STEPS, dev)
        fd = make_feeder() if make_feeder else None
        side = torch.cuda.Stream(device=dev) if interfere else None
        scratch = torch.empty_like(wav) if interfere else None

        def step():
            tok_v, len_v, _ = sink.slot()
            if interfere:
                with torch.cuda.stream(side):
                    scratch.copy_(pinned, non_blocking=True)
            src = fd.acquire() if fd else wav
            st = model.encode_device(src, lens)
            model.greedy_ctc_device(st, out=(tok_v, len_v))
            if fd:
                fd.release()
            sink.commit()

        t = bench.timed_loop(step, a.steps, 20, barrier, sink.drain)
        print(f"{name}: {t / a.steps * 1e3:.4f} ms/step", flush=True)

    for rep in range(2):
        run("resident", None)
        run("2 buffers, 1 stream (the round-5 HostFeeder)", lambda: Feeder(wav_host, dev, 2, 1))
        run("2 buffers, 2 streams", lambda: Feeder(wav_host, dev, 2, 2))
        run("2 buffers, 4 streams", lambda: Feeder(wav_host, dev, 2, 4))
        run("2 buffers, 1 high-priority stream", lambda: Feeder(wav_host, dev, 2, 1, priority=-1))
        run("3 buffers, 1 stream (two steps of slack)", lambda: Feeder(wav_host, dev, 3, 1))
        run("3 buffers, 2 streams", lambda: Feeder(wav_host, dev, 3, 2))
        run("resident + a copy nobody waits for", None, interfere=True)
        run("bench HostFeeder with the probed copy stream (bench.pick_copy_stream)", lambda: bench.HostFeeder(wav_host, dev))


if __name__ == "__main__":
    main()
