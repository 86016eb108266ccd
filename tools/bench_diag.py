"""Where does a bench step's wall time go?  Variants of the greedy step loop (developer diagnostic)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
from espnet_amd.tasks.asr import ASRTask

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = ASRTask.build_model(bench.model_config("small", "bfloat16")).to(dev).eval()
B = 32
wav_host = bench.synth_batch(0, B)
wav = wav_host.to(dev)
lens = [bench.N_SAMPLES] * B
T = model.encoder.output_frames(1 + bench.N_SAMPLES // 160)

def plain():
    return model.greedy_ctc_device(model.encode_device(wav, lens))

def loop(name, fn, n=100, fin=None):
    with torch.no_grad():
        for _ in range(10): fn()
        if fin: fin()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        if fin: fin()
        torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter()-t0)/n*1e3:.3f} ms/step", flush=True)

loop("plain (no collation)", plain)
loop("plain again", plain)
sink = bench.HypothesisSink(0, 1, B, T, dev)
def with_sink():
    _, tok, tl = plain(); sink.push(tok, tl)
loop("sink.push", with_sink, fin=sink.drain)
def sync_each():
    _, tok, tl = plain(); tok.cpu(); tl.cpu()
sink2 = bench.HypothesisSink(0, 1, B, T, dev)
sink2._deliver = lambda slot: sink2.events[slot].synchronize()
def with_sink_nounpack():
    _, tok, tl = plain(); sink2.push(tok, tl)
loop("sink.push without unpack", with_sink_nounpack, fin=sink2.drain)
import numpy as np
t0 = time.perf_counter()
for _ in range(100): sink.D.unpack_records(*sink.pinned[0], B, 1, as_arrays=True)
print(f"unpack_records alone: {(time.perf_counter()-t0)/100*1e3:.3f} ms", flush=True)
t0 = time.perf_counter()
for _ in range(100): sink.pinned[0][0][torch.arange(B)]
print(f"torch index on pinned: {(time.perf_counter()-t0)/100*1e3:.3f} ms", flush=True)
loop("blocking .cpu() each step", sync_each)
pin = torch.empty(B, T, dtype=torch.int32).pin_memory()
ev = torch.cuda.Event()
def async_copy_only():
    _, tok, tl = plain(); pin.copy_(tok, non_blocking=True); ev.record()
loop("async copy, no wait", async_copy_only)
def async_copy_wait_prev():
    ev.synchronize()
    _, tok, tl = plain(); pin.copy_(tok, non_blocking=True); ev.record()
loop("wait prev event, then launch", async_copy_wait_prev)
fd = bench.HostFeeder(wav_host, dev)
def with_feeder():
    r = model.greedy_ctc_device(model.encode_device(fd.acquire(), lens)); fd.release(); return r
loop("feeder only", with_feeder)
loop("plain final", plain)
