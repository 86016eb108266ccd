"""One line per run: the streaming workloads of bench.py (configs[4]: one stream through the hipGraph-captured step; 32
lock-step streams per tick) without the CPU baseline, for in-call A/B of the developer switches
(ESPNET_AMD_BLOCK_NO_HELPERS, ESPNET_AMD_STREAM_MHA_V1, ESPNET_AMD_STREAM_FUSED_MIN).  Usage: python tools/stream_ab.py [one|batch|groups|both|batch128]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "both"
torch.cuda.set_device(0)
out = {}
if what in ("one", "both"):
    r = bench.run_stream("bfloat16", 3, 1, cpu_base=False)
    out["one"] = {"audio_s_per_s": r["value"], "ms_per_call": r["config"]["call_latency_ms_median"]}
if what in ("batch", "both"):
    r = bench.run_stream_batch("bfloat16", 32, 3, 1)
    out["batch32"] = {"audio_s_per_s": r["value"], "tick_ms": r["tick_latency_ms_median"]}
if what in ("graph", "both"):
    r = bench.run_stream_batch("bfloat16", 32, 3, 1, graph=True)
    out["batch32_hipgraph"] = {"audio_s_per_s": r["value"], "tick_ms": r["tick_latency_ms_median"], "replays": r["hipgraph_replays"]}
if what in ("groups", "both"):
    for n in (32, 128):
        r = bench.run_stream_batch("bfloat16", n, 3, 1, groups=2)
        out[f"batch{n}_two_groups"] = {"audio_s_per_s": r["value"], "tick_ms": r["tick_latency_ms_median"]}
if what == "batch128":
    r = bench.run_stream_batch("bfloat16", 128, 3, 1)
    out["batch128"] = {"audio_s_per_s": r["value"], "tick_ms": r["tick_latency_ms_median"], "tick_ms_p95": r["tick_latency_ms_p95"],
                       "ms_per_utt": r["ms_per_step"]}
print(json.dumps(out), flush=True)
