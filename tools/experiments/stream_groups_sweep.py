"""Groups of lock-step streams with a tick in flight each (Speech2TextStreaming.batch_call_async): audio-s/s and tick latency
over streams per tick x groups.  `python tools/experiments/stream_groups_sweep.py`"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402

torch.cuda.set_device(0)
for n, gs in ((8, (1, 2, 4, 8)), (16, (1, 2, 4)), (32, (1, 2, 3, 4)), (64, (1, 2, 3)), (128, (1, 2, 3))):
    for g in gs:
        r = bench.run_stream_batch("bfloat16", n, 3, 1, groups=g)
        print(json.dumps({"streams_per_tick": n, "groups": g, "audio_s_per_s": r["value"], "tick_ms": r["tick_latency_ms_median"],
                          "tick_ms_p95": r["tick_latency_ms_p95"]}), flush=True)
