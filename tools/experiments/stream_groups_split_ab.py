import json, sys, os
from pathlib import Path
sys.path.insert(0, os.getcwd())
import torch
import bench
torch.cuda.set_device(0)
for n, g in ((32, 2), (32, 3), (64, 2), (64, 3), (16, 4)):
    r = bench.run_stream_batch("bfloat16", n, 3, 1, groups=g)
    print(json.dumps({"split": os.environ.get("ESPNET_AMD_STREAM_FFN_SPLIT", "auto"), "helpers_off": os.environ.get("ESPNET_AMD_BLOCK_NO_HELPERS", "0"), "streams_per_tick": n, "groups": g, "audio_s_per_s": r["value"], "tick_ms": r["tick_latency_ms_median"]}), flush=True)
