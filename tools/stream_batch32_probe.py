#!/usr/bin/env python3
"""The 32-stream tick of the streaming encoder (bench.py's `stream.batch32` leg) on its own, for
`rocprofv3 --kernel-trace --stats -- python tools/stream_batch32_probe.py` (profiles/r04zs_stream_batch32_kernel_stats.csv)."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

r = bench.run_stream_batch("bfloat16", 32, 3, 1)
print("batch32:", json.dumps({k: r[k] for k in r if k != "config"})[:300])
