#!/bin/bash
set -u
tag=${1:-r03q}; out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
echo "== streaming tests"; timeout 600 python -m pytest tests/test_gpu_streaming.py -q -x 2>&1 | tail -6 | tee "$out/pytest_streaming.txt"
timeout 400 python bench.py --workload stream --steps 3 --warmup 1 --no-cpu-baseline 2>"$out/bench_stream.err" | tee "$out/bench_stream.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['call_latency_ms_median']); print(d['batch32'])" || tail -5 "$out/bench_stream.err"
