#!/bin/bash
set -u
tag=${1:-r03p}; out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
echo "== streaming tests"; timeout 600 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_online_search.py -q -x 2>&1 | tail -4 | tee "$out/pytest_streaming.txt"
st() { local name=$1; shift; env "$@" timeout 300 python bench.py --workload stream --steps 3 --warmup 1 --no-cpu-baseline 2>"$out/bench_stream_$name.err" | tee "$out/bench_stream_$name.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['config']['call_latency_ms_median'], d['config']['call_latency_ms_p95'])" || tail -3 "$out/bench_stream_$name.err"; }
st fused X=1
st pair ESPNET_AMD_STREAM_NO_LN_GEMM=1
