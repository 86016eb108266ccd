#!/bin/bash
set -u
tag=${1:-r03l}; out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
echo "== attention / block tests"; timeout 600 python -m pytest tests/test_gpu_block.py -q -x 2>&1 | tail -4 | tee "$out/pytest_block.txt"
echo "== e2e"; timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x 2>&1 | tail -4 | tee "$out/pytest_e2e.txt"
timeout 300 python bench.py --quick --no-traffic --no-cpu-baseline --steps 600 --warmup 30 2>"$out/bench.err" | tee "$out/bench.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['families'])" || tail -5 "$out/bench.err"
