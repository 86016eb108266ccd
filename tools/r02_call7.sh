#!/bin/bash
# direct-fragment-load block kernel, quick loop: block parity + per-kernel times + stage stamps + quick bench
set -u
out=gpurun_out/r02k
mkdir -p "$out"
timeout 300 python -m pytest tests/test_gpu_block.py -m gpu -x -q 2>&1 | tail -3 | tee "$out/test_block.txt"
timeout 120 python tools/block_bench.py --iters 50 2>&1 | grep -E "block<" | tee "$out/block_bench.txt"
EM_BLOCK_STAMPS=1 timeout 120 python tools/block_bench.py --iters 2 2>&1 | grep -E "stamps" | tail -4 | tee "$out/stamps.txt"
if [ "${1:-}" = bench ]; then
  timeout 200 python bench.py --quick --no-traffic --no-cpu-baseline --steps 300 --warmup 20 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'families', d['roofline']['families'])" | tee "$out/bench_quick.txt"
fi
