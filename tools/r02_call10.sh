#!/bin/bash
# direct-fragment block kernels + L2 warm-up: full GPU parity suite, quick bench, per-kernel times
set -u
out=gpurun_out/r02n
mkdir -p "$out"
(time timeout 1500 python -m pytest tests -m gpu -q -x) > "$out/pytest_gpu.txt" 2>&1
tail -3 "$out/pytest_gpu.txt"
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee "$out/smoke.txt"
timeout 300 python bench.py --quick --no-traffic --no-cpu-baseline --steps 1000 --warmup 30 2>/dev/null | tail -1 > "$out/bench_quick.json"; cut -c1-260 "$out/bench_quick.json"
timeout 120 python tools/block_bench.py --iters 50 2>&1 | grep -E "block<|relpos" | tee "$out/block_bench.txt"
