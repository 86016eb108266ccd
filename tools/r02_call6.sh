#!/bin/bash
# direct-fragment-load block kernel: parity (block / e2e / full-size greedy), per-kernel times, quick bench
set -u
out=gpurun_out/r02j
mkdir -p "$out"
timeout 300 python -m pytest tests/test_gpu_block.py -m gpu -x -q 2>&1 | tail -5 | tee "$out/test_block.txt"
timeout 120 python tools/block_bench.py --iters 50 2>&1 | grep -E "block<|relpos" | tee "$out/block_bench.txt"
timeout 400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5 | tee "$out/test_e2e.txt"
timeout 200 python bench.py --quick --no-traffic --no-cpu-baseline --steps 500 --warmup 20 2>&1 | tail -2 | tee "$out/bench_quick.json"
