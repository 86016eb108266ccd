#!/bin/bash
# XCC-aware L2 warm-up: dispatch map of this box, block parity, quick bench
set -u
out=gpurun_out/r02t
mkdir -p "$out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/xcc_map tools/experiments/xcc_map.hip 2>/dev/null; /tmp/xcc_map | tee "$out/xcc_map.txt"
timeout 300 python -m pytest tests/test_gpu_block.py -m gpu -x -q 2>&1 | tail -2 | tee "$out/test_block.txt"
for i in 1 2; do
timeout 200 python bench.py --quick --no-traffic --no-cpu-baseline --no-roofline --steps 300 --warmup 20 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a "$out/plain.txt"
done
