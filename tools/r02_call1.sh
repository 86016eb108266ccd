#!/bin/bash
# round 2, GPU call 1: parity of the fused block kernels, micro-benchmarks, fused vs per-operator bench lines
set -u
out=gpurun_out/r02a
mkdir -p "$out"
timeout 500 python -m pytest tests/test_gpu_block.py -q -x 2>&1 | tail -25 | tee "$out/pytest_block.txt"
timeout 300 python -m pytest tests/test_gpu_e2e.py -q -s -k "bfloat16 or fused or float32" 2>&1 | grep -v "^$" | tail -40 | tee "$out/pytest_e2e.txt"
timeout 200 python tools/block_bench.py 2>&1 | tail -12 | tee "$out/block_bench.txt"
timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>"$out/bench_fused.err" | tail -1 | tee "$out/bench_fused.json" | cut -c1-1500
ESPNET_AMD_FUSED=0 timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>"$out/bench_unfused.err" | tail -1 | tee "$out/bench_unfused.json" | cut -c1-600
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof" -o bench -- \
   python "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline >"$OLDPWD/$out/prof_bench.log" 2>&1)
ls -R "$out/prof" | head
tail -3 "$out/bench_fused.err"
