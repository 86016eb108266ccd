#!/bin/bash
# Round-3 GPU call S: the fused subsampling kernel (persistent tiles, operands built between the stores).
#   bash tools/r03_call_s.sh <tag>     stamps of one workgroup, the conv tests, per-kernel stats, the quick bench line
set -u
tag=${1:-r03s}; out=$PWD/gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
R=$PWD
echo "== stamps"
EM_SUB2_STAMPS=1 timeout 120 python bench.py --quick --no-traffic --no-roofline --no-cpu-baseline --steps 3 --warmup 2 2>&1 | grep "sub2 stamps" | head -2 | tee "$out/stamps.txt"
echo "== tests"
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "sub12 or conv2d or subsampl" 2>&1 | tail -3
echo "== kernel stats"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$out/stats" -o s --output-format csv -- python "$R/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 50 --warmup 5 > "$out/stats.log" 2>&1 < /dev/null)
find "$out/stats" -name "*_kernel_trace.csv" -delete 2>/dev/null
f=$(find "$out/stats" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then head -16 "$f" | cut -c1-150; else echo "no stats file"; tail -5 "$out/stats.log"; fi
echo "== bench"
timeout 200 python bench.py --quick --no-traffic --no-cpu-baseline --steps 600 --warmup 30 2>/dev/null < /dev/null | tee "$out/bench_quick.json" | cut -c1-260
