#!/bin/bash
# Round-3 GPU call W: per-kernel stats of the beam-search label step.
set -u
tag=${1:-r03w}; out=$PWD/gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$out/search_stats" -o s --output-format csv -- python "$R/bench.py" --workload beam --steps 1 --warmup 1 --no-cpu-baseline --no-traffic > "$out/search_stats.log" 2>&1 < /dev/null)
find "$out/search_stats" -name "*_kernel_trace.csv" -delete 2>/dev/null
f=$(find "$out/search_stats" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then head -16 "$f" | cut -c1-190; else echo "no stats"; tail -5 "$out/search_stats.log"; fi
