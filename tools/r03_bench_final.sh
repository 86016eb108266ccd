#!/bin/bash
# Round-3 GPU call: smoke, the default bench line (roofline + traffic + cpu baseline + beam + stream legs), rocprofv3 table.
set -u
tag=${1:-r03af}; out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -4 | tee "$out/smoke.txt"
echo "== default bench"; (time timeout 900 python bench.py 2>"$out/bench_default.err" < /dev/null | tee "$out/bench_default.json" | cut -c1-300) 2>&1 | tail -5
echo "== rocprofv3 kernel stats, greedy"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof_greedy" -o bench --output-format csv -- \
   python "$OLDPWD/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 100 --warmup 10 >"$OLDPWD/$out/prof_greedy.log" 2>&1 < /dev/null)
find "$out/prof_greedy" -name "*_kernel_trace.csv" -delete 2>/dev/null
f=$(find "$out/prof_greedy" -name "*kernel_stats.csv" 2>/dev/null | head -1); if [ -n "$f" ]; then head -14 "$f" | cut -c1-160; else echo "no stats"; fi
