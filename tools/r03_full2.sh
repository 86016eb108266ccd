#!/bin/bash
# Round-3 GPU call: the whole GPU suite, smoke, the default bench line (with roofline + traffic + cpu baseline), profile.
set -u
tag=${1:-r03n}; out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
echo "== pytest -m gpu"; (time timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -6) 2>&1 | tee "$out/pytest_gpu.txt"
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -4 | tee "$out/smoke.txt"
echo "== default bench"; (time timeout 900 python bench.py 2>"$out/bench_default.err" | tee "$out/bench_default.json" | cut -c1-400) 2>&1 | tail -5
echo "== rocprofv3 kernel stats, greedy"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof_greedy" -o bench --output-format csv -- \
   python "$OLDPWD/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 100 --warmup 10 >"$OLDPWD/$out/prof_greedy.log" 2>&1)
find "$out/prof_greedy" -name "*_kernel_trace.csv" -delete
f=$(find "$out/prof_greedy" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-160
