#!/bin/bash
# Round-3 GPU call: the whole GPU suite, the greedy bench (quick) with its rocprofv3 table, the beam leg.
#   bash tools/r03_full.sh <tag>
set -u
tag=${1:-r03f}; out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
echo "== pytest -m gpu"; (time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6) 2>&1 | tee "$out/pytest_gpu.txt"
echo "== greedy bench (quick)"; timeout 300 python bench.py --quick --no-traffic --no-cpu-baseline --steps 600 --warmup 30 2>"$out/bench_quick.err" | tee "$out/bench_quick.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['families'])"
echo "== rocprofv3 kernel stats, greedy"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof_greedy" -o bench --output-format csv -- \
   python "$OLDPWD/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 100 --warmup 10 >"$OLDPWD/$out/prof_greedy.log" 2>&1)
find "$out/prof_greedy" -name "*_kernel_trace.csv" -delete
f=$(find "$out/prof_greedy" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-180
echo "== beam bench"; timeout 300 python bench.py --workload beam --steps 3 --warmup 1 --no-cpu-baseline 2>"$out/bench_beam.err" | tee "$out/bench_beam.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['search']['ms_per_search_step'])"
