#!/bin/bash
# Round-3 GPU call for the beam-search label step: kernel / search / scorer tests, then the configs[2] bench leg
# (and optionally its rocprofv3 per-kernel table).   bash tools/r03_search.sh <tag> [--profile]
set -u
tag=${1:-r03s}; out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_search.py tests/test_gpu_scorer_interface.py tests/test_gpu_online_search.py -q -x 2>&1 | tail -4 | tee "$out/pytest_search.txt"
echo "== beam bench"; timeout 300 python bench.py --workload beam --steps 3 --warmup 1 --no-cpu-baseline 2>"$out/bench_beam.err" | tee "$out/bench_beam.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['search'])"
if [ "${2:-}" = "--profile" ]; then
  echo "== beam kernel stats"
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/search_stats" -o s --output-format csv -- python "$OLDPWD/bench.py" --workload beam --steps 1 --warmup 1 --no-cpu-baseline > "$OLDPWD/$out/search_stats.log" 2>&1)
  find "$out/search_stats" -name "*_kernel_trace.csv" -delete
  f=$(find "$out/search_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -22 "$f" | cut -c1-200
fi
