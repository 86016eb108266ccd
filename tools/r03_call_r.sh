#!/bin/bash
set -u
tag=${1:-r03r}; out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_search.py tests/test_gpu_scorer_interface.py tests/test_gpu_online_search.py -q -x 2>&1 | tail -4 | tee "$out/pytest.txt"
beam() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py --workload beam --steps 3 --warmup 1 --no-cpu-baseline --no-traffic 2>"$out/bench_beam_$name.err" | tee "$out/bench_beam_$name.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['search']['ms_per_search_step'])" || tail -3 "$out/bench_beam_$name.err"
}
beam product X=1
beam wide ESPNET_AMD_LNG_WIDE=1024
beam product2 X=1
echo "== kernel stats (product)"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/search_stats" -o s --output-format csv -- python "$OLDPWD/bench.py" --workload beam --steps 1 --warmup 1 --no-cpu-baseline --no-traffic > "$OLDPWD/$out/search_stats.log" 2>&1)
find "$out/search_stats" -name "*_kernel_trace.csv" -delete
f=$(find "$out/search_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-150
