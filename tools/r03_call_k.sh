#!/bin/bash
# Round-3 GPU call K: fused conv1 + conv2 kernel (csrc/subsample2.hip): kernel test, end-to-end tests, greedy bench A/B, profile
set -u
tag=${1:-r03k}; out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
echo "== kernel test"; timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "sub12 or conv" 2>&1 | tail -8 | tee "$out/pytest_sub12.txt"
echo "== e2e / fullsize / block tests"; (time timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_block.py -q -x 2>&1 | tail -8) 2>&1 | tee "$out/pytest_e2e.txt"
greedy() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py --quick --no-traffic --no-cpu-baseline --steps 600 --warmup 30 2>"$out/bench_$name.err" | tee "$out/bench_$name.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['roofline']['families'])" || tail -5 "$out/bench_$name.err"
}
greedy fused X=1
greedy unfused ESPNET_AMD_NO_SUB12=1
echo "== rocprofv3 kernel stats, greedy (fused)"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof_greedy" -o bench --output-format csv -- \
   python "$OLDPWD/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 100 --warmup 10 >"$OLDPWD/$out/prof_greedy.log" 2>&1)
find "$out/prof_greedy" -name "*_kernel_trace.csv" -delete
f=$(find "$out/prof_greedy" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-180
