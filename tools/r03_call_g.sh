#!/bin/bash
# Round-3 GPU call G: the label-step work (gemm_mid, one-pass self-attention, up-front source attention, DPP arg-max
# rounds, fused pre-beam + candidates, fused tail, branch-free epilogues, batched skinny GEMM).
#   bash tools/r03_call_g.sh <tag>
set -u
tag=${1:-r03g}; out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
echo "== tests (kernels, search, scorers, online, streaming)"
(time timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_search.py tests/test_gpu_scorer_interface.py tests/test_gpu_online_search.py tests/test_gpu_streaming.py -q -x 2>&1 | tail -15) 2>&1 | tee "$out/pytest_search.txt"
beam() {  # name, env...
  local name=$1; shift
  echo "== beam bench: $name"
  env "$@" timeout 300 python bench.py --workload beam --steps 3 --warmup 1 --no-cpu-baseline 2>"$out/bench_beam_$name.err" | tee "$out/bench_beam_$name.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['search']['ms_per_search_step'])" || tail -5 "$out/bench_beam_$name.err"
}
beam product X=1
echo "== beam kernel stats (product)"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/search_stats" -o s --output-format csv -- python "$OLDPWD/bench.py" --workload beam --steps 1 --warmup 1 --no-cpu-baseline > "$OLDPWD/$out/search_stats.log" 2>&1)
find "$out/search_stats" -name "*_kernel_trace.csv" -delete
f=$(find "$out/search_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f" | cut -c1-200
echo "== stream bench"
timeout 300 python bench.py --workload stream --steps 3 --warmup 1 --no-cpu-baseline 2>"$out/bench_stream.err" | tee "$out/bench_stream.json" | cut -c1-400 || tail -5 "$out/bench_stream.err"
echo "== cfg3 per-GPU batch (64 x beam 10)"
timeout 300 python bench.py --workload beam --batch 64 --steps 1 --warmup 1 --no-cpu-baseline 2>"$out/bench_beam_b64.err" | tee "$out/bench_beam_b64.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['search']['ms_per_search_step'])" || tail -5 "$out/bench_beam_b64.err"
