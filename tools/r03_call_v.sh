#!/bin/bash
# Round-3 GPU call V: rows per workgroup of the decoder self-attention (label step A/B) + search tests.
set -u
tag=${1:-r03v}; out=$PWD/gpurun_out/$tag; mkdir -p "$out"
for g in 10 5 2 1 0; do
  echo "== SA_GROUP=$g (0 = default rule)"
  if [ $g = 0 ]; then unset ESPNET_AMD_SA_GROUP; else export ESPNET_AMD_SA_GROUP=$g; fi
  timeout 300 python bench.py --workload beam --steps 2 --warmup 1 --no-cpu-baseline --no-traffic 2>"$out/beam_$g.err" < /dev/null | tee "$out/bench_beam_g$g.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['search']['ms_per_search_step'])" || tail -3 "$out/beam_$g.err"
done
unset ESPNET_AMD_SA_GROUP
timeout 600 python -m pytest tests/test_gpu_search.py tests/test_gpu_scorer_interface.py -q -x 2>&1 | tail -3
