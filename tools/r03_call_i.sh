#!/bin/bash
# Round-3 GPU call I: timing attribution inside the pre-beam and tail kernels (EXPERIMENT switches: results wrong), tile A/B
set -u
tag=${1:-r03i}; out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
beam() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py --workload beam --steps 2 --warmup 1 --no-cpu-baseline 2>"$out/bench_beam_$name.err" | tee "$out/bench_beam_$name.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['search']['ms_per_search_step'])" || tail -3 "$out/bench_beam_$name.err"
}
beam product X=1
beam mid11 ESPNET_AMD_MID_TILE=11
beam mid21 ESPNET_AMD_MID_TILE=21
beam mid12 ESPNET_AMD_MID_TILE=12
for d in 1 2 4 8 15; do beam pre$d ESPNET_AMD_PREBEAM_DBG=$d; done
for d in 1 2 4 8 15; do beam tail$d ESPNET_AMD_TAIL_DBG=$d; done
beam product2 X=1
