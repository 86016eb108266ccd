#!/bin/bash
# Counter evidence for the beam-search label step (measure before rewriting): per-kernel durations
# (kernel-trace --stats), SQ wave-cycle breakdown, and HBM traffic (FETCH_SIZE / WRITE_SIZE in their own passes, as
# MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes) of `bench.py --workload beam` (configs[2]: B = 16, beam 10).
#   bash tools/r03_search_pmc.sh <tag>      -> gpurun_out/<tag>/search_*
set -u
tag=${1:-r03a}; out=$PWD/gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
cmd="python $PWD/bench.py --workload beam --steps 1 --warmup 1 --no-cpu-baseline"
cd /tmp
echo "== search: kernel stats"
timeout 400 rocprofv3 --kernel-trace --stats -d "$out/search_stats" -o s --output-format csv -- $cmd > "$out/search_stats.log" 2>&1
find "$out/search_stats" -name "*_kernel_trace.csv" -delete
f=$(find "$out/search_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-220
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  echo "== search: pmc pass $i: $ctrs"
  ESPNET_AMD_SEARCH_GRAPH=0 timeout 600 rocprofv3 --pmc $ctrs --kernel-trace -d "$out/search_pmc$i" -o p --output-format csv -- $cmd > "$out/search_pmc$i.log" 2>&1
  find "$out/search_pmc$i" -name "*_kernel_trace.csv" -delete
  python "$OLDPWD/tools/pmc_summary.py" "$out/search_pmc$i" --match "" --source "bench.py --workload beam --steps 1 --warmup 1, pass $i" > "$out/search_pmc$i.json" 2>"$out/search_pmc$i.err"
  find "$out/search_pmc$i" -name "*counter_collection.csv" -delete
  head -c 1500 "$out/search_pmc$i.json"
done
