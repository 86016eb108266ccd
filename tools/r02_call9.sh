#!/bin/bash
# in-situ per-kernel times of the block kernels (rocprofv3 --stats over a short bench run), with and without the
# L2 warm-up touches
set -u
out=gpurun_out/r02m
mkdir -p "$out"
for v in touch nt; do
  lib=$PWD/espnet_amd/lib/libespnet_amd.so
  [ $v = nt ] && lib=$PWD/espnet_amd/lib/dbg/lib_nt.so
  (cd /tmp && export TMPDIR=/tmp && ESPNET_AMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof_$v" -o bench --output-format csv -- \
     python "$OLDPWD/bench.py" --quick --no-traffic --no-cpu-baseline --no-roofline --steps 100 --warmup 10 > "$OLDPWD/$out/bench_$v.log" 2>&1)
  f=$(ls "$out"/prof_$v/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $v" | tee -a "$out/summary.txt"
  tail -1 "$out/bench_$v.log" | cut -c1-200 | tee -a "$out/summary.txt"
  [ -n "$f" ] && head -9 "$f" | cut -d, -f1-4 | sed 's/(EmBlockArgs.*)"/"/; s/(anonymous namespace):://g' | cut -c1-120 | tee -a "$out/summary.txt"
  rm -f "$out"/prof_$v/*kernel_trace.csv
done
