#!/bin/bash
# plain (no profiler) quick bench with and without the L2 warm-up, then the in-situ kernel stats of the default build
set -u
out=gpurun_out/r02o
mkdir -p "$out"
for v in touch nt touch nt; do
  lib=$PWD/espnet_amd/lib/libespnet_amd.so
  [ $v = nt ] && lib=$PWD/espnet_amd/lib/dbg/lib_nt.so
  ESPNET_AMD_LIB=$lib timeout 200 python bench.py --quick --no-traffic --no-cpu-baseline --no-roofline --steps 300 --warmup 20 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$v /" | tee -a "$out/plain.txt"
done
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof" -o bench --output-format csv -- \
   python "$OLDPWD/bench.py" --quick --no-traffic --no-cpu-baseline --no-roofline --steps 100 --warmup 10 > "$OLDPWD/$out/bench_prof.log" 2>&1)
grep -o '"ms_per_step": [0-9.]*' "$out/bench_prof.log" | head -1
head -12 "$out"/prof/*kernel_stats.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | grep block_kernel | cut -c1-100
rm -f "$out"/prof/*kernel_trace.csv
