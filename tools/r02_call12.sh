#!/bin/bash
# same box, same build: does the step time depend on the length of the run?  (L2 warm-up on / off, 300 / 2000 steps)
set -u
out=gpurun_out/r02q
mkdir -p "$out"
run() {  # $1 = touch|nt, $2 = steps
  lib=$PWD/espnet_amd/lib/libespnet_amd.so
  [ $1 = nt ] && lib=$PWD/espnet_amd/lib/dbg/lib_nt.so
  ESPNET_AMD_LIB=$lib timeout 200 python bench.py --quick --no-traffic --no-cpu-baseline --no-roofline --steps $2 --warmup 20 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$1 $2 /" | tee -a "$out/plain.txt"
}
run touch 300; run touch 2000; run nt 300; run nt 2000; run touch 300; run touch 4000
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4 | tee -a "$out/plain.txt"
