#!/bin/bash
# same box: static (workgroup id % 8) share vs XCC_ID + arrival-counter share vs no warm-up
set -u
out=gpurun_out/r02u
mkdir -p "$out"
run() {
  lib=$PWD/espnet_amd/lib/libespnet_amd.so
  [ $1 != xcc ] && lib=$PWD/espnet_amd/lib/dbg/lib_$1.so
  ESPNET_AMD_LIB=$lib timeout 200 python bench.py --quick --no-traffic --no-cpu-baseline --no-roofline --steps 300 --warmup 20 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$1 /" | tee -a "$out/plain.txt"
}
run v3; run xcc; run nt; run v3; run xcc
