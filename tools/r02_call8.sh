#!/bin/bash
# FFN loop: which part of an iteration costs what?  Developer builds of block.hip (EM_BLOCK_DBG bits:
# 4 = no barrier in the FFN iteration, 8 = no LDS traffic for H, 16 = no Swish) timed by tools/block_bench.py
set -u
out=gpurun_out/r02l
mkdir -p "$out"
echo "dbg 0" | tee "$out/ffn_parts.txt"
timeout 120 python tools/block_bench.py --iters 50 2>&1 | grep -E "block<" | tee -a "$out/ffn_parts.txt"
for d in 4 8 12 28; do
  echo "dbg $d" | tee -a "$out/ffn_parts.txt"
  ESPNET_AMD_LIB=$PWD/espnet_amd/lib/dbg/lib_$d.so timeout 120 python tools/block_bench.py --iters 50 2>&1 | grep -E "block<" | tee -a "$out/ffn_parts.txt"
done
