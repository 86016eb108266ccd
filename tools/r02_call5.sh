#!/bin/bash
# SQ counters of the final block / attention kernels (tools/block_bench.py) and the GEMM micro-benchmark table
set -u
out=gpurun_out/r02h
mkdir -p "$out"
timeout 120 python tools/block_bench.py --iters 50 2>&1 | grep -E "block<|relpos" | tee "$out/block_bench.txt"
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT --kernel-trace -d "$OLDPWD/$out/pmc" -o bb --output-format csv -- python "$OLDPWD/tools/block_bench.py" --iters 10 > "$OLDPWD/$out/pmc.log" 2>&1)
rm -f "$out"/pmc/*kernel_trace.csv
timeout 200 python tools/gemm_bench.py 2>&1 | tail -30 | tee "$out/gemm_bench.txt"
ls "$out" "$out/pmc"
