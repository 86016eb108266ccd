#!/bin/bash
# SQ counters of HEAD's block / attention kernels (tools/block_bench.py under rocprofv3 --pmc)
set -u
out=gpurun_out/r02w
mkdir -p "$out"
(cd /tmp && export TMPDIR=/tmp && timeout 100 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT --kernel-trace -d "$OLDPWD/$out/pmc" -o bb --output-format csv -- python "$OLDPWD/tools/block_bench.py" --iters 10 > "$OLDPWD/$out/pmc.log" 2>&1)
python tools/pmc_summary.py "$out/pmc" --source "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT --kernel-trace -- python tools/block_bench.py --iters 10 (HEAD: direct fragment loads; per-launch averages, 4 waves per workgroup)" > "$out/pmc_block_sq.json"
rm -rf "$out/pmc"
head -c 1500 "$out/pmc_block_sq.json"
