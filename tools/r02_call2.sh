#!/bin/bash
set -u
out=gpurun_out/r02b
mkdir -p "$out"
for d in 0 1 2 3 4 8 10 11 15; do echo "== dbg $d"; timeout 100 python tools/block_bench.py --dbg $d --iters 50 2>&1 | grep block; done | tee "$out/ablation.txt"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT --kernel-trace -d "$OLDPWD/$out/pmc" -o bb --output-format csv -- python "$OLDPWD/tools/block_bench.py" --iters 10 > "$OLDPWD/$out/pmc.log" 2>&1
cd "$OLDPWD"; ls -R $out/pmc | head; python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/r02b/pmc/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
        if r['Counter_Name']=='SQ_WAVE_CYCLES': cnt[k]+=1
    for k,v in agg.items():
        if 'block' in k or 'attn' in k:
            n=max(cnt[k],1); print(k, n, {c: round(x/n) for c,x in v.items()})
PY
