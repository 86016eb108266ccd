#!/bin/bash
# L2 hit rate of the block kernels in situ (greedy bench loop), with and without the L2 warm-up
set -u
out=gpurun_out/r02x
mkdir -p "$out"
for v in warmup nt; do
  lib=$PWD/espnet_amd/lib/libespnet_amd.so
  [ $v = nt ] && lib=$PWD/espnet_amd/lib/dbg/lib_nt.so
  (cd /tmp && export TMPDIR=/tmp && ESPNET_AMD_LIB=$lib timeout 60 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d "$OLDPWD/$out/pmc_$v" -o b --output-format csv -- \
     python "$OLDPWD/bench.py" --quick --no-traffic --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > "$OLDPWD/$out/pmc_$v.log" 2>&1)
  python tools/pmc_summary.py "$out/pmc_$v" --match block_kernel relpos_attn2 --source "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -- bench.py --quick --steps 20 ($v)" > "$out/l2_$v.json"
  rm -rf "$out/pmc_$v"
  python - "$out/l2_$v.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["kernels"].items():
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    print(sys.argv[1].split("/")[-1], k, "hit", h, "miss", m, "hit rate %.3f" % (h / max(1, h + m)))
PY
done
