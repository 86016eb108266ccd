#!/bin/bash
# Which state is this GPU box in?  (Boxes of the pool run the same build at 1.06-1.09 or at ~1.45-1.5 ms per greedy step;
# DESIGN.md section 6.)  Prints the clocks rocm-smi reports, the workgroup -> XCD map of a 256-workgroup launch
# (tools/experiments/xcc_map.hip: the block kernels' L2 warm-up assumes linear id % 8) and a 300-step bench line.
export TMPDIR=/tmp
echo "== clocks"; (rocm-smi --showclocks 2>/dev/null | grep -iE "sclk|mclk|fclk|socclk" | head -8) || true
(rocm-smi --showperflevel --showpower 2>/dev/null | grep -iE "perf|power" | head -6) || true
(rocm-smi --showcomputepartition --showmemorypartition 2>/dev/null | grep -iE "partition" | head -4) || true
echo "== workgroup -> XCD map"; [ -x tools/bin/xcc_map ] && tools/bin/xcc_map 2>&1 | cut -c1-200
echo "== bench"; timeout 200 python bench.py --quick --no-traffic --no-cpu-baseline --no-roofline --steps 300 --warmup 30 2>/dev/null < /dev/null | cut -c100-200
# one short counter pass of the L2 (TCC) side: hit / miss / fabric read requests per block_kernel launch.  A slow-state box
# runs the weight-streaming kernels 1.5x slower with the ALU-bound frontend unchanged: the pair
# (fast, slow) of these summaries is what tells whether the L2 hit rate or the fabric latency differs.
if [ "${1:-}" != "--no-pmc" ]; then
  out=${BOX_STATE_OUT:-$PWD/gpurun_out/box_state}; mkdir -p "$out"; R=$PWD
  (cd /tmp && timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace -d "$out/tcc" -o p --output-format csv -- \
     python "$R/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 5 --warmup 2 > "$out/tcc.log" 2>&1 < /dev/null)
  find "$out/tcc" -name "*_kernel_trace.csv" -delete 2>/dev/null
  python tools/pmc_summary.py "$out/tcc" --match block_kernel sub2_kernel relpos_attn2 --source "box_state: bench.py --quick --steps 5" > "$out/tcc.json" 2>"$out/tcc.err"
  find "$out/tcc" -name "*counter_collection.csv" -delete 2>/dev/null
  echo "== TCC counters per launch"; python - "$out/tcc.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["kernels"]
for k, v in d.items():
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    print(f"  {k[:40]:40s} launches {v['launches']:4d} hit {h:10d} miss {m:9d} hit-rate {h / max(1, h + m):.3f} "
          f"EA rdreq {v.get('TCC_EA0_RDREQ_sum', 0):9d}")
PY
fi
echo "== clocks after"; (rocm-smi --showclocks 2>/dev/null | grep -iE "sclk|mclk|fclk|socclk" | head -8) || true
