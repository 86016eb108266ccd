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
echo "== clocks after"; (rocm-smi --showclocks 2>/dev/null | grep -iE "sclk|mclk|fclk|socclk" | head -8) || true
