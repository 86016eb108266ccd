#!/bin/bash
# end-of-round numbers for the committed build: default bench (roofline + traffic + cpu baseline), in-situ kernel
# stats of the same workload, the beam-search workload
set -u
out=gpurun_out/${1:-r02p}
mkdir -p "$out"
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -3 | tee "$out/smoke.txt"
(time timeout 700 python bench.py > "$out/bench.json" 2> "$out/bench.err") 2> "$out/bench.time"
cut -c1-300 "$out/bench.json"; tail -3 "$out/bench.time"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof_greedy" -o bench --output-format csv -- \
   python "$OLDPWD/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 20 --warmup 5 >"$OLDPWD/$out/prof_greedy.log" 2>&1)
rm -f "$out"/prof_*/*_kernel_trace.csv
timeout 300 python bench.py --workload beam --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > "$out/bench_beam.json"; cut -c1-260 "$out/bench_beam.json"
ls "$out"
