#!/bin/bash
# default bench first thing on the box, then the in-situ kernel stats (100 steps)
set -u
out=gpurun_out/${1:-r02s}
mkdir -p "$out"
(time timeout 700 python bench.py > "$out/bench.json" 2> "$out/bench.err") 2> "$out/bench.time"
cut -c1-300 "$out/bench.json"; tail -3 "$out/bench.time"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof_greedy" -o bench --output-format csv -- \
   python "$OLDPWD/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 100 --warmup 10 >"$OLDPWD/$out/prof_greedy.log" 2>&1)
rm -f "$out"/prof_*/*_kernel_trace.csv
ls "$out"
