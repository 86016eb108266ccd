#!/bin/bash
# rocprofv3 kernel table of the bench main loop + the full default bench line
set -u
out=gpurun_out/r02d
mkdir -p "$out"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof" -o bench --output-format csv -- \
   python "$OLDPWD/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 20 --warmup 5 >"$OLDPWD/$out/prof_bench.log" 2>&1)
ls "$out/prof" | head
(time timeout 700 python bench.py > "$out/bench.json" 2> "$out/bench.err") 2> "$out/bench.time"
tail -c 300 "$out/bench.err"; cat "$out/bench.time"; cut -c1-400 "$out/bench.json"
