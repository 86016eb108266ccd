#!/bin/bash
# End-of-round validation of HEAD on one MI355X box (see tools/r02_final.sh) plus the recipe-level CLI bench, the
# E-Branchformer line and the streaming-with-online-beam line.
set -u
tag=${1:-r02g}
out=gpurun_out/$tag
mkdir -p "$out"
(time timeout 1500 python -m pytest tests -m gpu -q -x) > "$out/pytest_gpu.txt" 2>&1
tail -4 "$out/pytest_gpu.txt"
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -3 | tee "$out/smoke.txt"
(time timeout 700 python bench.py > "$out/bench.json" 2> "$out/bench.err") 2> "$out/bench.time"
cut -c1-300 "$out/bench.json"; tail -3 "$out/bench.time"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof_greedy" -o bench --output-format csv -- \
   python "$OLDPWD/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 20 --warmup 5 >"$OLDPWD/$out/prof_greedy.log" 2>&1)
rm -f "$out"/prof_*/*_kernel_trace.csv
timeout 300 python tools/cli_bench.py --n 2048 --batch-size 32 --workers 8 2>&1 | tail -4 | tee "$out/cli_bench.txt"
timeout 200 python bench.py --model ebf --quick --no-traffic --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 > "$out/bench_ebf.json"; cut -c1-200 "$out/bench_ebf.json"
timeout 200 python bench.py --workload stream --stream-beam 10 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > "$out/bench_stream_beam10.json"; cut -c1-200 "$out/bench_stream_beam10.json"
ls "$out"
