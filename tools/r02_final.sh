#!/bin/bash
# End-of-round validation on one MI355X box: full GPU parity suite, smoke, the default bench line, and the
# rocprofv3 kernel tables of the greedy main loop and of the beam workload.  Everything lands in gpurun_out/<tag>/.
set -u
tag=${1:-r02f}
out=gpurun_out/$tag
mkdir -p "$out"
(time timeout 1500 python -m pytest tests -m gpu -q -x) > "$out/pytest_gpu.txt" 2>&1
tail -4 "$out/pytest_gpu.txt"
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee "$out/smoke.txt"
(time timeout 700 python bench.py > "$out/bench.json" 2> "$out/bench.err") 2> "$out/bench.time"
cut -c1-300 "$out/bench.json"; tail -3 "$out/bench.time"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof_greedy" -o bench --output-format csv -- \
   python "$OLDPWD/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 20 --warmup 5 >"$OLDPWD/$out/prof_greedy.log" 2>&1)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof_beam" -o beam --output-format csv -- \
   python "$OLDPWD/bench.py" --workload beam --quick --no-cpu-baseline --steps 2 --warmup 1 >"$OLDPWD/$out/prof_beam.log" 2>&1)
rm -f "$out"/prof_*/*_kernel_trace.csv
ls "$out" "$out"/prof_greedy "$out"/prof_beam
