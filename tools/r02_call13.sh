#!/bin/bash
# why do some bench invocations report 1.56 ms and others 1.39 ms per step?  Same box, one variable at a time.
set -u
out=gpurun_out/r02r
mkdir -p "$out"
run() {  # label, args...
  l=$1; shift
  timeout 300 python bench.py "$@" 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$l /" | tee -a "$out/plain.txt"
}
run "quick no-roofline 300" --quick --no-traffic --no-cpu-baseline --no-roofline --steps 300 --warmup 20
run "quick roofline 300" --quick --no-traffic --no-cpu-baseline --steps 300 --warmup 20
run "quick roofline 1000 w30" --quick --no-traffic --no-cpu-baseline --steps 1000 --warmup 30
run "quick roofline 2000 w50" --quick --no-traffic --no-cpu-baseline --steps 2000 --warmup 50
run "quick no-roofline 2000 w50" --quick --no-traffic --no-cpu-baseline --no-roofline --steps 2000 --warmup 50
