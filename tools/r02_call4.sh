#!/bin/bash
# rocprofv3 kernel table of the beam workload (configs[2])
set -u
out=gpurun_out/r02e
mkdir -p "$out"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof" -o beam --output-format csv -- \
   python "$OLDPWD/bench.py" --workload beam --quick --no-cpu-baseline --steps 2 --warmup 1 >"$OLDPWD/$out/prof_beam.log" 2>&1)
ls "$out/prof" | head; tail -c 400 "$out/prof_beam.log"
rm -f "$out/prof/beam_kernel_trace.csv"
