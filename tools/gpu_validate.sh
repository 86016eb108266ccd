#!/bin/bash
# One-call GPU validation for a gpurun box: GPU parity suite, smoke, bench line, and (optionally) the rocprofv3
# kernel table of the bench command.  Writes everything under gpurun_out/<tag>/ so it is merged back.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/gpu_validate.sh r02a [--profile]'
# Afterwards copy what should be judged into profiles/ (see profiles/README.md).
set -u
tag=${1:-run}
out=gpurun_out/$tag
mkdir -p "$out"
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee "$out/pytest_gpu.txt"
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -3 | tee "$out/smoke.txt"
timeout 200 python bench.py 2>"$out/bench.err" | tail -1 | tee "$out/bench_small_b32.json" | cut -c1-300
if [ "${2:-}" = "--profile" ]; then
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof" -o bench -- \
     python "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline >"$OLDPWD/$out/prof_bench.log" 2>&1)
  ls -R "$out/prof" | head -20   # summarise with tools/prof_summary.py (see its docstring) once merged back
fi
