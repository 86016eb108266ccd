#!/bin/bash
# Round-3 GPU call: block-kernel A/B variants (tools/build_block_variants.sh), sub-stage stamps, in-situ bench, and the
# counter passes of the beam-search label step.  Everything lands under gpurun_out/<tag>/.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r03_ab.sh r03a "v1 v4 v16 v21" "v21"'
set -u
tag=${1:-r03a}; variants=${2:-}; insitu=${3:-}
out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
D=espnet_amd/lib/dbg
echo "== product: kernel tests"; (time timeout 400 python -m pytest tests/test_gpu_block.py -q -x 2>&1 | tail -4) 2>&1 | tee "$out/pytest_block.txt"
echo "== block_bench product"; timeout 120 python tools/block_bench.py --iters 200 2>&1 | tee "$out/bb_product.txt"
for v in $variants; do
  echo "== block_bench $v"; ESPNET_AMD_LIB=$D/lib_$v.so timeout 120 python tools/block_bench.py --iters 200 2>&1 | grep "block<" | tee "$out/bb_$v.txt"
done
if [ -f $D/lib_fine.so ]; then
  echo "== fine stamps"; EM_BLOCK_STAMPS=1 ESPNET_AMD_LIB=$D/lib_fine.so timeout 120 python tools/block_bench.py --iters 1 > "$out/stamps_fine.txt" 2>&1; tail -n 40 "$out/stamps_fine.txt" | cut -c1-400
fi
for v in $insitu; do
  echo "== kernel tests $v"; ESPNET_AMD_LIB=$D/lib_$v.so timeout 400 python -m pytest tests/test_gpu_block.py -q -x 2>&1 | tail -3 | tee "$out/pytest_block_$v.txt"
done
echo "== in situ product"; timeout 300 python bench.py --quick --no-traffic --no-cpu-baseline --steps 600 --warmup 30 2>"$out/bench_product.err" | tee "$out/bench_product.json" | cut -c1-160
for v in $insitu; do
  echo "== in situ $v"; ESPNET_AMD_LIB=$D/lib_$v.so timeout 300 python bench.py --quick --no-traffic --no-cpu-baseline --steps 600 --warmup 30 2>"$out/bench_$v.err" | tee "$out/bench_$v.json" | cut -c1-160
done
echo "== rocprofv3 kernel stats, greedy (product)"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof_greedy" -o bench --output-format csv -- \
   python "$OLDPWD/bench.py" --quick --no-traffic --no-roofline --no-cpu-baseline --steps 100 --warmup 10 >"$OLDPWD/$out/prof_greedy.log" 2>&1)
rm -f "$out"/prof_greedy/*/*_kernel_trace.csv "$out"/prof_greedy/*_kernel_trace.csv
f=$(find "$out/prof_greedy" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
