#!/bin/bash
set -u
out=gpurun_out/r03f; mkdir -p $out; export TMPDIR=/tmp
echo "== kernel + search tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_search.py tests/test_gpu_scorer_interface.py tests/test_gpu_online_search.py -q -x 2>&1 | tail -4 | tee $out/pytest.txt
for g in 1 0; do
echo "== greedy bench GEMM256=$g"; ESPNET_AMD_GEMM256=$g timeout 300 python bench.py --quick --no-traffic --no-cpu-baseline --steps 600 --warmup 30 2>$out/bench_g$g.err | tee $out/bench_g$g.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['families'])"
done
echo "== e2e fullsize greedy"; timeout 400 python -m pytest tests/test_gpu_fullsize.py::test_greedy_b32_matches_oracle_elementwise -q -x 2>&1 | tail -3
bash tools/r03_search.sh r03f --profile
