# 512-wide encoders: q | k | v walked behind the macaron row-block launch (+ E-Branchformer's norm_mlp leaving the same launch) against the
# two projection GEMMs (+ a LayerNorm launch) (ESPNET_AMD_NO_ROWS_QKV=1)
set -u
out=gpurun_out/${1:-r06am}; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_ebranchformer.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -x -k "ebranchformer or large or ebf" 2>&1 | tail -12 | tee $out/pytest.txt
pr() { python -c "
import sys, json
j=json.loads(sys.stdin.read()); print('value', j.get('value'), 'ms_per_step', j.get('ms_per_step'))"; }
for cfg in "ebf 32 3" "ebf 32 2"; do
  set -- $cfg
  for NO in 1 0 1 0; do
    echo "== $1 B=$2 in_flight=$3 no_rows_qkv=$NO" | tee -a $out/ab.txt
    if [ $NO = 1 ]; then export ESPNET_AMD_NO_ROWS_QKV=1; else unset ESPNET_AMD_NO_ROWS_QKV; fi
    timeout 200 python bench.py --model $1 --batch $2 --steps 100 --warmup 10 --quick --no-cpu-baseline --no-traffic --no-roofline --in-flight $3 2>$out/err_$1_$2_$NO.txt | tail -1 | pr | tee -a $out/ab.txt
  done
done
