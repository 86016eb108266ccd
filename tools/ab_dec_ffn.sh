set -u
out=gpurun_out/r06ac; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "dec_ffn" 2>&1 | tail -15 | tee $out/pytest_dec_ffn.txt
timeout 600 python -m pytest tests/test_gpu_search.py tests/test_gpu_fullsize.py tests/test_gpu_online_search.py tests/test_gpu_scorer_interface.py -q -x 2>&1 | tail -15 | tee $out/pytest_search.txt
for B in 64 16; do
  for S in 1 0 512 256 128; do
    echo "== B=$B ESPNET_AMD_DEC_FFN_SPLIT=$S" | tee -a $out/ab.txt
    ESPNET_AMD_DEC_FFN_SPLIT=$S timeout 200 python bench.py --workload beam --batch $B --steps 2 --warmup 1 --quick --no-cpu-baseline --no-traffic --in-flight 1 2>$out/err_${B}_${S}.txt | tail -1 | python -c "
import sys, json
j=json.loads(sys.stdin.read()); s=j.get('search') or j.get('beam',{}).get('search') or {}
print('value', j.get('value'), 'ms_per_step', j.get('ms_per_step'), 'ms_per_search_step', s.get('ms_per_search_step'))" | tee -a $out/ab.txt
  done
done
