# A/B of the label step's fragment-major projections (round 6): ESPNET_AMD_DEC_FFN_SPLIT=1 = off (row-major matrices);
# ESPNET_AMD_DEC_FFN_ROWS=16|32 = rows per workgroup of ln_frag_gemm_kernel (default: 32 from 320 rows)
set -u
out=gpurun_out/${1:-r06ad}; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "dec_ffn or ln_gemm_frag" 2>&1 | tail -15 | tee $out/pytest_dec_ffn.txt
timeout 600 python -m pytest tests/test_gpu_search.py tests/test_gpu_fullsize.py tests/test_gpu_online_search.py tests/test_gpu_scorer_interface.py -q -x 2>&1 | tail -15 | tee $out/pytest_search.txt
pr() { python -c "
import sys, json
j=json.loads(sys.stdin.read()); s=j.get('search') or {}
print('value', j.get('value'), 'ms_per_step', j.get('ms_per_step'), 'ms_per_search_step', s.get('ms_per_search_step'))"; }
for B in 64 32 16; do
  for V in "1 0" "0 16" "0 32" "1 0" "0 16" "0 32"; do
    set -- $V
    echo "== B=$B ESPNET_AMD_DEC_FFN_SPLIT=$1 ESPNET_AMD_DEC_FFN_ROWS=$2" | tee -a $out/ab.txt
    ESPNET_AMD_DEC_FFN_SPLIT=$1 ESPNET_AMD_DEC_FFN_ROWS=$2 timeout 200 python bench.py --workload beam --batch $B --steps 2 --warmup 1 --quick --no-cpu-baseline --no-traffic --in-flight 1 2>$out/err_${B}_$1_$2.txt | tail -1 | pr | tee -a $out/ab.txt
  done
done
cd /tmp && export TMPDIR=/tmp
for B in 64; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof$B -o s -- python /root/repo/bench.py --workload beam --batch $B --steps 1 --warmup 1 --quick --no-cpu-baseline --no-traffic --in-flight 1 > /root/repo/$out/prof$B.log 2>&1
  f=$(find /root/repo/$out/prof$B -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f /root/repo/$out/search_rows$((B*10))_kernel_stats.csv && head -12 /root/repo/$out/search_rows$((B*10))_kernel_stats.csv | cut -c1-150
  rm -rf /root/repo/$out/prof$B
done
