set -u
out=gpurun_out/${1:-r06ax}; mkdir -p $out
timeout 700 python -m pytest tests/test_gpu_search.py tests/test_gpu_fullsize.py tests/test_gpu_scorer_interface.py tests/test_gpu_cli.py -q -x 2>&1 | tail -6 | tee $out/pytest.txt
pr() { python -c "
import sys, json
j=json.loads(sys.stdin.read()); s=j.get('search') or {}
print('value', j.get('value'), 'ms_per_step', j.get('ms_per_step'), 'ms_per_search_step', s.get('ms_per_search_step'))"; }
for B in 64 16; do
  for i in nofrag frag nofrag frag; do if [ $i = nofrag ]; then export ESPNET_AMD_NO_MEM_FRAG=1; else unset ESPNET_AMD_NO_MEM_FRAG; fi
    echo "== B=$B run $i" | tee -a $out/ab.txt
    timeout 200 python bench.py --workload beam --batch $B --steps 2 --warmup 1 --quick --no-cpu-baseline --no-traffic --in-flight 1 2>$out/err_${B}_$i.txt | tail -1 | pr | tee -a $out/ab.txt
  done
done
