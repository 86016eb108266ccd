set -u
out=gpurun_out/${1:-r06al}; mkdir -p $out
pr() { python -c "
import sys, json
j=json.loads(sys.stdin.read()); s=j.get('search') or {}
print('value', j.get('value'), 'ms_per_step', j.get('ms_per_step'), 'ms_per_search_step', s.get('ms_per_search_step'))"; }
for B in 16; do
 for IF in 8 12 16 8 12; do
  echo "== beam B=$B threaded lanes=$IF" | tee -a $out/ab.txt
  BENCH_LANE_THREADS=1 timeout 300 python bench.py --workload beam --batch $B --steps 24 --warmup 1 --quick --no-cpu-baseline --no-traffic --in-flight $IF 2>$out/err_beam_${B}_$IF.txt | tail -1 | pr | tee -a $out/ab.txt
 done
done
