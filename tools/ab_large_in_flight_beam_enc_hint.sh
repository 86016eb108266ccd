set -u
out=gpurun_out/${1:-r06ah}; mkdir -p $out
pr() { python -c "
import sys, json
j=json.loads(sys.stdin.read()); s=j.get('search') or {}
print('value', j.get('value'), 'ms_per_step', j.get('ms_per_step'), 'ms_per_search_step', s.get('ms_per_search_step'))"; }
for IF in 2 3 4; do
  echo "== large B=64 in_flight=$IF" | tee -a $out/ab.txt
  timeout 200 python bench.py --model large --batch 64 --steps 60 --warmup 10 --quick --no-cpu-baseline --no-traffic --no-roofline --in-flight $IF 2>$out/err_l64_$IF.txt | tail -1 | pr | tee -a $out/ab.txt
done
for E in 1 4; do
 for B in 16 64; do
  echo "== beam B=$B lanes=4 threaded BENCH_ENC_IN_FLIGHT=$E" | tee -a $out/ab.txt
  BENCH_ENC_IN_FLIGHT=$E BENCH_LANE_THREADS=1 timeout 300 python bench.py --workload beam --batch $B --steps 12 --warmup 1 --quick --no-cpu-baseline --no-traffic --in-flight 4 2>$out/err_beam_${B}_$E.txt | tail -1 | pr | tee -a $out/ab.txt
 done
done
