set -u
out=gpurun_out/${1:-r06aj}; mkdir -p $out
pr() { python -c "
import sys, json
j=json.loads(sys.stdin.read())
for k in ('beam','beam_cfg3_per_gpu'):
    b=j.get(k)
    if b: print(k, b.get('value'), b.get('ms_per_step'), (b.get('search') or {}).get('ms_per_search_step'), b.get('error'))"; }
for L in beam_cfg3_per_gpu beam,beam_cfg3_per_gpu encoder_large_b64,encoder_ebranchformer_b32,beam_cfg3_per_gpu; do
  echo "== BENCH_ONLY_LEGS=$L --no-traffic" | tee -a $out/ab.txt
  BENCH_ONLY_LEGS=$L timeout 400 python bench.py --steps 100 --no-traffic 2>$out/err_$(echo $L | tr ',' '_').txt | tail -1 | pr | tee -a $out/ab.txt
done
echo "== BENCH_ONLY_LEGS=beam,beam_cfg3_per_gpu with traffic" | tee -a $out/ab.txt
BENCH_ONLY_LEGS=beam,beam_cfg3_per_gpu timeout 400 python bench.py --steps 100 2>$out/err_traffic.txt | tail -1 | pr | tee -a $out/ab.txt
