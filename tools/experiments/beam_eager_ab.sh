# threaded lanes: label steps replayed from a hipGraph (K steps per graph) against eager launches from em_search_steps
for b in 16 64; do for e in 0 1 0 1; do
  if [ $e = 1 ]; then export BENCH_SEARCH_EAGER=1; else unset BENCH_SEARCH_EAGER; fi
  echo -n "B=$b eager=$e: "
  timeout 900 python bench.py --workload beam --batch $b --no-cpu-baseline --no-traffic 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'audio-s/s', d['ms_per_step'], 'ms per batch; one search alone', d['search']['ms_per_search_step'], 'ms per label step')"
done; done
