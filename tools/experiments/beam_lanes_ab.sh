# joint searches in flight: lanes x stream choice (picked = streams probed to run side by side, plain = torch.cuda.Stream() as they come)
for st in picked plain; do for n in 2 3 4; do
  if [ $st = plain ]; then export BENCH_LANE_STREAMS=plain; else unset BENCH_LANE_STREAMS; fi
  echo -n "$st lanes=$n: "
  timeout 600 python bench.py --workload beam --in-flight $n --no-cpu-baseline --no-traffic --steps 4 --warmup 1 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'audio-s/s', d['ms_per_step'], 'ms per batch')"
done; done
