# joint searches in flight through the bench's own dispatch loop (decode_dynamic_lanes): one host thread / a host thread per lane
for b in 16 64; do for cfg in "2" "2 --lane-threads" "3 --lane-threads" "4 --lane-threads" "6 --lane-threads"; do
  echo -n "B=$b in-flight $cfg: "
  timeout 900 python bench.py --workload beam --batch $b --in-flight $cfg --no-cpu-baseline --no-traffic --steps 6 --warmup 1 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'audio-s/s', d['ms_per_step'], 'ms per batch')"
done; done
