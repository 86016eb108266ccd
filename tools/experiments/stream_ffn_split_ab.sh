for fm in 1; do for v in 0 1 4 8 0 1; do
export ESPNET_AMD_STREAM_FUSED_MIN=$fm ESPNET_AMD_STREAM_FFN_SPLIT=$v
echo -n "fused_min=$fm ffn_split=$v: "
timeout 600 python bench.py --workload stream --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'one stream, call median', d['config']['call_latency_ms_median'], 'ms; batch32', d.get('batch32',{}).get('value'), 'batch128', d.get('batch128',{}).get('value'))"
done; done
