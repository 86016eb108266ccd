(timeout 600 python -m pytest -q -x tests/test_gpu_streaming.py tests/test_gpu_online_search.py 2>&1 | tail -6)
for rep in 1 2; do for v in 0 1; do
  if [ $v = 1 ]; then export ESPNET_AMD_STREAM_SPLIT_ATT=1; else unset ESPNET_AMD_STREAM_SPLIT_ATT; fi
  for n in 1 32 128; do echo -n "split_att=$v: "; timeout 200 python tools/experiments/stream_split_sweep.py --one $n 2>&1 | grep streams; done
done; done
