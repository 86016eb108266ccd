# split-FFN meeting point: plain stores + fences (lib_v1024), agent-scope stores / loads + fences (lib_v2048), agent-scope stores / loads, no fences (product)
for rep in 1 2; do for v in product v1024 v2048; do
  if [ $v = product ]; then unset ESPNET_AMD_LIB; else export ESPNET_AMD_LIB=$PWD/espnet_amd/lib/dbg/lib_$v.so; fi
  for n in 1 32; do echo -n "$v: "; timeout 200 python tools/experiments/stream_split_sweep.py --one $n 2>&1 | grep streams; done
done; done
unset ESPNET_AMD_LIB
for n in 1 32; do EM_BLOCK_STAMPS=1 timeout 200 python tools/experiments/stream_split_sweep.py --one $n 2>&1 | grep "stamps" | tail -8 | cut -c1-300; done
