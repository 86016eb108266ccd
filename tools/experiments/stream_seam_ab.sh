# split-FFN meeting point: plain stores / loads + release / acquire fences (product); lib_v8192: agent-scope (sc1) stores, no release fence;
# lib_v4096: agent-scope loads, no acquire fence; lib_v12288: both (bash tools/build_block_variants.sh v4096 v8192 v12288)
for rep in 1 2; do for v in product v4096 v8192 v12288; do
  if [ $v = product ]; then unset ESPNET_AMD_LIB; else export ESPNET_AMD_LIB=$PWD/espnet_amd/lib/dbg/lib_$v.so; fi
  for n in 1 32; do echo -n "$v: "; timeout 200 python tools/experiments/stream_split_sweep.py --one $n 2>&1 | grep streams; done
done; done
unset ESPNET_AMD_LIB
for n in 1 32; do EM_BLOCK_STAMPS=1 timeout 200 python tools/experiments/stream_split_sweep.py --one $n 2>&1 | grep "stamps" | tail -8 | cut -c1-300; done
