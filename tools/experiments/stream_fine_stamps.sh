export ESPNET_AMD_LIB=$PWD/espnet_amd/lib/dbg/lib_fine.so
for n in 1 32; do echo "== $n streams, fine stamps"; EM_BLOCK_STAMPS=1 timeout 200 python tools/experiments/stream_split_sweep.py --one $n 2>&1 | grep "stamps" | tail -24 | cut -c1-420; done
export ESPNET_AMD_LIB=$PWD/espnet_amd/lib/dbg/lib_nt.so
for n in 1 32; do echo -n "no touch: "; timeout 200 python tools/experiments/stream_split_sweep.py --one $n 2>&1 | grep streams; done
unset ESPNET_AMD_LIB
for n in 1 32; do echo -n "product: "; timeout 200 python tools/experiments/stream_split_sweep.py --one $n 2>&1 | grep streams; done
for n in 1 32; do echo -n "no helpers: "; ESPNET_AMD_BLOCK_NO_HELPERS=1 timeout 200 python tools/experiments/stream_split_sweep.py --one $n 2>&1 | grep streams; done
