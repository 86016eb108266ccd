# usage: bash tools/experiments/stream_variants_ab.sh <variant> ...   (espnet_amd/lib/dbg/lib_<variant>.so against the product, 1 / 32 / 128 streams, twice)
for rep in 1 2; do for v in product "$@"; do
  if [ $v = product ]; then unset ESPNET_AMD_LIB; else export ESPNET_AMD_LIB=$PWD/espnet_amd/lib/dbg/lib_$v.so; fi
  echo -n "$v:"; for n in 1 32 128; do timeout 200 python tools/experiments/stream_split_sweep.py --one $n 2>&1 | grep streams | sed 's/ us per call//' | tr '\n' ' '; done; echo
done; done
