#!/bin/bash
# Developer builds of csrc/block.hip next to the product library, selected with ESPNET_AMD_LIB=... (espnet_amd/lib.py):
#   espnet_amd/lib/dbg/lib_nt.so     -DEM_BLOCK_NO_TOUCH   (no L2 warm-up)
#   espnet_amd/lib/dbg/lib_<d>.so    -DEM_BLOCK_DBG=<d>    (1 no MFMA / epilogue, 4 no FFN barrier, 8 no H exchange, 16 no Swish)
#   espnet_amd/lib/dbg/lib_v<n>.so   -DEM_BLOCK_VAR=<n>    (A/B variants: 1 pinned-group FFN iteration, 4 FFN chunk order
#                                                           rotated per utterance, 16 packed-f32 depthwise conv; sums combine)
#   espnet_amd/lib/dbg/lib_fine.so   -DEM_BLOCK_FINE=1     (EM_BLOCK_STAMPS=1 prints sub-stage stamps of all four waves)
# Used for profiles/r02l, r02m, r02o, r02q and the round-3 A/B calls (tools/gpu_calls.sh).  dbg builds give wrong results by
# design: timing only.   usage: bash tools/build_block_variants.sh nt 4 v1 v16 fine
set -eu
cd "$(dirname "$0")/.."
python -m espnet_amd.build >/dev/null
mkdir -p espnet_amd/lib/dbg
objs=$(ls espnet_amd/lib/*.o | grep -v "/block.o")
for v in "$@"; do
  case "$v" in
    ffn*)  # csrc/ffn_rows.hip -DEM_FFN_DBG=<d>: 1 no MFMAs, 2 no weight requests, 4 no LDS operand reads, 8 cycle stamps
      o2=$(ls espnet_amd/lib/*.o | grep -v "/ffn_rows.o")
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -DEM_FFN_DBG=${v#ffn} \
        -Iinclude -Iespnet_amd/csrc -c espnet_amd/csrc/ffn_rows.hip -o espnet_amd/lib/dbg/ffn_rows_$v.o 2>/dev/null
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o espnet_amd/lib/dbg/lib_$v.so $o2 espnet_amd/lib/dbg/ffn_rows_$v.o
      rm -f espnet_amd/lib/dbg/ffn_rows_$v.o
      echo "built espnet_amd/lib/dbg/lib_$v.so"; continue ;;
    nt) def="-DEM_BLOCK_NO_TOUCH=1" ;;
    finent) def="-DEM_BLOCK_FINE=1 -DEM_BLOCK_NO_TOUCH=1" ;;
    fine*) def="-DEM_BLOCK_FINE=1 -DEM_BLOCK_VAR=${v#fine}"; [ "$v" = fine ] && def="-DEM_BLOCK_FINE=1" ;;
    v*) def="-DEM_BLOCK_VAR=${v#v}" ;;
    *) def="-DEM_BLOCK_DBG=$v" ;;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form \
    $def -Iinclude -Iespnet_amd/csrc -c espnet_amd/csrc/block.hip -o espnet_amd/lib/dbg/block_$v.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o espnet_amd/lib/dbg/lib_$v.so $objs espnet_amd/lib/dbg/block_$v.o
  rm -f espnet_amd/lib/dbg/block_$v.o
  echo "built espnet_amd/lib/dbg/lib_$v.so"
done
