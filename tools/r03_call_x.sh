#!/bin/bash
# Round-3 GPU call X: decoder self-attention, waves per row x rows per workgroup (label step A/B).
set -u
for cfg in "1 10" "1 2" "2 2" "2 1" "4 1" "4 2" "8 1"; do
  set -- $cfg
  export ESPNET_AMD_SA_SPLIT=$1 ESPNET_AMD_SA_GROUP=$2
  echo -n "split $1 group $2: "
  timeout 300 python bench.py --workload beam --steps 2 --warmup 1 --no-cpu-baseline --no-traffic 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['search']['ms_per_search_step'])"
done
