#!/bin/bash
# One-call validation of the tree on a gpurun box: the whole GPU suite, smoke, the default bench line (all legs).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_validate_full.sh r06ai'
set -u
tag=${1:-run}; out=gpurun_out/$tag; mkdir -p "$out"
echo "== pytest -m gpu"; (time timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8) 2>&1 | tee "$out/pytest_gpu.txt"
echo "== smoke"; timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -5 | tee "$out/smoke.txt"
echo "== default bench"; (time timeout 600 python bench.py 2>"$out/bench_default.err" | tail -1 > "$out/bench_default.json") 2>&1 | tail -4
cut -c1-400 "$out/bench_default.json"
python - "$out/bench_default.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
def g(o, *ks):
    for k in ks:
        o = o.get(k, {}) if isinstance(o, dict) else {}
    return o
print("value", j.get("value"), "ms", j.get("ms_per_step"), "frac", g(j, "roofline", "frac"), "whole", g(j, "roofline", "whole_step", "frac_of_mfma_peak_over_wall_time"))
for k in ("pcie_inclusive", "one_stream", "encoder_large_b64", "encoder_ebranchformer_b32", "beam", "beam_cfg3_per_gpu", "stream"):
    o = j.get(k, {})
    print(k, {kk: o.get(kk) for kk in ("value", "ms_per_step", "vs_resident") if kk in o}, g(o, "search", "ms_per_search_step") or "")
print("box_state", g(j, "box_state", "state"))
PY
