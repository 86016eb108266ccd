set -u
out=gpurun_out/r06ak; mkdir -p $out
timeout 120 python tools/null_stream_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/null_stream_probe.txt
(time timeout 600 python bench.py 2>$out/bench_default.err | tail -1 > $out/bench_default.json) 2>&1 | tail -4
python - $out/bench_default.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
def g(o, *ks):
    for k in ks:
        o = o.get(k, {}) if isinstance(o, dict) else {}
    return o
print("value", j.get("value"), "ms", j.get("ms_per_step"), "frac", g(j, "roofline", "frac"), "whole", g(j, "roofline", "whole_step", "frac_of_mfma_peak_over_wall_time"))
for k in ("pcie_inclusive", "one_stream", "f32_mode", "encoder_large_b64", "encoder_ebranchformer_b32", "beam", "beam_cfg3_per_gpu", "stream"):
    o = j.get(k, {})
    print(k, {kk: o.get(kk) for kk in ("value", "ms_per_step", "vs_resident", "error") if kk in o}, g(o, "search", "ms_per_search_step") or "")
st = j.get("stream", {})
for k in ("batch32", "batch128", "batch32_two_groups", "batch128_two_groups", "batch64_three_groups"):
    print(k, g(st, k).get("value"), g(st, k).get("ms_per_step"))
print("box_state", g(j, "box_state", "state"))
PY
