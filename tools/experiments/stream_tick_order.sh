# launch order of one steady-state tick of 32 lock-step streams (rocprofv3 kernel trace of tools/stream_ab.py batch), or - second argument "one" - of one call of one stream through the hipGraph step
export TMPDIR=/tmp; R=$PWD; d=$R/gpurun_out/${1:-r06w}/tick_trace; mkdir -p $d
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d "$d" -o s --output-format csv -- python "$R/tools/stream_ab.py" ${2:-batch} > "$d.log" 2>&1 < /dev/null)
t=$(find "$d" -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "frontend_logmel" in n]
lo, hi = idx[-6], idx[-5]
t0 = int(rows[lo]["Start_Timestamp"])
prev_e = t0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_e) / 1e3:6.1f}  +{(e - s) / 1e3:7.1f}  grid {r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')}x{r.get('Grid_Size_Z','?')} wg {r.get('Workgroup_Size_X','?')}  {r['Kernel_Name'][:80]}")
    prev_e = e
PY
find "$d" -name "*_kernel_trace.csv" -delete 2>/dev/null
