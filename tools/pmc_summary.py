#!/usr/bin/env python3
"""Per-kernel averages of a `rocprofv3 --pmc ... --kernel-trace --output-format csv` run.

    python tools/pmc_summary.py <dir with *counter_collection.csv> [--match block_kernel relpos] > out.json

Sums every counter per dispatch, averages over the dispatches of a kernel name (template arguments kept, argument
lists dropped), and adds the usual SQ fractions (share of SQ_WAVE_CYCLES spent waiting / issue-stalled / active).
"""
import argparse
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--match", nargs="*", default=["block_kernel", "relpos_attn"])
    ap.add_argument("--source", default="")
    a = ap.parse_args()
    files = glob.glob(f"{a.dir}/**/*counter_collection.csv", recursive=True)
    if not files:
        sys.exit(f"no *counter_collection.csv under {a.dir}")
    per = defaultdict(lambda: defaultdict(float))  # kernel -> counter -> sum
    disp = defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if a.match and not any(m in k for m in a.match):
                continue
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    out = {"source": a.source, "kernels": {}}
    for k, c in sorted(per.items()):
        n = len(disp[k])
        d = {name: round(v / n) for name, v in sorted(c.items())}
        d["launches"] = n
        wc = d.get("SQ_WAVE_CYCLES")
        if wc:
            for name, key in (("SQ_WAIT_ANY", "frac_wait_any"), ("SQ_WAIT_INST_ANY", "frac_wait_inst_any"),
                              ("SQ_ACTIVE_INST_ANY", "frac_active_inst_any"), ("SQ_ACTIVE_INST_LDS", "frac_active_inst_lds")):
                if name in d:
                    d[key] = round(d[name] / wc, 3)
        out["kernels"][k] = d
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
