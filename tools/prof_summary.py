#!/usr/bin/env python3
"""Summarise a `rocprofv3 --kernel-trace --stats` run (the *_kernel_stats.csv it writes) into a
small markdown table under profiles/.

    python tools/prof_summary.py gpurun_out/prof_r1a/bench_kernel_stats.csv profiles/r1a_bench.md \
        --steps 18 --title "bench.py --steps 10 --warmup 3 (+5 roofline steps)"
"""
import argparse
import csv


def short(name: str) -> str:
    import re

    if name.startswith("_Z"):
        # llvm-cxxfilt of this ROCm does not know the __bf16 mangling (DF16b): decode by hand
        m = re.search(r"(\d\d)([a-z][A-Za-z_0-9]*?kernel)", name)
        base = m.group(2) if m else name
        targs = []
        if "I" in name[m.end():m.end() + 1] if m else False:
            tail = name[m.end():]
            ty = "bf16" if tail.startswith("IDF16b") else ("f32" if tail.startswith("If") else "?")
            targs = [ty] + re.findall(r"Li(\d+)E", tail.split("EEv")[0] + "E")
        return base + ("<" + ",".join(targs) + ">" if targs else "")
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*", "", name)
    return name[:96]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stats_csv")
    ap.add_argument("out_md")
    ap.add_argument("--steps", type=int, default=1, help="hot-path passes inside the profiled run")
    ap.add_argument("--title", default="")
    ap.add_argument("--top", type=int, default=24)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.stats_csv)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = [f"# rocprofv3 --kernel-trace --stats: {a.title}", "",
             f"source: `{a.stats_csv}`; {a.steps} hot-path passes in the run; "
             f"sum of kernel time {tot/1e6:.3f} ms = {tot/1e6/a.steps:.3f} ms per pass", "",
             "| kernel | calls | calls/pass | avg us | total ms | % |", "|---|---|---|---|---|---|"]
    for r in rows[: a.top]:
        lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {int(r['Calls'])/a.steps:.1f} | "
                     f"{float(r['AverageNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                     f"{float(r['Percentage']):.1f} |")
    open(a.out_md, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
