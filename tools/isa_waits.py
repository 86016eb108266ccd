#!/usr/bin/env python3
"""Static check of a hipcc -S listing for the pattern that cost the block kernels their prologues (round 3): global
loads that hipcc follows with `s_waitcnt vmcnt(0)` on the spot (a load under a uniform branch, a load -> store copy
loop, a returning atomic), i.e. dependent global round trips in front of the real work.

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o x.s csrc/x.hip && python tools/isa_waits.py x.s [substr]

Per kernel: global loads, `vmcnt(0)` waits, barriers, and the timeline of memory operations as run lengths."""
import re
import sys


def kernels(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s*;\s*@", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is not None:
            out[cur].append(line.rstrip("\n"))
            if "s_endpgm" in line:
                cur = None
    return out


def timeline(body):
    ev = []
    for i, x in enumerate(body):
        t = x.split()
        if not t:
            continue
        op = t[0]
        if op == "s_waitcnt":
            op = " ".join(t[:2]) if len(t) > 1 else op
        elif not re.match(r"(global_|buffer_|ds_|s_barrier|v_mfma|s_load|flat_)", op):
            continue
        if ev and ev[-1][1] == op:
            ev[-1][2] += 1
        else:
            ev.append([i, op, 1])
    return ev


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else None
    for name, body in kernels(path).items():
        nl = sum(1 for x in body if re.search(r"\b(global|buffer)_load", x))
        if nl == 0:
            continue
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:70]
        n0 = sum(1 for x in body if re.search(r"s_waitcnt vmcnt\(0\)", x))
        print(f"{short:72s} loads {nl:4d}  vmcnt(0) {n0:3d}  barriers {sum('s_barrier' in x for x in body):3d}  lines {len(body)}")
        if sub and sub in name:
            for i, op, n in timeline(body):
                print(f"    {i:5d} {op} x{n}")


if __name__ == "__main__":
    main()
