"""Minimal stand-in for `humanfriendly` (size/time parsing used by the reference's
frontend ctor and model summary).  Test infrastructure only; no arithmetic on the hot path."""
import re

_UNITS = {"": 1, "b": 1, "k": 1000, "kb": 1000, "m": 10**6, "mb": 10**6, "g": 10**9, "gb": 10**9,
          "kib": 1024, "mib": 1024**2, "gib": 1024**3}


def parse_size(size, binary=False):
    if isinstance(size, (int, float)):
        return int(size)
    m = re.fullmatch(r"\s*([0-9.]+)\s*([a-zA-Z]*)\s*", str(size))
    if not m:
        raise ValueError(size)
    return int(float(m.group(1)) * _UNITS[m.group(2).lower()])


def format_size(n, binary=False):
    return f"{n} bytes"


def format_timespan(s):
    return f"{s:.2f} seconds"
