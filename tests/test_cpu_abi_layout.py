"""The ctypes mirrors of the C-ABI structs (espnet_amd/lib.py) against include/espnet_amd.h as a C compiler lays them out:
size of every struct, offset of every field, by name.  A field added to the header and not to the mirror (or the other way
round, or in another order) makes the library read the caller's memory at the wrong place - silently; this test is the check
that does not need a GPU.  gcc compiles a program that prints sizeof / offsetof for exactly the fields the mirrors declare."""
import ctypes as C
import inspect
import shutil
import subprocess
from pathlib import Path

import pytest

from espnet_amd import lib as L

REPO = Path(__file__).resolve().parent.parent


def _mirrors():
    return [(n, c) for n, c in inspect.getmembers(L, inspect.isclass)
            if issubclass(c, C.Structure) and c is not C.Structure and n.startswith("Em")]


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs a C compiler")
def test_ctypes_mirrors_match_the_header(tmp_path):
    structs = _mirrors()
    assert len(structs) >= 16  # every `typedef struct Em...{` of the header that crosses the boundary by value or pointer
    src = ["#include <stdio.h>", "#include <stddef.h>", '#include "espnet_amd.h"', "int main(void) {"]
    for n, c in structs:
        src.append(f'  printf("{n} . %zu\\n", sizeof({n}));')
        for f in c._fields_:
            src.append(f'  printf("{n} {f[0]} %zu\\n", offsetof({n}, {f[0]}));')
    src += ["  return 0;", "}"]
    (tmp_path / "layout.c").write_text("\n".join(src))
    r = subprocess.run(["gcc", "-std=c11", "-I", str(REPO / "include"), str(tmp_path / "layout.c"), "-o",
                        str(tmp_path / "layout")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]  # (a field name the header does not have fails here)
    out = subprocess.run([str(tmp_path / "layout")], capture_output=True, text=True, check=True).stdout
    want = {}
    for line in out.splitlines():
        s, f, v = line.split()
        want[(s, f)] = int(v)
    bad = []
    for n, c in structs:
        if want[(n, ".")] != C.sizeof(c):
            bad.append((n, "sizeof", want[(n, ".")], C.sizeof(c)))
        for f in c._fields_:
            if want[(n, f[0])] != getattr(c, f[0]).offset:
                bad.append((n, f[0], want[(n, f[0])], getattr(c, f[0]).offset))
    assert not bad, bad


def test_header_structs_all_have_a_mirror():
    """Every struct the header defines with a body is mirrored (EmProfile is opaque: created and read through functions)."""
    import re

    text = (REPO / "include" / "espnet_amd.h").read_text()
    declared = set(re.findall(r"typedef struct (Em\w+) \{", text))
    mirrored = {n for n, _ in _mirrors()}
    assert declared == mirrored, (declared - mirrored, mirrored - declared)
