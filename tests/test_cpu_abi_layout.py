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


def test_bound_signatures_match_the_header_prototypes():
    """Every prototype of the header against the ctypes signature the host layer binds it with (espnet_amd.lib._SIGNATURES):
    same number of parameters, each of the same kind (pointer / 32-bit int / float / double / 64-bit size), same kind of
    result.  ctypes would pass a float where the library reads an int without a word."""
    import re

    text = (REPO / "include" / "espnet_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = re.findall(r"([A-Za-z_][\w \*]*?)\b(em_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text)
    scalar = {"float": "f32", "double": "f64", "int32_t": "i32", "int": "i32", "size_t": "u64", "int64_t": "i64",
              "long long": "i64", "uint32_t": "u32", "unsigned": "u32", "uint64_t": "u64", "void": None}

    def kind_c(p):
        p = p.strip()
        if p in ("void", ""):
            return None
        if "*" in p or "[" in p:
            return "ptr"
        t = re.sub(r"\b(const|volatile)\b", "", p).split()
        return scalar[" ".join(t[:-1]) if len(t) > 1 else t[0]]

    def kind_ct(a):
        if a is None:
            return None
        if a in (C.c_void_p, C.c_char_p) or (hasattr(a, "_type_") and not isinstance(a._type_, str)):
            return "ptr"
        return {C.c_int32: "i32", C.c_int: "i32", C.c_float: "f32", C.c_double: "f64", C.c_size_t: "u64",
                C.c_int64: "i64", C.c_uint32: "u32", C.c_uint64: "u64"}[a]

    assert {n for _, n, _ in protos} == set(L._SIGNATURES)
    bad = []
    for ret, name, args in protos:
        res, cargs = L._SIGNATURES[name]
        want = [k for k in (kind_c(a) for a in args.split(",")) if k]
        got = [kind_ct(a) for a in cargs]
        if want != got:
            bad.append((name, want, got))
        r = ret.strip()
        if ("ptr" if "*" in r else scalar[r]) != kind_ct(res):
            bad.append((name, "result", r, res))
    assert not bad, bad


def test_fragment_major_packing_and_flag_macros(tmp_path):
    """Round 6: `pack_frag16` is the layout include/espnet_amd.h states for em_dec_ffn / em_ln_gemm_frag -
    [R/16][K/32][lane = 16 * (k % 32 / 8) + r % 16][k % 8], rows zero-padded to `pad_rows` - and the Python mirrors of the
    round's macros (EM_ENC_IN_FLIGHT, EM_LNF_*) are the header's values as gcc evaluates them."""
    import subprocess

    import torch

    from espnet_amd import lib as L

    R, K = 40, 96
    w = torch.arange(R * K, dtype=torch.float32).reshape(R, K)
    f = L.pack_frag16(w, pad_rows=32).reshape(-1)
    Rp = 64
    assert f.numel() == Rp * K
    for r, k in ((0, 0), (5, 7), (15, 31), (16, 32), (39, 95), (23, 40)):
        lane = 16 * ((k % 32) // 8) + r % 16
        idx = (((r // 16) * (K // 32) + k // 32) * 64 + lane) * 8 + k % 8
        assert f[idx].item() == w[r, k].item(), (r, k)
    back = f.reshape(Rp // 16, K // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(Rp, K)
    assert torch.equal(back[:R], w) and (back[R:] == 0).all()
    src = tmp_path / "m.c"
    src.write_text('#include <stdio.h>\n#include "espnet_amd.h"\nint main(void) { printf("%d %d %d %d %d\\n", EM_ENC_IN_FLIGHT(3), '
                   'EM_ENC_IN_FLIGHT(17), EM_LNF_RELU_FRAG, EM_LNF_STORE, EM_LNF_STORE_F32); return 0; }\n')
    exe = tmp_path / "m"
    inc = str(__import__("pathlib").Path(L.__file__).resolve().parents[1] / "include")
    subprocess.run(["gcc", "-I", inc, str(src), "-o", str(exe)], check=True)
    vals = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert vals == [L.EM_ENC_IN_FLIGHT(3), L.EM_ENC_IN_FLIGHT(17), L.EM_LNF_RELU_FRAG, L.EM_LNF_STORE, L.EM_LNF_STORE_F32]
    for bit in (L.EM_ENC_ISOLATE_UTTS, L.EM_ENC_NO_FUSED, L.EM_ENC_POS_PROJECTED, L.EM_ENC_POS_PACKED):
        assert L.EM_ENC_IN_FLIGHT(15) & bit == 0  # (the count's bits are its own)
