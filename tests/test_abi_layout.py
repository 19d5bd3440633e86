"""The C structs of include/pna_amd.h and their ctypes mirrors in pna_amd/_lib.py must agree field by field: name, offset, size.
A mismatch would not fail loudly -- the kernels would read another field's bytes -- so the header is compiled (gcc, host side only)
into a program that prints sizeof / offsetof of every field, and the output is compared with ctypes."""
import ctypes
import os
import re
import shutil
import subprocess

import pytest

from pna_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pna_amd.h")


def _structs():
    """{struct name: [field names]} parsed from the header (fields end in ';' inside a typedef struct block)."""
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        assert m.group(1) == m.group(3)
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if decl:
                fields.append(re.match(r".*?(\w+)\s*(\[[^\]]*\])?$", decl, flags=re.S).group(1))
        out[m.group(1)] = fields
    return out


def _ctypes_class(struct):
    return getattr(_lib, "".join(p.capitalize() for p in struct.split("_")))


def test_every_struct_of_the_header_has_a_ctypes_mirror_with_the_same_layout(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    structs = _structs()
    assert len(structs) >= 10 and "pna_fused_degree_args" in structs and "pna_bn_tail_args" in structs
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "pna_amd.h"', "int main(void) {"]
    for s, fields in structs.items():
        lines.append(f'  printf("{s} . %zu 0\\n", sizeof({s}));')
        for f in fields:
            lines.append(f'  printf("{s} {f} %zu %zu\\n", sizeof((({s}*)0)->{f}), offsetof({s}, {f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-std=c11", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True, capture_output=True)
    rows = [l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()]
    c_layout = {}
    for s, f, size, off in rows:
        c_layout.setdefault(s, {})[f] = (int(size), int(off))
    for s, fields in structs.items():
        cls = _ctypes_class(s)
        assert ctypes.sizeof(cls) == c_layout[s]["."][0], (s, ctypes.sizeof(cls), c_layout[s]["."][0])
        names = [n for n, *_ in cls._fields_]
        i = 0
        for f in fields:                                     # in declaration order; a C array may be mirrored element by element
            size, off = c_layout[s][f]
            d = getattr(cls, names[i])
            if d.size == size:
                assert names[i] == f and d.offset == off, (s, f, names[i], (d.size, d.offset), (size, off))
                i += 1
                continue
            covered = 0
            while covered < size:
                d = getattr(cls, names[i])
                assert d.offset == off + covered, (s, f, names[i], d.offset, off + covered)
                covered += d.size
                i += 1
            assert covered == size, (s, f, covered, size)
        assert i == len(names), (s, names[i:])
