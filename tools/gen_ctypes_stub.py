#!/usr/bin/env python
"""ctypes mirrors of every struct of include/pna_amd.h, GENERATED from the header: the block INTEGRATION.md shows a maintainer of the
reference (tests/test_integration_stub.py checks that the block in INTEGRATION.md is this program's output and that its layouts are
gcc's).  Round 3's hand-written stub had fallen five fields behind the header.

    python tools/gen_ctypes_stub.py            # prints the block
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pna_amd.h")
SCALARS = {"int32_t": "c_int32", "uint32_t": "c_uint32", "int64_t": "c_int64", "uint64_t": "c_uint64", "float": "c_float", "int": "c_int"}


def parse():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(PNA_MAX_\w+)\s+(\d+)", text)}
    out = []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"(.*?)(\w+)\s*(?:\[(\w+)\])?$", decl)
            ctype, name, dim = fm.group(1).strip(), fm.group(2), fm.group(3)
            if "*" in ctype:
                py = "ctypes.c_void_p"
            elif ctype.replace("const ", "") in SCALARS:
                py = "ctypes." + SCALARS[ctype.replace("const ", "")]
            else:
                py = cls_name(ctype.replace("const ", "").strip())          # a nested struct (pna_tuning)
            if dim:
                py = f"{py} * {consts.get(dim, dim)}"
            fields.append((name, py))
        out.append((m.group(1), fields))
    return out


def cls_name(struct):
    return "".join(p.capitalize() for p in struct.split("_"))


def generate():
    lines = ["import ctypes", "", "", "class _PnaArgs(ctypes.Structure):",
             "    def __init__(self, *a, **k):                      # struct_size: sizeof of THIS mirror (include/pna_amd.h, PNA_ARGS_INIT)",
             "        super().__init__(*a, **k)",
             "        if hasattr(self, \"struct_size\"):",
             "            self.struct_size = ctypes.sizeof(self)", ""]
    for struct, fields in parse():
        lines += ["", f"class {cls_name(struct)}(_PnaArgs):          # struct {struct}", "    _fields_ = ["]
        row = "        "
        for name, py in fields:
            item = f"(\"{name}\", {py}), "
            if len(row) + len(item) > 128:
                lines.append(row.rstrip())
                row = "        "
            row += item
        lines += [row.rstrip(), "    ]", ""]
    return "\n".join(lines).rstrip() + "\n"


if __name__ == "__main__":
    print(generate(), end="")
