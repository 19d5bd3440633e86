"""INTEGRATION.md shows the reference-side ctypes binding.  Round 3's hand-written struct had fallen five fields behind the header
(a maintainer copying it would have handed the library a struct 40 bytes short).  Since round 4 the block is generated from
include/pna_amd.h (tools/gen_ctypes_stub.py); here: the block in INTEGRATION.md IS the generator's output, its layouts are gcc's, and
the library refuses an args struct that is shorter than its own."""
import ctypes
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _block():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- BEGIN GENERATED ctypes mirror.*?-->\n```python\n(.*?)```\n<!-- END GENERATED ctypes mirror -->", text, flags=re.S)
    assert m, "INTEGRATION.md lost its generated block"
    return m.group(1)


def test_the_stub_in_integration_md_is_the_generators_output():
    import gen_ctypes_stub
    body = _block()
    assert body.split("\n", 1)[0].startswith("# models/dgl/_pna_amd_structs.py")
    assert body.split("\n", 1)[1] == gen_ctypes_stub.generate(), "INTEGRATION.md is stale: paste the output of tools/gen_ctypes_stub.py"


def test_the_stub_has_gccs_layout(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    ns = {}
    exec(_block(), ns)                                                     # what a maintainer would paste
    classes = {k: v for k, v in ns.items() if isinstance(v, type) and issubclass(v, ctypes.Structure) and k.startswith("Pna")}
    assert len(classes) >= 11
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "pna_amd.h"', "int main(void) {"]
    for name, cls in classes.items():
        struct = re.sub(r"(?<!^)(?=[A-Z])", "_", name).lower()
        lines.append(f'  printf("{name} . %zu 0\\n", sizeof({struct}));')
        for f, *_ in cls._fields_:
            lines.append(f'  printf("{name} {f} %zu %zu\\n", sizeof((({struct}*)0)->{f}), offsetof({struct}, {f}));')
    lines += ["  return 0;", "}"]
    src, exe = tmp_path / "l.c", tmp_path / "l"
    src.write_text("\n".join(lines))
    subprocess.run([gcc, "-std=c11", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True, capture_output=True)
    for name, f, size, off in (l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()):
        cls = classes[name]
        if f == ".":
            assert ctypes.sizeof(cls) == int(size), (name, ctypes.sizeof(cls), size)
        else:
            fld = getattr(cls, f)
            assert (fld.size, fld.offset) == (int(size), int(off)), (name, f, fld.size, fld.offset, size, off)
        if name != "PnaTuning":
            assert cls._fields_[0][0] == "struct_size" and cls().struct_size == ctypes.sizeof(cls)


def test_a_short_args_struct_is_refused():
    """ABI 19 on: struct_size first in every args struct; an entry point checks it before it looks at anything else (no GPU needed)."""
    from pna_amd import _lib
    L = _lib.lib()
    L.pna_last_error.restype = ctypes.c_char_p
    cases = [(_lib.PnaSegreduceArgs, L.pna_segreduce_fwd_f32), (_lib.PnaPosttransArgs, L.pna_posttrans_f32), (_lib.PnaPosttransArgs, L.pna_posttrans_x3_f32),
             (_lib.PnaFusedDegreeArgs, L.pna_fused_degree_f32), (_lib.PnaFusedSimpleArgs, L.pna_fused_simple_f32),
             (_lib.PnaSmallLinearArgs, L.pna_small_linear_f32), (_lib.PnaTowerLayerArgs, L.pna_tower_layer_f32), (_lib.PnaBnTailArgs, L.pna_bn_tail_fwd_f32),
             (_lib.PnaSegreduceBwdArgs, L.pna_segreduce_bwd_f32), (_lib.PnaPosttransDwArgs, L.pna_posttrans_dw_f32),
             (_lib.PnaPosttransDwGroupedArgs, L.pna_posttrans_dw_grouped_f32)]
    for cls, fn in cases:
        a = cls()
        assert a.struct_size == ctypes.sizeof(cls)
        for short in (0, ctypes.sizeof(cls) - 8):
            a.struct_size = short
            rc = fn(ctypes.byref(a), None)
            assert rc == -1, (cls.__name__, short, rc)
            assert b"struct_size" in L.pna_last_error(), L.pna_last_error()
