"""ctypes/numpy front-end of oracle/pna_oracle.c -- TEST INFRASTRUCTURE ONLY.

build() compiles the C restatement with gcc (+OpenMP) into oracle/_build/libpna_oracle.so.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "pna_oracle.c")
LIB = os.path.join(HERE, "_build", "libpna_oracle.so")
AGG_CODES = {"mean": 0, "sum": 1, "max": 2, "min": 3, "std": 4, "var": 5}
_lib = None


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["gcc", "-O3", "-fopenmp", "-ffp-contract=off", "-shared", "-fPIC", SRC, "-o", LIB + ".tmp", "-lm"],
                       check=True)
        os.replace(LIB + ".tmp", LIB)
    return LIB


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.pna_oracle_segreduce.restype = ctypes.c_int
        _lib.pna_oracle_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return lib().pna_oracle_num_threads()


def set_threads(n):
    lib().pna_oracle_set_threads(ctypes.c_int(int(n)))


def degree_scalers(rowptr, avg_log):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    V = rowptr.size - 1
    amp, att = np.empty(V, np.float32), np.empty(V, np.float32)
    lib().pna_oracle_degree_scalers(_p(rowptr), ctypes.c_int32(V), ctypes.c_float(avg_log), _p(amp), _p(att))
    return amp, att


def segreduce(rowptr, col, x, F, aggregators, row_scales=(None,), dst_term=None, edge_term=None, edge_weight=None,
              acc_double=False, col_offset=0, out=None):
    """(V, S*A*F) numpy fp32; x:(rows, >= col_offset+F).  col=None -> x edge-resident.
    `out` may be a preallocated (V, S*A*F) fp32 array (timing runs: keeps page faults out of the clock)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = None if col is None else np.ascontiguousarray(col, dtype=np.int32)
    x, dst_term, edge_term, edge_weight = _f32(x), _f32(dst_term), _f32(edge_term), _f32(edge_weight)
    V = rowptr.size - 1
    A, S = len(aggregators), len(row_scales)
    if out is None:
        out = np.empty((V, A * S * F), np.float32)
    assert out.shape == (V, A * S * F) and out.dtype == np.float32 and out.flags.c_contiguous
    codes = (ctypes.c_int32 * A)(*[AGG_CODES[a] for a in aggregators])
    scales = [_f32(r) for r in row_scales]
    sp = (ctypes.c_void_p * S)(*[None if r is None else r.ctypes.data for r in scales])
    o4 = col_offset * 4

    def ptr(a):
        return None if a is None else ctypes.c_void_p(a.ctypes.data + o4)
    rc = lib().pna_oracle_segreduce(
        ptr(x), ctypes.c_int64(x.shape[1]), _p(rowptr), _p(col), ctypes.c_int32(V), ctypes.c_int32(F),
        ptr(dst_term), ctypes.c_int64(0 if dst_term is None else dst_term.shape[1]),
        ptr(edge_term), ctypes.c_int64(0 if edge_term is None else edge_term.shape[1]),
        _p(edge_weight), ctypes.c_int32(A), codes, ctypes.c_int32(S), sp, _p(out), ctypes.c_int64(out.shape[1]),
        ctypes.c_int32(F), ctypes.c_int32(1 if acc_double else 0))
    if rc != 0:
        raise RuntimeError("pna_oracle_segreduce failed")
    return out
