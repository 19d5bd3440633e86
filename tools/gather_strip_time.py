#!/usr/bin/env python
"""Times the strip-wise A-layout gather prototype (tools/ubench/gather_strip.hip, DESIGN.md 4.7 point 7) on the C3 graph in
plan order against the production row-wise gather, and checks its aggregate against the production one.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/gather_strip.hip -o tools/ubench/libgather_strip.so
    python tools/gather_strip_time.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, functional as PF, degree_groups as DG, ops  # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

L = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libgather_strip.so"))
L.gather_strip.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                           ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
L.gather_strip_all.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                               ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
V, E, F = 1_000_000, 10_000_000, 75
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True).to(dev).eval()
h = torch.randn(V, 80, device=dev)[:, :F]
plan = DG.plan_of(g)
csr = g.csr
ntiles = plan.NV // 16
out = torch.zeros(ntiles * 64, device=dev)
st = torch.cuda.current_stream().cuda_stream


def ev(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best


def strip(unroll, lds, agg=None):
    rc = L.gather_strip(csr.rowptr.data_ptr(), csr.col.data_ptr(), h.data_ptr(), h.stride(0), F, plan.perm.data_ptr(), ntiles, out.data_ptr(),
                        agg.data_ptr() if agg is not None else None, agg.stride(0) if agg is not None else 0, unroll, lds, st)
    assert rc == 0, rc


with torch.no_grad():
    agg_ref = PF.degree_grouped_aggregate(layer, g, h, plan)
    t_ref = ev(lambda: PF.degree_grouped_aggregate(layer, g, h, plan, out=agg_ref))
    # correctness of the prototype: its aggregate rows against the production ones (group rows only; max / min bit-exact)
    agg = torch.zeros(plan.NV, 320, device=dev)[:, :300]
    strip(8, 0, agg)
    torch.cuda.synchronize()
    real = plan.perm >= 0
    a, b = agg[real], agg_ref[:plan.NV][real]
    assert torch.equal(a[:, F:3 * F], b[:, F:3 * F]), "max / min differ"
    err = ((a - b).abs() / (1 + b.abs())).max().item()
    print(f"prototype aggregate vs production: max / min bit-exact, mean / std max rel err {err:.2e}", flush=True)
    edges = int((csr.rowptr[1:] - csr.rowptr[:-1])[plan.perm[real].long()].sum())
    print(f"production gather (all rows, writes the aggregate): {t_ref:.3f} ms; group rows hold {edges} of {E} edges", flush=True)
    def strip_all(unroll, lds):
        rc = L.gather_strip_all(csr.rowptr.data_ptr(), csr.col.data_ptr(), h.data_ptr(), h.stride(0), F, plan.perm.data_ptr(), ntiles,
                                out.data_ptr(), unroll, lds, st)
        assert rc == 0, rc
    strip(8, 0)
    torch.cuda.synchronize()
    c1 = out.clone()
    strip_all(4, 0)
    torch.cuda.synchronize()
    print(f"one-pass variant checksum vs three-pass: max rel diff {((out - c1).abs() / (1 + c1.abs())).max().item():.2e}", flush=True)
    for unroll in (2, 4):
        for lds, tag in ((0, "full occupancy"), (64 * 1024, "<= 2 workgroups = 8 wavefronts per CU")):
            print(f"one-pass strip gather (96 statistics per lane) U={unroll} {tag}: {ev(lambda: strip_all(unroll, lds)):.3f} ms without output", flush=True)
    for unroll in (8,):
        for lds, tag in ((0, "full occupancy"), (64 * 1024, "<= 2 workgroups = 8 wavefronts per CU")):
            t = ev(lambda: strip(unroll, lds))
            print(f"three-pass strip gather U={unroll} {tag}: {t:.3f} ms without output", flush=True)
