#!/usr/bin/env python
"""Driver for the rocprofv3 passes (kernel trace and PMC): the hot kernels on the roofline workload, plus a
calibration launch of the same gather kernel whose true fabric read volume is known.

Calibration graph: every source row is read exactly once (col = a permutation), rows are 384-byte aligned
(pitch 96 floats = exactly three 128-byte lines), so the bytes that must cross the L2 -> fabric boundary are
E * 384 (+ 4 B/edge ids + rowptr) regardless of caching.  FETCH_SIZE of that launch gives the byte scale of
the counter for THIS access pattern (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide reads on gfx950).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, functional as PF  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
F = 75
aggs = ["mean", "max", "min", "std"]

# ---- calibration: Vc destinations x 8 in-edges, each of the 8*Vc source rows read exactly once
Vc = 500_000
Ec = 8 * Vc
perm = torch.randperm(Ec, generator=torch.Generator().manual_seed(0)).to(dev)
gc = Graph(perm, torch.arange(Ec, device=dev) // 8, Vc)
gc.num_src = Ec
xc = torch.zeros(Ec, 96, device=dev)[:, :F]
xc.normal_()
with torch.no_grad():
    for _ in range(n):
        PF.aggregate(gc, xc, F, aggs)
torch.cuda.synchronize()
del xc, gc, perm

# ---- the roofline workload
V, E = 1_000_000, 10_000_000
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
amp, att = g.degree_scalers(2.2488)
xb = torch.zeros(V, 80, device=dev)
x = xb[:, :F]
x.copy_(torch.randn(V, F, generator=torch.Generator().manual_seed(1234)))
W = (torch.randn(F, 12 * F, generator=torch.Generator().manual_seed(1)) / 30).to(dev)
b = torch.zeros(F, device=dev)
with torch.no_grad():
    for _ in range(n):
        agg = PF.aggregate(g, x, F, aggs)
    from pna_amd import ops
    for arith in ("bf16x3", "f32"):           # both contraction kernels: the bf16x3 default and the exact f32-MFMA one
        for _ in range(n):
            y = ops.posttrans(agg, 4 * F, W, [None, amp, att], b, arith=arith)
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
