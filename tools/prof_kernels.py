#!/usr/bin/env python
"""Minimal driver for rocprofv3 passes: the two hot kernels on the roofline workload, a few launches each."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, functional as PF  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

dev = torch.device("cuda:0")
V, E, F = 1_000_000, 10_000_000, 75
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
amp, att = g.degree_scalers(2.2488)
x = torch.randn(V, F, generator=torch.Generator().manual_seed(1234)).to(dev)
W = (torch.randn(F, 12 * F, generator=torch.Generator().manual_seed(1)) / 30).to(dev)
b = torch.zeros(F, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
with torch.no_grad():
    for _ in range(n):
        agg = PF.aggregate(g, x, F, ["mean", "max", "min", "std"])
    for _ in range(n):
        y = PF.posttrans(agg, 4 * F, W, b, [None, amp, att])
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
