#!/usr/bin/env python
"""Driver for rocprofv3 passes over the degree-grouped simple layer on the C3 graph (gather in plan order + the two grouped
contractions)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph  # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
V, E, F = 1_000_000, 10_000_000, 75
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True).to(dev).eval()
h = torch.randn(V, 80, device=dev)[:, :F]
with torch.no_grad():
    assert layer._degree_grouped_path(g, h)
    for _ in range(n):
        layer(g, h)
torch.cuda.synchronize()
print("ok")
