#!/usr/bin/env python
"""Degree-grouped vs ordinary simple layer over output widths / scaler sets (1 M nodes, 10 M edges): is the grouping a gain
everywhere it is enabled?  (development tool)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, degree_groups as DG  # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

dev = torch.device("cuda:0")
V, E = 1_000_000, 10_000_000
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}


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


for F, scalers in [(20, "identity amplification attenuation"), (32, "identity amplification attenuation"), (50, "identity amplification attenuation"),
                   (64, "identity amplification attenuation"), (75, "identity amplification"), (75, "identity amplification attenuation"),
                   (96, "identity amplification attenuation"), (128, "identity amplification")]:
    layer = PNASimpleLayer(F, F, "mean max min std", scalers, avg, 0.0, True, True).to(dev).eval()
    h = torch.randn(V, (F + 3) // 4 * 4, device=dev)[:, :F]
    with torch.no_grad():
        DG.ENABLED = True
        on = layer._degree_grouped_path(g, h)
        t_g = ev(lambda: layer(g, h))
        DG.ENABLED = False
        t_p = ev(lambda: layer(g, h))
        DG.ENABLED = True
    print(f"F={F} scalers={len(scalers.split())}: grouped path {'on' if on else 'off'} {t_g:.3f} ms, ordinary {t_p:.3f} ms", flush=True)
