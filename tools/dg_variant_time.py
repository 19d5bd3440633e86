#!/usr/bin/env python
"""Times the two halves of the degree-grouped layer on the C3 graph with whatever library PNA_AMD_LIB names (development)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import _lib
if os.environ.get("PNA_AMD_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["PNA_AMD_LIB"])
from pna_amd import Graph, functional as PF, degree_groups as DG
from pna_amd.dgl.pna_layer import PNASimpleLayer
from pna_amd.synth import powerlaw_graph
dev = torch.device("cuda:0")
V, E, F = 1_000_000, 10_000_000, 75
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True).to(dev).eval()
hb = torch.randn(V, 80, device=dev); h = hb[:, :F]
plan = DG.plan_of(g)


def ev(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best


for align in [int(v) for v in os.environ.get("DG_ALIGN", "1,32").split(",")] * int(os.environ.get("DG_REPEAT", "2")):
  with torch.no_grad():
    DG.AGG_ALIGN = align
    agg = PF.degree_grouped_aggregate(layer, g, h, plan)
    y = torch.empty(V, F, device=dev)
    print(sys.argv[1] if len(sys.argv) > 1 else "", f"pitch {agg.stride(0)}: gather {ev(lambda: PF.degree_grouped_aggregate(layer, g, h, plan, out=agg)):.3f} ms",
          f"contraction (grouped + rest) {ev(lambda: PF.degree_grouped_posttrans(layer, g, h, agg, plan, out=y)):.3f} ms",
          f"layer {ev(lambda: layer(g, h)):.3f} ms", flush=True)
