#!/usr/bin/env python
"""Development: the verification instantiation (agg_out) of pna_fused_degree_f32 on a small graph, step by step with syncs."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import _lib
if os.environ.get("PNA_AMD_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["PNA_AMD_LIB"])
from pna_amd import Graph, degree_groups as DG, functional as PF
from pna_amd.dgl.pna_layer import PNASimpleLayer
from pna_amd.synth import powerlaw_graph
dev = torch.device("cuda:0")
V, E, F = 200_000, 2_000_000, int(os.environ.get("FD_F", 75))
src, dst = powerlaw_graph(V, E, seed=V % 89, device=dev)
g = Graph(src, dst, V)
layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.3)}, 0.0, True, True).to(dev).eval()
h = torch.randn(V, (F + 7) // 8 * 8, device=dev)[:, :F]
DG.MIN_ROWS = 1
plan = DG.plan_of(g)
print("plan", plan.G, plan.NV, plan.NR, flush=True)
with torch.no_grad():
    call = PF.FusedDegreeCall(layer, g, h)
    call.group_rows(); torch.cuda.synchronize(); print("plain group_rows ok", flush=True)
    call.rest_rows(); torch.cuda.synchronize(); print("rest_rows ok", flush=True)
    dump = torch.zeros(plan.NV, 4 * F, device=dev)
    c2 = PF.FusedDegreeCall(layer, g, h, agg_out=dump)
    c2.group_rows(); torch.cuda.synchronize(); print("dump group_rows ok", flush=True)
    ref = PF.degree_grouped_aggregate(layer, g, h, plan)[:plan.NV]
    real = plan.perm >= 0
    print("stats identical:", bool(torch.equal(dump[real], ref[real])), flush=True)
