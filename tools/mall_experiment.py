#!/usr/bin/env python
"""Would the aggregate's round trip disappear if it stayed in the 256 MB Infinity Cache?  (development experiment)

The degree-grouped layer writes a 1.28 GB aggregate and reads it back.  If gather and contraction were interleaved in chunks
of ~64 k rows over a REUSED 84 MB buffer, the aggregate could live in the memory-side cache.  Upper bound of that, measured
without building it: (a) the gather with every output row wrapped into a 65 536-row window (same reads, same number of
written bytes, but the written lines are overwritten in place), (b) the contraction with its A rows wrapped into the same
window (library built with -DX3_DEV_A_WRAP_ROWS=65536; PNA_AMD_LIB selects it).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import _lib  # noqa: E402
if os.environ.get("PNA_AMD_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["PNA_AMD_LIB"])
from pna_amd import Graph, functional as PF, degree_groups as DG  # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

dev = torch.device("cuda:0")
V, E, F = 1_000_000, 10_000_000, 75
WRAP = 65536
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True).to(dev).eval()
h = torch.randn(V, 80, device=dev)[:, :F]
plan = DG.plan_of(g)


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


tag = os.environ.get("PNA_AMD_LIB", "shipped library")
if os.environ.get("NT_STORE"):          # NT_STORE=-1: ordinary (cache-allocating) output stores instead of the non-temporal default
    from pna_amd import ops
    ops.set_tuning(nt_store=int(os.environ["NT_STORE"]))
    tag += f", nt_store={os.environ['NT_STORE']}"
with torch.no_grad():
    agg = PF.degree_grouped_aggregate(layer, g, h, plan)
    y = torch.empty(V, F, device=dev)
    t_g = ev(lambda: PF.degree_grouped_aggregate(layer, g, h, plan, out=agg))
    t_c = ev(lambda: PF.degree_grouped_posttrans(layer, g, h, agg, plan, out=y))
    # (a) the gather writing into a 65 536-row window
    items, hout = plan.items, plan.heavy_out
    n_seg = plan._n_seg
    wi = items.clone()
    wi[n_seg:, 0] = wi[n_seg:, 0] % WRAP
    plan.items = wi.contiguous()
    plan.heavy_out = (hout % WRAP).contiguous() if hout is not None else None
    t_gw = ev(lambda: PF.degree_grouped_aggregate(layer, g, h, plan, out=agg))
    plan.items, plan.heavy_out = items, hout
print(f"[{tag}] gather {t_g:.3f} ms, gather with output rows wrapped into {WRAP} rows {t_gw:.3f} ms, contraction (grouped + rest) {t_c:.3f} ms", flush=True)
