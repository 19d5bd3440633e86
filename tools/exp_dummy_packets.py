#!/usr/bin/env python
"""What do the one-kernel layer's padding records cost?  A tile of in-degree D owns round_up(D, 4) edge records, the ones past D
repeating edge D - 1 (16 % of all packets on the benchmark graph).  Three graphs through the SAME kernel:
  A  the benchmark graph (9.47 M real edges in the degree groups, 10.97 M packets);
  B  A with every in-degree rounded UP to a multiple of 4 by extra random edges (every packet real, as many packets as A);
  C  A with every in-degree >= 4 rounded DOWN to a multiple of 4 (every packet real, fewer packets).
T_B - T_A = what a real packet costs over a padding packet; the line through C and B at A's number of REAL edges = the time a
kernel without padding records would take on A.

    python tools/exp_dummy_packets.py [json-out]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, degree_groups as DG, functional as PF  # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

dev = torch.device("cuda:0")
V, E, F = 1_000_000, 10_000_000, 75


def ev(fn, n=20, reps=5):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best


src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
order = torch.argsort(dst, stable=True)
src, dst = src[order], dst[order]
deg = torch.bincount(dst, minlength=V)
rowptr = torch.cumsum(deg, 0) - deg
rank = torch.arange(E, device=dev) - rowptr[dst]
# B: extra random in-edges up to the next multiple of 4
extra = (-deg) % 4
dst_x = torch.repeat_interleave(torch.arange(V, device=dev), extra)
gen = torch.Generator(device=dev).manual_seed(7)
src_x = torch.randint(0, V, (dst_x.numel(),), device=dev, generator=gen)
# C: drop the last D & 3 in-edges of rows with D >= 4
keep = (deg[dst] < 4) | (rank < (deg[dst] // 4 * 4))
graphs = {"A": (src, dst), "B": (torch.cat([src, src_x]), torch.cat([dst, dst_x])), "C": (src[keep], dst[keep])}

torch.manual_seed(0)
out = {}
h = torch.randn(V, (F + 7) // 8 * 8, device=dev)[:, :F]
calls = {}
for name, (s_, d_) in graphs.items():
    g = Graph(s_, d_, V)
    avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
    layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True).to(dev).eval()
    with torch.no_grad():
        assert DG.fused_applies(g, h, F, F)
        plan = DG.plan_of(g)
        call = PF.FusedDegreeCall(layer, g, h)
        d_t = plan.fused_tables()[0][:, 1].long()
        real = int(d_t.sum().item()) * 16
        calls[name] = call
        out[name] = {"edges": int(s_.numel()), "group_rows": plan.NV, "packets_x16": plan.fused_tables()[2] * 16, "real_edge_slots_x16": real}
with torch.no_grad():
    for rep in range(3):
        for name, call in calls.items():
            t = ev(call.group_rows)
            out[name].setdefault("ms", []).append(t)
            print(f"rep {rep} graph {name}: {out[name]['packets_x16'] / 1e6:.2f} M packet slots, {out[name]['real_edge_slots_x16'] / 1e6:.2f} M real: {t:.4f} ms", flush=True)
tA, tB, tC = (min(out[k]["ms"]) for k in "ABC")
pA, pB, pC = (out[k]["real_edge_slots_x16"] for k in "ABC")
est = tC + (tB - tC) * (pA - pC) / max(pB - pC, 1)
out["estimate_without_padding_ms"] = est
print(f"A {tA:.4f} ms, B {tB:.4f} ms, C {tC:.4f} ms; a kernel without padding records on A: about {est:.4f} ms ({(1 - est / tA) * 100:.1f} % faster)")
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
