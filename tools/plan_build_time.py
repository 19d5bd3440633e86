#!/usr/bin/env python
"""Where the per-graph set-up of the one-kernel layer goes: each stage of the degree plan on the C3 graph, host wall clock
with the device synchronised around it, on a FIRST graph (torch's kernels load on first use) and on a second, fresh Graph object."""
import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, degree_groups as DG, functional as PF   # noqa: E402
from pna_amd.synth import powerlaw_graph                           # noqa: E402
dev = torch.device("cuda:0")
V, E = 1_000_000, 10_000_000
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)


def clock(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, (time.perf_counter() - t) * 1e3


out = []
for rnd in range(3):
    g = Graph(src, dst, V)
    rec = {}
    _, rec["csr"] = clock(lambda: g.csr)
    _, rec["heavy_schedule"] = clock(lambda: g.heavy_schedule())
    _, rec["work_items"] = clock(lambda: g.work_items())
    plan, rec["DegreePlan.__init__"] = clock(lambda: DG.plan_of(g))
    _, rec["fused_tables"] = clock(lambda: plan.fused_tables())
    _, rec["rest_items"] = clock(lambda: plan.rest_items(g) if plan.NR else None)
    for sp in (0, DG.FUSED_SPARE_WGS):
        _, rec[f"fused_balance(spare={sp})"] = clock(lambda: plan.fused_balance(PF._fused_grid(dev, sp, plan.NV // 64)))
    rec["total_plan"] = sum(v for k, v in rec.items() if k != "csr")
    out.append(rec)
    print(rnd, json.dumps({k: round(v, 2) for k, v in rec.items()}), flush=True)
