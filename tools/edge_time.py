#!/usr/bin/env python
"""C3-scale gather of the tower layers with edge features: no edge term / type table (<= 4 types, ABI 14) /
per-edge term, hand-scheduled kernel; and the compiler-scheduled kernel with the per-edge term (what these calls ran on before)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, ops
from pna_amd.synth import powerlaw_graph
dev = torch.device("cuda:0")
V, E, F = 1_000_000, 10_000_000, 75
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
c = g.csr
x, dt = torch.randn(V, F, device=dev), torch.randn(V, F, device=dev)
table = torch.randn(4, F, device=dev)
types = torch.randint(0, 4, (E,), device=dev, dtype=torch.int32)
per_edge = table[types.long()].contiguous()
aggs = ["mean", "max", "min", "std"]
out = torch.empty(V, 4 * F, device=dev)
kw = dict(tower_stride_in=F, dst_term=dt, heavy=g.heavy_schedule(), workspace=g.workspace, out=out)


def ev(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


res = {
    "no_edge_term_ms": ev(lambda: ops.segreduce(c.rowptr, c.col, x, F, aggs, (None,), items=g.work_items(), tune=dict(generic=2), **kw)),
    "type_table_ms": ev(lambda: ops.segreduce(c.rowptr, c.col, x, F, aggs, (None,), edge_term=table, edge_type=types, items=g.work_items(), tune=dict(generic=2), **kw)),
    "per_edge_term_ms": ev(lambda: ops.segreduce(c.rowptr, c.col, x, F, aggs, (None,), edge_term=per_edge, items=g.work_items(), tune=dict(generic=2), **kw)),
    "per_edge_term_generic_kernel_ms": ev(lambda: ops.segreduce(c.rowptr, c.col, x, F, aggs, (None,), edge_term=per_edge, tune=dict(generic=1), **kw)),
}
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
