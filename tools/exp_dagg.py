#!/usr/bin/env python
"""d agg = sum_s scale_s (.) (gy W_s) at C3 size (M = 1e6, 75 -> 300): one call (the launcher's 80-column blocks) against
explicit 128-column slices and against the degree-grouped one-block form over the plan's rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pna_amd import Graph, ops, degree_groups as DG
from pna_amd.synth import powerlaw_graph
dev = torch.device("cuda:0")
V, E, N, K = 1_000_000, 10_000_000, 75, 300
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
amp, att = g.degree_scalers(2.0)
scales = [None, amp, att]
gy = torch.randn(V, N, device=dev)
wt = torch.randn(K, 3 * N, device=dev) / 8


def ev(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


out = torch.empty(V, K, device=dev)
one = lambda: ops.posttrans(gy, N, wt, scales, None, out=out, arith="bf16x3")
ref = one().clone()
def sliced(width):
    def f():
        for c0 in range(0, K, width):
            c1 = min(K, c0 + width)
            ops.posttrans(gy, N, wt[c0:c1].contiguous(), scales, None, out=out[:, c0:c1], arith="bf16x3")
        return out
    return f
print("one call            %.3f ms" % ev(one))
for w in (128, 100, 80, 75):
    f = sliced(w)
    err = (f() - ref).abs().max().item()
    print("slices of %3d cols  %.3f ms  (max diff %.1e)" % (w, ev(f), err))
# degree-grouped: rows in plan order, one block, four 80-row slices of the weight
plan = DG.plan_of(g)
def grouped():
    gp = ops.pack_rows(gy, plan.perm_all()[:plan.NV])
    for c0 in range(0, K, 80):
        c1 = min(K, c0 + 80)
        w = wt[c0:c1].contiguous()
        img, stride = DG.combined_images(w, N, scales, plan)
        ops.posttrans(gp, N, w, [None], None, out=out[:, c0:c1], row_perm=plan.perm, tile_image=plan.tile_image, w_img=img, image_stride=stride, n_out=c1 - c0)
    return out
r = grouped()
real = plan.perm[plan.perm >= 0].long()
print("degree-grouped (group rows only, incl. row packing + image packing) %.3f ms  (max diff on those rows %.1e)" % (ev(grouped), (r[real] - ref[real]).abs().max().item()))
