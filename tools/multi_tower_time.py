#!/usr/bin/env python
"""Where a 5-tower divide_input=False PNALayer's time goes on the one-kernel path at C3: projection, dense term, the five
launches, the rest rows -- each alone (HIP events), and the layer."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, functional as PF, degree_groups as DG   # noqa: E402
from pna_amd.dgl import pna_layer as PL                             # noqa: E402
from pna_amd.synth import powerlaw_graph                            # noqa: E402
dev = torch.device("cuda:0")
V, E, F = 1_000_000, 10_000_000, 75
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
torch.manual_seed(0)
layer = PL.PNALayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True, towers=5, divide_input=False, residual=True).to(dev).eval()
h = torch.randn(V, 80, device=dev)[:, :F]
sn = torch.rand(V, 1, device=dev) + 0.5


def ev(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


with torch.no_grad():
    towers = list(layer.towers)
    Wt, Wn = PL._projection_cache_padded_multi(towers, F, PF.tower_projection_pitch(F))
    Wpad, bpad = Wt.t().contiguous(), torch.zeros(Wt.shape[1], device=dev)
    out = {"layer_ms": ev(lambda: layer(g, h, None, sn)), "projection_torch_mm_ms": ev(lambda: torch.mm(h, Wt)),
           "projection_x3_kernel_ms": ev(lambda: PF.linear_act(h, Wpad, bpad))}
    x_src = torch.mm(h, Wt)
    from pna_amd import ops
    out["projection_resident_weight_kernel_ms"] = ev(lambda: ops.project(h, F, Wn))
    out["projection_f32_mfma_kernel_ms"] = ev(lambda: ops.posttrans(h, F, Wpad, [None], bpad, arith="f32"))
    call = PF.FusedMultiTowerCall(layer, g, h, sn, x_src)
    call.set_spare(False)
    out["dense_term_ms"] = ev(call.dense_term)
    PF.DENSE_TERM_RESIDENT = False
    out["dense_term_contraction_and_rank_update_ms"] = ev(call.dense_term)
    PF.DENSE_TERM_RESIDENT = True
    out["five_launches_ms"] = ev(call.group_rows)
    one = call.launch_order[0][1]
    out["first_launch_ms"] = ev(lambda: call.check(call.fn(one, call.stream), "x"))
    last = call.launch_order[-1][1]
    out["last_launch_ms"] = ev(lambda: call.check(call.fn(last, call.stream), "x"))
    out["rest_rows_ms"] = ev(call.rest_rows)
    simple = PL.PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True).to(dev).eval()
    c2 = PF.FusedDegreeCall(simple, g, h, x=h)
    c2.set_spare(False)
    out["simple_layer_group_rows_ms"] = ev(c2.group_rows)
print(json.dumps({k: round(v, 4) for k, v in out.items()}))
