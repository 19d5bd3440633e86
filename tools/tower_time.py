#!/usr/bin/env python
"""C3-scale tower layer (SURVEY 8d C3 iii: PNALayer(towers=1, F=75)) and a 5-tower ZINC-shaped layer at the same scale: the
degree-grouped path (functional.tower_layer_degree_grouped) against the ordinary kernels, HIP events."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, degree_groups as DG, functional as PF
from pna_amd.dgl.pna_layer import PNALayer
from pna_amd.synth import powerlaw_graph
dev = torch.device("cuda:0")
V, E = 1_000_000, 10_000_000
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V, [V // 4] * 4)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
snorm = g.snorm_n()


def ev(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best


out = {}
for T, F in ((1, 75), (5, 75)):
    torch.manual_seed(T)
    layer = PNALayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True, towers=T, divide_input=False, residual=True).to(dev).eval()
    h = torch.randn(V, 80, device=dev)[:, :F]
    with torch.no_grad():
        DG.ENABLED = True
        assert PF.tower_layer_degree_grouped_applies(layer, g, h)
        t_f = None
        if PF.tower_layer_degree_fused_applies(layer, g, h):
            y_f = layer(g, h, None, snorm).clone()
            t_f = ev(lambda: layer(g, h, None, snorm))
            towers = list(layer.towers)
            from pna_amd.dgl.pna_layer import _projection_cache_padded
            Wp, bp = _projection_cache_padded(towers[0], F, PF.tower_projection_pitch(F))
            x_cat = PF.linear_act(h, Wp, bp)
            call = PF.FusedTowerCall(layer, g, h, snorm, x_cat)
            parts = {"projection_ms": ev(lambda: PF.linear_act(h, Wp, bp)), "fused_group_rows_ms": ev(call.group_rows), "rest_rows_ms": ev(call.rest_rows)}
        DG.FUSED = False
        y_g = layer(g, h, None, snorm).clone()
        t_g = ev(lambda: layer(g, h, None, snorm))
        DG.FUSED = True
        DG.ENABLED = False
        y_p = layer(g, h, None, snorm)
        t_p = ev(lambda: layer(g, h, None, snorm))
        DG.ENABLED = True
    rel = lambda a, b: ((a - b).abs() / b.abs().max(dim=1, keepdim=True).values.clamp_min(1e-30)).max().item()   # noqa: E731  (of the row's max)
    err = rel(y_g, y_p)
    out[f"towers{T}_F{F}"] = {"degree_grouped_ms": t_g, "ordinary_ms": t_p, "max_diff_of_row_max": err}
    print(f"towers={T} F={F}: degree-grouped {t_g:.3f} ms, ordinary {t_p:.3f} ms, max diff {err:.1e} of the row max", flush=True)
    if t_f is not None:
        errf = rel(y_f, y_p)
        out[f"towers{T}_F{F}"].update(one_kernel_ms=t_f, one_kernel_max_diff_of_row_max=errf, **parts)
        print(f"   one-kernel {t_f:.3f} ms (max diff {errf:.1e}): " + ", ".join(f"{k} {v:.3f}" for k, v in parts.items()), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
