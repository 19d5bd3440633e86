#!/usr/bin/env python
"""pna_posttrans_dw_f32 against the library route it replaces at the C3 shape (M = 1e6, N = 75, K = 300, Kh = 75, 3 scalers)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import autograd as AG, ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, K, Kh = 1_000_000, 75, 300, 75
gy, a, h = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev), torch.randn(M, Kh, device=dev)
amp = torch.rand(M, device=dev) + 0.5
att = 1 / amp
scales = [None, amp, att]


def ev(fn, n=10, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best


def library():
    G3 = gy.new_empty(M, 3, N)
    G3[:, 0].copy_(gy)
    torch.mul(gy, amp.unsqueeze(1), out=G3[:, 1])
    torch.mul(gy, att.unsqueeze(1), out=G3[:, 2])
    gw = AG._tall_tn(G3.view(M, 3 * N), a)
    return torch.cat([AG._tall_tn(gy, h)] + [gw[s * N:(s + 1) * N] for s in range(3)], dim=1), AG._column_sums(gy)


from pna_amd import Graph, degree_groups as DG  # noqa: E402
from pna_amd.dgl.pna_layer import _row_scales  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402
src, dst = powerlaw_graph(M, 10 * M, seed=1234, device=dev)
g = Graph(src, dst, M)
plan = DG.plan_of(g)
dscales = _row_scales(g, ["identity", "amplification", "attenuation"], {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}, dev)
t_g = ev(lambda: ops.posttrans_dw_grouped(gy, a, K, h, dscales, plan))
gwg, _ = ops.posttrans_dw_grouped(gy, a, K, h, dscales, plan)
wantg = torch.cat([gy.double().t() @ h.double()] + [(gy.double() if r is None else gy.double() * r.double().reshape(-1, 1)).t() @ a.double() for r in dscales], dim=1)
t_k = ev(lambda: ops.posttrans_dw(gy, a, K, h, scales))
t_l = ev(library)
gw, gb = ops.posttrans_dw(gy, a, K, h, scales)
lw, lb = library()
g64 = gy.double()
want = torch.cat([g64.t() @ h.double()] + [(g64 if r is None else g64 * r.double().unsqueeze(1)).t() @ a.double() for r in scales], dim=1)
out = {"M": M, "N": N, "K": K, "Kh": Kh, "kernel_degree_plan_order_ms": t_g, "kernel_per_row_scalers_ms": t_k, "library_route_ms": t_l,
       "degree_plan_order_GB_per_s_algorithmic": 4.0 * M * (N + K + Kh + 1) / t_g / 1e6, "workspace_entries": plan.dw_tables(256)[4],
       "max_err_degree_plan_order_rel_to_max": ((gwg.double() - wantg).abs().max() / wantg.abs().max()).item(),
       "flops_executed": 2.0 * M * 240 * 384, "per_row_kernel_TFLOPs_fp32_equivalent": 2.0 * M * (225 * 300 + 75 * 76) / t_k / 1e9,
       "algorithmic_bytes": 4.0 * M * (N + K + Kh + 2),
       "max_err_kernel_rel_to_max": ((gw.double() - want).abs().max() / want.abs().max()).item(),
       "max_err_library_rel_to_max": ((lw.double() - want).abs().max() / want.abs().max()).item()}
print(json.dumps(out))
