#!/usr/bin/env python
"""The one-kernel layer (pna_fused_degree_f32) on the C3 graph: a quick parity check against the two-kernel degree-grouped path,
HIP-event timings of its two launches and of the whole layer on both paths (interleaved repeats in one process), and -- with the
experiments build (tools/build_experiments.sh, PNA_AMD_LIB=pna_amd/lib/libpna_amd_exp.so) -- the kernel's per-wavefront phase
timers (gather | multiply | epilogue cycles) and the PNA_FD_WGS (workgroups per CU) knob.

    python tools/fd_time.py [json-out]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import _lib  # noqa: E402
if os.environ.get("PNA_AMD_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["PNA_AMD_LIB"])
from pna_amd import Graph, degree_groups as DG, functional as PF  # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

dev = torch.device("cuda:0")
V, E, F = int(os.environ.get("FD_V", 1_000_000)), int(os.environ.get("FD_E", 10_000_000)), int(os.environ.get("FD_F", 75))
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
torch.manual_seed(0)
layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True)
with torch.no_grad():
    for p in layer.parameters():
        p.copy_(torch.randn_like(p) / (p.shape[-1] ** 0.5 if p.dim() == 2 else 3.0))
layer = layer.to(dev).eval()
h = torch.randn(V, (F + 7) // 8 * 8, device=dev)[:, :F]
plan = DG.plan_of(g)


def ev(fn, n=20, reps=3):
    for _ in range(3):
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


out = {"V": V, "E": E, "F": F, "groups": plan.G, "rest_rows": plan.NR, "padded_rows": plan.NV}
with torch.no_grad():
    assert DG.fused_applies(g, h, F, F), "the one-kernel path does not apply"
    call = PF.FusedDegreeCall(layer, g, h)
    out["id_records"] = plan.fused_tables()[2]
    y_f = PF.simple_layer_degree_fused(layer, g, h).clone()
    DG.FUSED = False
    y_g = layer(g, h).clone()
    DG.FUSED = True
    s = y_g.abs().max().item()
    out["max_diff_vs_two_kernel_of_max"] = (y_f - y_g).abs().max().item() / s
    out["rows_differing_over_2e-6"] = int(((y_f - y_g).abs().max(dim=1).values > 2e-6 * s).sum())
    print(f"parity: one-kernel vs two-kernel grouped {out['max_diff_vs_two_kernel_of_max']:.2e} of max|y| ({out['rows_differing_over_2e-6']} rows over 2e-6)", flush=True)
    hc = h.contiguous()                                       # (V, F) rows of 300 bytes: what a drop-in caller passes
    if hc.untyped_storage().nbytes() // 4 < (V - 1) * F + (F + 7) // 8 * 8:
        hc = torch.cat([hc.reshape(-1), torch.zeros(8, device=dev)])[:V * F].view(V, F)
    assert DG.fused_applies(g, hc, F, F), "the one-kernel path does not take a contiguous table"
    call_c = PF.FusedDegreeCall(layer, g, hc)
    y_c = PF.simple_layer_degree_fused(layer, g, hc).clone()
    out["contiguous_x_max_diff_vs_pitch80_of_max"] = (y_c - y_f).abs().max().item() / s
    for rep in range(2):
        out[f"rep{rep}_fused_group_rows_contiguous_x_ms"] = ev(call_c.group_rows)
        print(f"rep {rep}: fused group-rows kernel over a CONTIGUOUS (V, {F}) table {out[f'rep{rep}_fused_group_rows_contiguous_x_ms']:.3f} ms (max diff vs pitch 80: {out['contiguous_x_max_diff_vs_pitch80_of_max']:.1e})", flush=True)
        t_f, t_r = ev(call.group_rows), ev(call.rest_rows)
        t_layer_f = ev(lambda: layer(g, h))
        DG.FUSED = False
        t_layer_g = ev(lambda: layer(g, h))
        DG.FUSED = True
        print(f"rep {rep}: fused group-rows kernel {t_f:.3f} ms, rest path {t_r:.3f} ms, layer one-kernel {t_layer_f:.3f} ms, layer two-kernel {t_layer_g:.3f} ms", flush=True)
        out[f"rep{rep}"] = {"fused_group_rows_ms": t_f, "rest_rows_ms": t_r, "layer_one_kernel_ms": t_layer_f, "layer_two_kernel_ms": t_layer_g}
    for seg, rpg in [(128, 4), (128, 1), (64, 1), (32, 1), (32, 2), (16, 1)]:          # the rest launch: hub segment length x items per lane group
        DG.REST_SEG_LEN, DG.REST_ROWS_PER_GROUP, plan._rest_items = seg, rpg, None
        c2 = PF.FusedDegreeCall(layer, g, h)
        c2.group_rows()
        y2 = c2.rest_rows().clone()
        d = (y2 - y_g).abs().max().item() / s
        out[f"rest_seg{seg}_rpg{rpg}_ms"] = ev(c2.rest_rows)
        print(f"rest launch, {seg}-edge hub segments, {rpg} items per lane group: {out[f'rest_seg{seg}_rpg{rpg}_ms']:.3f} ms (max diff vs two-kernel {d:.1e})", flush=True)
    DG.REST_SEG_LEN, DG.REST_ROWS_PER_GROUP, plan._rest_items = 32, 1, None
    if "exp" in os.path.basename(_lib.LIB_PATH):
        import ctypes
        props = torch.cuda.get_device_properties(0)
        nw = props.multi_processor_count * 2 * 4
        dbg = torch.zeros(nw * 4, dtype=torch.int64, device=dev)
        os.environ["PNA_FD_DBG_PTR"] = hex(dbg.data_ptr())
        call.group_rows()
        torch.cuda.synchronize()
        del os.environ["PNA_FD_DBG_PTR"]
        d = dbg.view(nw, 4).double()
        d = d[d[:, 3] > 0]
        tot = d[:, 3].mean().item()
        out["phase_timers"] = {"wavefronts": int(d.shape[0]), "total_cycles_mean": tot, "gather_frac": (d[:, 0] / d[:, 3]).mean().item(),
                               "multiply_frac": (d[:, 1] / d[:, 3]).mean().item(), "epilogue_frac": (d[:, 2] / d[:, 3]).mean().item(),
                               "total_cycles_min": d[:, 3].min().item(), "total_cycles_max": d[:, 3].max().item()}
        print("phase timers:", json.dumps(out["phase_timers"]), flush=True)
        for abl, what in [(0, "nothing skipped"), (1, "no MFMAs"), (2, "fragment maths only for the first chunk of a tile"), (3, "no MFMAs, no fragment maths"),
                          (4, "no fold (loads still issued and waited for)"), (8, "no y stores"), (32, "non-temporal y stores"), (64, "NO s_setprio 1 through the multiply phase and epilogue"), (0, "nothing skipped (again)"), (16, "no B-fragment reads"), (19, "no MFMA / fragment maths / B reads"),
                          (31, "everything skipped: loads, waits, barriers, weight copies, epilogue arithmetic")]:
            os.environ["PNA_FD_ABL"] = str(abl)
            out[f"ablation_{abl}_ms"] = ev(call.group_rows)
            print(f"ablation {abl:2d} ({what}): {out[f'ablation_{abl}_ms']:.3f} ms", flush=True)
        del os.environ["PNA_FD_ABL"]
        for wgs in (1, 2):
            os.environ["PNA_FD_WGS"] = str(wgs)
            try:
                out[f"wgs_per_cu_{wgs}_ms"] = ev(call.group_rows)
                print(f"workgroups per CU {wgs}: {out[f'wgs_per_cu_{wgs}_ms']:.3f} ms", flush=True)
            except Exception as ex:    # noqa: BLE001
                print(f"workgroups per CU {wgs}: {ex}", flush=True)
        del os.environ["PNA_FD_WGS"]
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
