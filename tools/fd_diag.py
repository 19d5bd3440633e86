#!/usr/bin/env python
"""Diagnostics of ONE shape of the one-kernel layer (pna_fused_degree_f32): HIP-event time of the group-rows kernel and -- with the
experiments build (tools/build_experiments.sh, PNA_AMD_LIB=pna_amd/lib/libpna_amd_exp.so) -- its per-wavefront phase timers
(gather | multiply | epilogue cycles) and ablations; `--pmc`: a handful of launches for a rocprofv3 --pmc pass.

    FD_V=2000000 FD_E=20000000 FD_F=128 python tools/fd_diag.py [json-out]              # BASELINE configs[4]'s per-GPU shape
    FD_V=... rocprofv3 --pmc FETCH_SIZE -- python tools/fd_diag.py --pmc
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
pmc = "--pmc" in sys.argv
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
torch.manual_seed(0)
layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True).to(dev).eval()
h = torch.randn(V, (F + 7) // 8 * 8, device=dev)[:, :F]
plan = DG.plan_of(g)


def ev(fn, n=10, reps=3):
    for _ in range(2):
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


with torch.no_grad():
    assert DG.fused_applies(g, h, F, F), "the one-kernel path does not apply"
    call = PF.FusedDegreeCall(layer, g, h, x=h)
    beside = plan.rest_overlap_applies(F)
    call.set_spare(beside)
    if pmc:
        for _ in range(int(os.environ.get("FD_PMC_LAUNCHES", 8))):
            call.group_rows()
        torch.cuda.synchronize()
        print(f"PMC V={V} E={E} F={F} spare={int(call.args.spare_workgroups)} rows={plan.NV} records={plan.fused_tables()[2]}")
        sys.exit(0)
    e_g = plan.edge_split()[0]
    rows_g = int((plan.perm >= 0).sum())
    out = {"V": V, "E": E, "F": F, "groups": plan.G, "rest_rows": plan.NR, "padded_rows": plan.NV, "rows": rows_g, "edges": e_g,
           "spare_workgroups": int(call.args.spare_workgroups), "lib": os.path.basename(_lib.LIB_PATH)}
    out["group_rows_ms"] = ev(call.group_rows)
    if os.environ.get("FD_PARITY"):                          # the kernel's rows against the two-kernel grouped path (same statistics, same weights)
        y1 = call.group_rows().clone()
        live = plan.perm[plan.perm >= 0].long()
        bits = y1[live].contiguous().view(torch.int32).long()             # a checksum of the output BITS: equal between two builds = bit-identical results
        w = torch.arange(1, bits.numel() + 1, device=dev).view_as(bits) % 1000003
        out["output_bits_checksum"] = [int(bits.sum()), int((bits * w).sum())]
        print("output bits checksum:", out["output_bits_checksum"], flush=True)
        DG.FUSED = False
        y2 = layer(g, h)
        DG.FUSED = True
        s_ = y2.abs().max().item()
        out["max_diff_vs_two_kernel_of_max"] = (y1[live] - y2[live]).abs().max().item() / s_
        out["rows_differing_over_2e-6"] = int(((y1[live] - y2[live]).abs().max(dim=1).values > 2e-6 * s_).sum())
        print(f"parity vs two-kernel grouped path: {out['max_diff_vs_two_kernel_of_max']:.2e} of max|y|, {out['rows_differing_over_2e-6']} rows over 2e-6", flush=True)
        del y2
    alg = e_g * (4 * F + 4) + plan.NV + rows_g * (8 * F + 4)
    out["algorithmic_bytes"] = alg
    out["frac_of_8TBps"] = alg / (out["group_rows_ms"] * 1e-3) / 8e12
    print(f"group rows {out['group_rows_ms']:.3f} ms = {out['frac_of_8TBps']:.3f} of 8 TB/s on {alg / 1e9:.2f} GB", flush=True)
    if "exp" in os.path.basename(_lib.LIB_PATH):
        props = torch.cuda.get_device_properties(0)
        nw = props.multi_processor_count * 2 * 4
        dbg = torch.zeros(nw * 8, dtype=torch.int64, device=dev)
        os.environ["PNA_FD_DBG_PTR"] = hex(dbg.data_ptr())
        call.group_rows()
        torch.cuda.synchronize()
        del os.environ["PNA_FD_DBG_PTR"]
        d = dbg.view(nw, 8).double()
        d = d[d[:, 3] > 0]
        out["phase_timers"] = {"wavefronts": int(d.shape[0]), "total_cycles_mean": d[:, 3].mean().item(), "gather_frac": (d[:, 0] / d[:, 3]).mean().item(),
                               "multiply_frac": (d[:, 1] / d[:, 3]).mean().item(), "epilogue_frac": (d[:, 2] / d[:, 3]).mean().item(),
                               "total_cycles_min": d[:, 3].min().item(), "total_cycles_max": d[:, 3].max().item(),
                               "inside_multiply": {"copy_wait_frac": (d[:, 4] / d[:, 3]).mean().item(),
                                                   "first_barrier_of_a_pass_frac": (d[:, 5] / d[:, 3]).mean().item(),
                                                   "other_barriers_frac": (d[:, 6] / d[:, 3]).mean().item()}}
        print("phase timers:", json.dumps(out["phase_timers"]), flush=True)
        for abl, what in [(0, "nothing skipped"), (1, "no MFMAs"), (16, "no B-fragment reads"), (17, "no MFMAs, no B-fragment reads"),
                          (19, "no MFMA / fragment maths / B reads"), (4, "no fold (loads still issued and waited for)"), (8, "no y stores"),
                          (64, "no s_setprio"), (23, "no MFMA / fragment maths / B reads / fold: loads, waits, barriers, weight copies, epilogue"),
                          (128, "no weight copies (L2 -> LDS stream gone)"), (151, "no weight copies, no MFMA / fragment / B reads / fold"),
                          (256, "every gather packet reads row 0 (no HBM gather traffic)"), (384, "no weight copies, gather from row 0"),
                          (407, "no compute, no weight copies, gather from row 0: barriers, waits, descriptors, epilogue"),
                          (415, "... and no y stores"),
                          (0, "nothing skipped (again)")]:
            os.environ["PNA_FD_ABL"] = str(abl)
            out[f"ablation_{abl}_ms"] = ev(call.group_rows, n=5, reps=2)
            print(f"ablation {abl:2d} ({what}): {out[f'ablation_{abl}_ms']:.3f} ms", flush=True)
        del os.environ["PNA_FD_ABL"]
        for wgs in (1, 2):
            os.environ["PNA_FD_WGS"] = str(wgs)
            out[f"wgs_per_cu_{wgs}_ms"] = ev(call.group_rows, n=5, reps=2)
            print(f"workgroups per CU {wgs}: {out[f'wgs_per_cu_{wgs}_ms']:.3f} ms", flush=True)
        del os.environ["PNA_FD_WGS"]
if len(sys.argv) > 1 and not sys.argv[1].startswith("--"):
    json.dump(out, open(sys.argv[1], "w"), indent=1)
