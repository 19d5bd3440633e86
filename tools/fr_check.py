#!/usr/bin/env python
"""The one-kernel layer with wavefront roles (pna_fused_roles_f32) against the round-3 one-kernel layer (pna_fused_degree_f32) and
the two-kernel path: parity (outputs; statistics bit for bit through agg_out), the give-up flag, HIP-event timings, and -- with the
experiments build (PNA_AMD_LIB=pna_amd/lib/libpna_amd_exp.so) -- the wait counters of both roles, the SIMD placement of the
workgroup's wavefronts and the ablation knobs (PNA_FR_ABL) / ring depth (PNA_FR_RING).

    python tools/fr_check.py [json-out]          FR_STAGES=small,c3,abl (default: all)
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
STAGES = os.environ.get("FR_STAGES", "small,c3,abl").split(",")
EXP = "exp" in os.path.basename(_lib.LIB_PATH)
out = {}


def make(V, E, F, N=None, seed=1234, pitch=None):
    N = F if N is None else N
    src, dst = powerlaw_graph(V, E, seed=seed, device=dev)
    g = Graph(src, dst, V)
    avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
    torch.manual_seed(0)
    layer = PNASimpleLayer(F, N, "mean max min std", "identity amplification attenuation", avg, 0.0, True, F == N)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(torch.randn_like(p) / (p.shape[-1] ** 0.5 if p.dim() == 2 else 3.0))
    layer = layer.to(dev).eval()
    pitch = (F + 7) // 8 * 8 if pitch is None else pitch
    h = torch.randn(V, pitch, device=dev)[:, :F] if pitch != F else torch.randn(V, F, device=dev)
    return g, layer, h


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


def parity(tag, V, E, F, N=None, pitch=None, stats=True):
    g, layer, h = make(V, E, F, N, pitch=pitch)
    plan = DG.plan_of(g)
    N = layer.out_dim
    r = {"V": V, "E": E, "F": F, "N": N, "pitch": h.stride(0), "groups": plan.G, "rest_rows": plan.NR, "virtual_rows": plan.NV}
    with torch.no_grad():
        if not plan.G:
            print(tag, "no degree groups", flush=True)
            return r
        hp = h if h.stride(0) % 4 == 0 else torch.nn.functional.pad(h, (0, (F + 7) // 8 * 8 - F))[:, :F]
        y_old = torch.full((V, (N + 3) // 4 * 4), float("nan"), device=dev)[:, :N]
        agg_old = torch.zeros(plan.NV, 4 * F, device=dev) if stats else None
        c_old = PF.FusedDegreeCall(layer, g, hp, x=hp, out=y_old, agg_out=agg_old)
        c_old.set_spare(False)
        c_old.group_rows()
        c_old.rest_rows()
        y_new = torch.full((V, (N + 3) // 4 * 4), float("nan"), device=dev)[:, :N]
        agg_new = torch.zeros(plan.NV, 4 * F, device=dev) if stats else None
        c_new = PF.FusedRolesCall(layer, g, h, x=h, out=y_new, agg_out=agg_new)
        c_new.group_rows()
        c_new.rest_rows()
        torch.cuda.synchronize()
        r["err_flag"] = int(c_new.err.item())
        s = y_old.abs().max().item()
        d = (y_new - y_old).abs()
        r["nan_rows_new"] = int(torch.isnan(y_new).any(dim=1).sum())
        r["max_diff_of_max"] = (d.max().item() / s) if r["nan_rows_new"] == 0 else float("nan")
        r["rows_over_2e-6"] = int((d.max(dim=1).values > 2e-6 * s).sum())
        if stats:
            live = plan.perm >= 0
            same = torch.equal(agg_new[live], agg_old[live])
            r["stats_bit_identical"] = bool(same)
            if not same:
                bad = (agg_new[live] != agg_old[live])
                r["stats_differing_elements"] = int(bad.sum())
                r["stats_differing_rows"] = int(bad.any(dim=1).sum())
                ii = bad.nonzero()[:5].tolist()
                r["stats_examples"] = [(i, j, float(agg_new[live][i, j]), float(agg_old[live][i, j])) for i, j in ii]
        print(tag, json.dumps(r), flush=True)
    return r


if "small" in STAGES:
    out["small_f75"] = parity("small F=75", 40_000, 400_000, 75)
    out["small_f75_contig"] = parity("small F=75 contiguous rows", 40_000, 400_000, 75, pitch=75)
    out["small_f64"] = parity("small F=64", 30_000, 300_000, 64)
    out["small_f40_n50"] = parity("small F=40 N=50", 30_000, 300_000, 40, 50)
    out["small_f20"] = parity("small F=20 N=24", 30_000, 240_000, 20, 24)
    out["small_f50"] = parity("small F=50 N=75", 30_000, 300_000, 50, 75)

if "c3" in STAGES or "abl" in STAGES:
    V, E, F = 1_000_000, 10_000_000, 75
    out["c3_parity"] = parity("C3", V, E, F, stats=False)
    g, layer, h = make(V, E, F)
    hc = h.contiguous()
    plan = DG.plan_of(g)
    with torch.no_grad():
        c_old = PF.FusedDegreeCall(layer, g, h, x=h)
        c_old.set_spare(False)
        c_new = PF.FusedRolesCall(layer, g, h, x=h)
        c_con = PF.FusedRolesCall(layer, g, hc, x=hc)
        t = {}
        for rep in range(2):
            t[f"old_group_rows_ms_{rep}"] = ev(c_old.group_rows)
            t[f"roles_group_rows_ms_{rep}"] = ev(c_new.group_rows)
            t[f"roles_contiguous_x_ms_{rep}"] = ev(c_con.group_rows)
        c_new.set_spare(16)
        t["roles_spare16_ms"] = ev(c_new.group_rows)
        c_new.set_spare(0)
        t["err_flag"] = int(c_new.err.item())
        out["c3_time"] = t
        print("C3 timings", json.dumps(t), flush=True)
        for cost in (0.0, 1.5, 6.0):
            DG.ROLES_TILE_COST = cost
            c2 = PF.FusedRolesCall(layer, g, h, x=h)
            out[f"c3_tile_cost_{cost}"] = ev(c2.group_rows)
            print(f"tile cost {cost}: {out[f'c3_tile_cost_{cost}']:.4f} ms", flush=True)
        DG.ROLES_TILE_COST = 3.0
        DG.ROLES_XCD_INTERLEAVE = True
        c2 = PF.FusedRolesCall(layer, g, h, x=h)
        out["c3_xcd_contiguous_eighths_ms"] = ev(c2.group_rows)
        print(f"XCD x takes the contiguous eighth x: {out['c3_xcd_contiguous_eighths_ms']:.4f} ms", flush=True)
        DG.ROLES_XCD_INTERLEAVE = False
        if EXP and "abl" in STAGES:
            n_wg = _lib.lib().pna_fused_roles_grid(0)
            dbg = torch.zeros(n_wg * 8 * 4 + 2 * n_wg * 4 * 8, dtype=torch.int64, device=dev)
            os.environ["PNA_FR_DBG_PTR"] = hex(dbg.data_ptr())
            c_new.group_rows()
            torch.cuda.synchronize()
            del os.environ["PNA_FR_DBG_PTR"]
            mph = dbg[n_wg * 32:n_wg * 64].view(n_wg, 4, 8).double()
            mst = dbg[n_wg * 64:].view(n_wg, 4, 8).double()
            d = dbg[:n_wg * 32].view(n_wg, 8, 4)
            tot = d[:, :, 3].double()
            hw = (d[:, :, 2] & 0xFFFFFFFF)
            simd = ((hw >> 4) & 3)
            w = {"G_total_cycles_mean": tot[:, :4].mean().item(), "M_total_cycles_mean": tot[:, 4:].mean().item(),
                 "total_cycles_max": tot.max().item(), "total_cycles_min": tot[tot > 0].min().item(),
                 "G_wait_for_M_frac": (d[:, :4, 0].double() / tot[:, :4]).mean().item(),
                 "M_wait_for_stats_frac": (d[:, 4:, 0].double() / tot[:, 4:]).mean().item(),
                 "M_wait_for_other_M_frac": (d[:, 4:, 1].double() / tot[:, 4:]).mean().item(),
                 "spins_mean": (d[:, :, 2] >> 32).double().mean().item(),
                 "pairs_on_one_simd_frac": (simd[:, :4] == simd[:, 4:]).double().mean().item(),
                 "simd_of_waves_wg0": simd[0].tolist(), "simd_of_waves_wg1": simd[1].tolist()}
            # per-workgroup spread of the finishing time (the static partition's balance)
            wg_end = tot.max(dim=1).values
            w["wg_total_cycles_p05_p50_p95_max"] = [wg_end.quantile(q).item() for q in (0.05, 0.5, 0.95)] + [wg_end.max().item()]
            names = ["hand-over (polls + reads)", "residual issue + accumulator reset", "multiply", "vmcnt wait", "epilogue", "descriptor", "image"]
            ntile = (plan.roles_tables(n_wg)[4][:, 1] - plan.roles_tables(n_wg)[4][:, 0]).double().cpu().to(dev).clamp(min=1)
            w["M_cycles_per_tile_by_phase"] = {names[i]: (mph[:, :, i].mean(dim=1) / ntile).mean().item() for i in range(7)}
            out["c3_counters"] = w
            print("counters", json.dumps(w), flush=True)
            # per workgroup: tiles, records, cycles (for the partition's cost model)
            desc, _, _, _, wg_range = plan.roles_tables(n_wg)
            wr = wg_range.long().cpu()
            nrec = desc[:, 1].clamp(min=1).long().cpu()
            cum = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(nrec, 0)])
            per = []
            for b in range(n_wg):
                lo, hi = int(wr[b, 0]), int(wr[b, 1])
                per.append([lo, hi, int(cum[hi] - cum[lo]), float(tot[b].max()), float(d[b, :4, 0].double().mean()), float(d[b, 4:, 0].double().mean())])
            out["c3_per_wg"] = per
            def m_phases():
                dbg2 = torch.zeros(n_wg * 8 * 4 + 2 * n_wg * 4 * 8, dtype=torch.int64, device=dev)
                os.environ["PNA_FR_DBG_PTR"] = hex(dbg2.data_ptr())
                c_new.group_rows()
                torch.cuda.synchronize()
                del os.environ["PNA_FR_DBG_PTR"]
                mp = dbg2[n_wg * 32:n_wg * 64].view(n_wg, 4, 8).double()
                ms = dbg2[n_wg * 64:].view(n_wg, 4, 8).double()
                return [round((mp[:, :, i].mean(dim=1) / ntile).mean().item()) for i in range(7)] + ["step:"] + [round((ms[:, :, i].mean(dim=1) / ntile).mean().item()) for i in range(5)]
            for abl, what in [(0, "nothing skipped"), (64, "ONE weight image for every tile (wrong results: L2 residency test)"), (1, "no MFMAs"), (2, "no statistics maths"), (4, "no fold"), (8, "no y stores"), (16, "no weight reads"),
                              (3, "no MFMAs, no statistics maths"), (18, "no statistics maths, no weight reads"), (32, "M side: hand-over and epilogue only"),
                              (36, "no fold, M side hand-over and epilogue only"), (0, "nothing skipped (again)")]:
                os.environ["PNA_FR_ABL"] = str(abl)
                k = f"ablation_{abl}_ms" + ("_again" if f"ablation_{abl}_ms" in out else "")
                out[k] = ev(c_new.group_rows)
                out[k.replace("_ms", "_M_phase_cycles_per_tile")] = ph = m_phases()
                print(f"ablation {abl:2d} ({what}): {out[k]:.4f} ms; M cycles per tile [hand-over, residual issue, multiply, vmcnt, epilogue, descriptor, image] = {ph}", flush=True)
            del os.environ["PNA_FR_ABL"]
            for ring, pg, pm in [(3, 0, 2), (4, 0, 2)]:
                os.environ["PNA_FR_RING"], os.environ["PNA_FR_PRIO_G"], os.environ["PNA_FR_PRIO_M"] = str(ring), str(pg), str(pm)
                key = f"ring{ring}_prioG{pg}_prioM{pm}"
                out[key + "_ms"] = ev(c_new.group_rows)
                ph = m_phases()
                os.environ["PNA_FR_ABL"] = "32"
                out[key + "_no_multiply_ms"] = ev(c_new.group_rows)
                ph32 = m_phases()
                del os.environ["PNA_FR_ABL"]
                print(f"ring {ring}, s_setprio G {pg} M {pm}: {out[key + '_ms']:.4f} ms, M phases {ph}; without the multiply {out[key + '_no_multiply_ms']:.4f} ms, M phases {ph32}", flush=True)
            for k_ in ("PNA_FR_RING", "PNA_FR_PRIO_G", "PNA_FR_PRIO_M"):
                del os.environ[k_]
            out["err_flag_after_ablations"] = int(c_new.err.item())
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
