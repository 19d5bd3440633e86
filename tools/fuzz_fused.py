#!/usr/bin/env python
"""Randomised parity sweep of the one-kernel layer (pna_fused_degree_f32, all shape classes incl. the wide ones and the tower mode)
against the two-kernel degree-grouped path on random graphs / shapes / epilogue options: statistics bit-identical, outputs within
2e-6 of max|y| (the same combined weight up to the order of its fp32 combination).  Prints one line per case and a summary.
    python tools/fuzz_fused.py [seconds] [seed] [big]      big: graphs of 0.6 .. 1.2 M nodes and F >= 64 -- the rest rows then run
                                                           BESIDE the kernel (functional.run_fused_call)"""
import os, sys, time, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, degree_groups as DG, functional as PF
from pna_amd.dgl.pna_layer import PNALayer, PNASimpleLayer
from pna_amd.synth import powerlaw_graph

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
n_beside = 0
dev = torch.device("cuda:0")
DG.MIN_ROWS, DG.MIN_OUT, DG.TWO_SCALER_MIN_OUT, PF.SMALL_SIMPLE_ROWS, PF.SMALL_TOWER_ROWS = 1, 1, 1, 0, 0
t0, n_ok, n_skip, worst = time.time(), 0, 0, 0.0
while time.time() - t0 < budget:
    tower = rnd.random() < 0.25
    wide = (not tower) and rnd.random() < 0.3
    n_towers, kind = 1, "tower" if tower else "wide " if wide else "plain"
    if tower:
        F = rnd.randint(49, 80); N = rnd.choice([F, rnd.randint(4, 80)])
        if rnd.random() < 0.4:                              # round 6: T towers over the whole input (one launch per tower + the dense term)
            n_towers, kind = rnd.randint(2, 5), "multi"
            N = n_towers * rnd.randint(max(1, 4 // n_towers + 1), 80 // n_towers)
    elif wide:
        F, N = rnd.choice([(rnd.randint(97, 128), rnd.randint(81, 128)), (rnd.randint(97, 128), rnd.randint(4, 80)), (rnd.randint(49, 64), rnd.randint(81, 128))])
    else:
        F = rnd.randint(17, 80); N = rnd.randint(4, 80)
        r6 = rnd.random()
        if r6 < 0.15:                                       # round 6: feature panels [0, 64) + [64, F)
            F, kind = rnd.randint(81, 96), "fpan "
            N = rnd.choice([F, rnd.randint(4, 150)])
        elif r6 < 0.3:                                      # round 6: output-column panels
            N, kind = rnd.randint(81, 230), "cpan "
    if BIG and not wide:
        F = max(F, rnd.randint(64, 80))
    V = rnd.choice([600000, 900000, 1200000]) if BIG else rnd.choice([3000, 20000, 70000, 150000])
    E = int(V * rnd.choice([2, 4, 8, 14]))
    E += (E - 2 * V) % 2
    src, dst = powerlaw_graph(V, E, seed=rnd.randint(0, 10 ** 6), device=dev)
    if rnd.random() < 0.5:                                  # some rows without in-edges
        keep = dst >= rnd.randint(1, 200)
        src, dst = src[keep], dst[keep]
    g = Graph(src, dst, V)
    scalers = rnd.choice(["identity amplification attenuation", "identity amplification attenuation", "identity amplification", "amplification attenuation", "identity"])
    aggregators = "mean max min std"
    if not tower and rnd.random() < 0.25:                   # round 6: any distinct aggregators out of mean / sum / max / min / std
        pool = ["mean", "sum", "max", "min", "std"]
        rnd.shuffle(pool)
        aggregators = " ".join(pool[:rnd.randint(1, 5)])
    torch.manual_seed(rnd.randint(0, 10 ** 6))
    # round 4: any row pitch >= F takes the kernel -- the 32-byte aligned pitch of rounds 2-3, the 16-byte one, or none (contiguous rows)
    pitch = rnd.choice([128 if F > 96 else (F + 7) // 8 * 8, (F + 3) // 4 * 4, F])
    h = torch.randn(V, pitch, device=dev)[:, :F]
    if tower:
        scalers = "identity amplification attenuation"
        layer = PNALayer(F, N, "mean max min std", scalers, {"log": torch.tensor(2.1)}, 0.0, rnd.random() < 0.5, rnd.random() < 0.5, towers=n_towers, divide_input=False,
                         residual=(F == N and rnd.random() < 0.5)).to(dev).eval()
        args = (g, h, None, g.snorm_n())
    else:
        layer = PNASimpleLayer(F, N, aggregators, scalers, {"log": torch.tensor(2.1)}, 0.0, True, F == N and rnd.random() < 0.5).to(dev).eval()
        args = (g, h)
    with torch.no_grad():
        if tower:
            ok_path = PF.tower_layer_degree_grouped_applies(layer, g, h) and PF.tower_layer_degree_fused_applies(layer, g, h)
        else:
            ok_path = layer._degree_grouped_path(g, h) and DG.fused_applies(g, h, F, N, aggregators.split())
        for m in layer.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(); m.running_var.uniform_(0.5, 2.0)
        if not ok_path:
            n_skip += 1
            continue
        DG.FUSED = True
        y_f = layer(*args)
        n_beside += int(DG.plan_of(g).rest_overlap_applies(F))
        if not tower and aggregators == "mean max min std" and len(DG.fused_panels(F, N)) == 1 and len(scalers.split()) >= 2:
            plan = DG.plan_of(g)
            dump = torch.zeros(plan.NV, 4 * F, device=dev)
            PF.simple_layer_degree_fused(layer, g, h, agg_out=dump)
            ref_agg = PF.degree_grouped_aggregate(layer, g, h, plan)[:plan.NV]
            real = plan.perm >= 0
            assert torch.equal(dump[real], ref_agg[real]), ("statistics differ", V, E, F, N)
        DG.FUSED = False
        y_g = layer(*args)
        DG.FUSED = True
    s = y_g.abs().max().item()
    err = (y_f - y_g).abs().max().item() / max(s, 1e-30)
    bar = 1e-5 if tower else 4e-6 if (kind in ("fpan ", "cpan ") or aggregators != "mean max min std") else 2e-6
    if not (torch.isfinite(y_f).all() and err <= bar):
        with torch.no_grad():
            diag = {}
            for ar in ("guarded", "bf16x3", "fp16x2"):
                DG.FUSED_ARITH = ar
                y_a = layer(*args)
                d = (y_a - y_g).abs()
                diag[ar] = (d.max().item() / max(s, 1e-30), int((d.amax(1) > bar * s).sum()))
            DG.FUSED_ARITH = "guarded"
            d = (y_f - y_g).abs().amax(1)
            bad_rows = torch.nonzero(d > bar * s).flatten()
            deg = g.in_degrees()
            print("FAILED case: worst rows", bad_rows[:10].tolist(), "their degrees", deg[bad_rows[:10]].tolist(), "in group rows", [bool((DG.plan_of(g).perm == r).any()) for r in bad_rows[:10].tolist()],
                  "per arith (max rel, rows over)", diag, "graph_norm/batch_norm/residual", getattr(layer, "residual", None), flush=True)
        raise AssertionError(("output differs", kind, V, E, F, N, n_towers, scalers, aggregators, err))
    worst = max(worst, err)
    n_ok += 1
    print(f"ok  {kind} V={V} E={src.numel()} F={F} N={N} towers={n_towers} pitch={pitch} scalers={len(scalers.split())} aggregators={aggregators.replace(' ', '+')} err={err:.1e}", flush=True)
print(f"SUMMARY {n_ok} cases passed ({n_beside} with the rest rows beside the kernel), {n_skip} skipped (path did not apply), worst relative difference {worst:.2e}, {time.time() - t0:.0f} s")
