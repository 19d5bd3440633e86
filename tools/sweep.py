#!/usr/bin/env python
"""Tuning sweep of the fused segment-reduce kernel on the roofline workload (one GPU call, many
variants, interleaved rounds, min and median per variant).  Writes gpurun_out/sweep_<tag>.json.

    python tools/sweep.py --tag r01 [--V 1000000 --E 10000000 --F 75]
"""
import argparse
import itertools
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, ops  # noqa: E402
from pna_amd.graph import build_heavy_schedule  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--V", type=int, default=1_000_000)
    ap.add_argument("--E", type=int, default=10_000_000)
    ap.add_argument("--F", type=int, default=75)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    V, E, F = args.V, args.E, args.F
    src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
    g = Graph(src, dst, V)
    c = g.csr
    amp, att = g.degree_scalers(2.2488)
    xs = {}
    base = torch.randn(V, F, generator=torch.Generator().manual_seed(1234)).to(dev)
    for ld in (75, 76, 80, 96):
        buf = torch.zeros(V, ld, device=dev)
        buf[:, :F] = base
        xs[ld] = buf[:, :F]
    heavy = {0: None}
    for thr, seg in ((32, 32), (64, 64), (128, 64), (128, 128), (256, 128), (512, 256), (96, 96), (192, 96), (192, 192), (256, 256)):
        heavy[(thr, seg)] = build_heavy_schedule(c.rowptr, c.max_degree, thr, seg)
    cols = {"real": c.col, "l2": (c.col % 4096).contiguous(),                       # every gather hits L2: issue-bound floor
            "seq": (torch.arange(E, device=dev, dtype=torch.int32) % V).contiguous()}  # streaming sources
    aggs = ["mean", "max", "min", "std"]
    alg_bytes = E * (4 * F + 4) + 4 * (V + 1) + V * 16 * F

    variants = []
    # 0. memory-system experiments: same instruction stream, different source locality
    for cm in ("l2", "seq"):
        for U in (4, 8):
            variants.append(dict(ld=75, heavy=(64, 64), S=1, col=cm, tune=dict(unroll=U, rows_per_group=4)))
    for cm in ("real", "l2"):
        variants.append(dict(ld=75, heavy=(64, 64), S=1, col=cm, nocheck=1, tune=dict(unroll=4, rows_per_group=4, debug=1)))
    # 1. tuning grid at ld=75, default heavy (64,64), 4F output
    for U, R in itertools.product((4,), (4, 8, 16)):
        variants.append(dict(ld=75, heavy=(64, 64), S=1, tune=dict(unroll=U, rows_per_group=R)))
    for U in (2, 4):
        variants.append(dict(ld=75, heavy=(64, 64), S=1, tune=dict(unroll=U, rows_per_group=4, generic=1)))
    for hk in ((64, 64), (128, 128)):
        for ld in (75, 80):
            for R in (4, 8):
                variants.append(dict(ld=ld, heavy=hk, S=1, order="natural", tune=dict(unroll=4, rows_per_group=R)))
    for U in (2, 3, 4, 5, 6, 8):
        for R in (2, 4):
            variants.append(dict(ld=80, heavy=(128, 128), S=1, tune=dict(unroll=U, rows_per_group=R)))
    for hk in ((96, 96), (192, 96), (192, 192), (256, 256)):
        variants.append(dict(ld=80, heavy=hk, S=1, tune=dict(unroll=4, rows_per_group=4)))
    for hk in ((64, 64), (128, 128)):
        for ld in (75, 80):
            variants.append(dict(ld=ld, heavy=hk, S=1, tune=dict(unroll=4, rows_per_group=8)))
    # 2. leading dimension of x
    for ld in (76, 80, 96):
        for U in (4, 8):
            variants.append(dict(ld=ld, heavy=(64, 64), S=1, tune=dict(unroll=U, rows_per_group=4)))
    # 3. heavy schedule
    for hk in heavy:
        if hk != (64, 64):
            variants.append(dict(ld=75, heavy=hk, S=1, tune=dict(unroll=4, rows_per_group=4)))
    # 4. store policy, materialised 12F output, scalar path
    variants.append(dict(ld=75, heavy=(64, 64), S=1, tune=dict(unroll=4, rows_per_group=4, nt_store=-1)))
    variants.append(dict(ld=75, heavy=(64, 64), S=3, tune=dict(unroll=4, rows_per_group=4)))
    variants.append(dict(ld=75, heavy=(64, 64), S=3, tune=dict(unroll=4, rows_per_group=4, nt_store=-1)))
    variants.append(dict(ld=75, heavy=(64, 64), S=1, tune=dict(unroll=4, rows_per_group=4, vec=1)))
    variants.append(dict(ld=80, heavy=(64, 64), S=1, tune=dict(unroll=4, rows_per_group=4, lanes_per_row=32)))
    if args.quick:
        variants = variants[::6]

    outs = {}
    for S in (1, 3):
        outs[S] = torch.empty(V, 4 * S * F, device=dev)
    big = torch.empty(V, 4 * 96, device=dev)
    for bs in (76, 80, 96):
        variants.append(dict(ld=80, heavy=(128, 128), S=1, bs=bs, nocheck=1, tune=dict(unroll=4, rows_per_group=8)))
    for na in (1, 2, 3):
        variants.append(dict(ld=80, heavy=(128, 128), S=1, na=na, nocheck=1, tune=dict(unroll=4, rows_per_group=8)))

    def run(v):
        scales = [None] if v["S"] == 1 else [None, amp, att]
        hk = v["heavy"]
        ag = aggs[:v.get("na", 4)]
        if "bs" in v:
            o, bs = big[:, :4 * v["bs"]], v["bs"]
        else:
            o, bs = outs[v["S"]][:, :len(ag) * v["S"] * F], F
        return ops.segreduce(c.rowptr, cols[v.get("col", "real")], xs[v["ld"]], F, ag, scales, out=o, block_stride=bs,
                             heavy=heavy[hk if hk else 0], workspace=g.workspace, tune=v["tune"],
                             items=g.work_items(hk[0] if hk else 0, hk[1] if hk else 0, v.get("order", "natural"), v.get("window", 96)))

    ref_v = dict(ld=75, heavy=(64, 64), S=1, tune=dict(unroll=4, rows_per_group=4))
    ref = run(ref_v).clone()
    times = [[] for _ in variants]
    for i, v in enumerate(variants):                       # correctness of every variant first (4F part bit-identical
        o = run(v)                                         # for variants sharing the heavy schedule)
        if v["heavy"] == ref_v["heavy"] and v.get("col", "real") == "real" and not v.get("nocheck"):
            same_map = not ({"vec", "lanes_per_row", "generic"} & set(v["tune"]))     # heavy-row fold order depends on the lane mapping
            if same_map:
                assert torch.equal(o[:, :4 * F], ref), v
            else:
                torch.testing.assert_close(o[:, :4 * F], ref, rtol=1e-4, atol=1e-4)
    for r in range(args.rounds):
        for i, v in enumerate(variants):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(v)
            a.record()
            for _ in range(args.iters):
                run(v)
            b.record()
            torch.cuda.synchronize()
            times[i].append(a.elapsed_time(b) / args.iters)
    res = []
    for v, t in zip(variants, times):
        t = sorted(t)
        res.append(dict(v, col=v.get("col", "real"), heavy=list(v["heavy"]) if v["heavy"] else None, ms_min=t[0], ms_med=t[len(t) // 2],
                        frac_hbm_min=alg_bytes / (t[0] * 1e-3) / 8e12, gedges_s=E / (t[0] * 1e-3) / 1e9))
    res.sort(key=lambda r: r["ms_med"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", f"sweep_{args.tag}.json")
    json.dump(dict(V=V, E=E, F=F, alg_bytes=alg_bytes, results=res), open(path, "w"), indent=1)
    for r in res:
        print(f"{r['ms_med']:.3f} {r['ms_min']:.3f} col={r['col']} ld={r['ld']} heavy={r['heavy']} S={r['S']} bs={r.get('bs')} na={r.get('na')} ord={r.get('order')}/{r.get('window')} {r['tune']}")


if __name__ == "__main__":
    main()
