#!/usr/bin/env python
"""Round-2 kernel experiments on the roofline workload (one GPU call, several sections, JSON to gpurun_out/).

    python tools/exp_r02.py [occupancy] [posttrans] [tower] [c5]

occupancy -- the gather kernel with its blocks per CU capped by dummy dynamic LDS (needs the experiments build,
             tools/build_experiments.sh): how many gather wavefronts per CU the HBM-bound phase needs.  This is the
             evidence behind DESIGN.md's "why the layer is not one wave-specialised kernel" section.
posttrans -- the bf16x3 and exact-f32 contractions at C3 size, with and without the fused tail.
tower     -- C3-scale tower aggregation (message = x[src] + dst_term[dst]): hand-scheduled kernel vs the compiler-
             scheduled template, bit-identity of the two.
c5        -- the per-GPU shard shape of BASELINE.json configs[4] (V = 2 M, E = 20 M, F = 128) on this one GPU.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXP_LIB = os.path.join(ROOT, "pna_amd", "lib", "libpna_amd_exp.so")
sections = sys.argv[1:] or ["occupancy", "posttrans", "kscan", "tower", "c5"]
if "occupancy" in sections and os.path.exists(EXP_LIB):
    from pna_amd import _lib
    _lib.LIB_PATH = EXP_LIB                       # tools only: the package itself never loads this file
from pna_amd import Graph, ops, functional as PF  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

dev = torch.device("cuda:0")
AGGS = ["mean", "max", "min", "std"]
out = {}


def timeit(fn, iters=10, rounds=3, warm=2):
    for _ in range(warm):
        fn()
    best, ts = None, []
    for _ in range(rounds):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters)
    return {"min_ms": min(ts), "median_ms": sorted(ts)[len(ts) // 2]}


V, E, F = 1_000_000, 10_000_000, 75
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
xb = torch.zeros(V, 80, device=dev)
x = xb[:, :F]
x.copy_(torch.randn(V, F, generator=torch.Generator().manual_seed(1234)))
amp, att = g.degree_scalers(2.2488)

if "occupancy" in sections:
    res = []
    have_exp = os.path.exists(EXP_LIB)
    for lds_kb in ((0, 26, 32, 40, 53) if have_exp else (0,)):
        for U in (4, 6, 8):
            for skip in ((0, 1) if have_exp else (0,)):
                tune = dict(unroll=U, rows_per_group=4, debug=(lds_kb << 8) | skip)
                t = timeit(lambda: ops.segreduce(g.csr.rowptr, g.csr.col, x, F, AGGS, heavy=g.heavy_schedule(), workspace=g.workspace,
                                                 items=g.work_items(), tune=tune))
                blocks = 8 if lds_kb == 0 else min(8, 160 // lds_kb)
                res.append(dict(lds_kb=lds_kb, blocks_per_cu=blocks, waves_per_cu=4 * blocks, unroll=U, skip_stores=skip, **t))
                print("occ", res[-1], flush=True)
    out["occupancy"] = res

if "posttrans" in sections:
    with torch.no_grad():
        agg = PF.aggregate(g, x, F, AGGS)
        W = (torch.randn(F, 12 * F, generator=torch.Generator().manual_seed(1)) / 30).to(dev)
        b = torch.randn(F, generator=torch.Generator().manual_seed(2)).to(dev)
        cs, ct = (torch.rand(F) + 0.5).to(dev), torch.randn(F).to(dev)
        y80 = torch.empty(V, 80, device=dev)[:, :F]
        res = {}
        for pl in (2, 3):
            res[f"bf16x3_pipeline{pl}_plain"] = timeit(lambda: ops.posttrans(agg, 4 * F, W, [None, amp, att], b, arith="bf16x3", pipeline=pl))
            res[f"bf16x3_pipeline{pl}_tail"] = timeit(lambda: ops.posttrans(agg, 4 * F, W, [None, amp, att], b, col_scale=cs, col_shift=ct,
                                                                         relu=True, residual=x, out=y80, arith="bf16x3", pipeline=pl))
            print("posttrans pipeline", pl, res[f"bf16x3_pipeline{pl}_plain"], res[f"bf16x3_pipeline{pl}_tail"], flush=True)
        y2 = ops.posttrans(agg, 4 * F, W, [None, amp, att], b, col_scale=cs, col_shift=ct, relu=True, residual=x, arith="bf16x3", pipeline=2)
        y3 = ops.posttrans(agg, 4 * F, W, [None, amp, att], b, col_scale=cs, col_shift=ct, relu=True, residual=x, arith="bf16x3", pipeline=3)
        res["pipelines_bit_identical"] = bool(torch.equal(y2, y3))
        for arith in ("bf16x3", "f32"):
            res[arith + "_plain"] = timeit(lambda: ops.posttrans(agg, 4 * F, W, [None, amp, att], b, arith=arith))
            res[arith + "_tail_pitch75"] = timeit(lambda: ops.posttrans(agg, 4 * F, W, [None, amp, att], b, col_scale=cs, col_shift=ct,
                                                                        relu=True, residual=x, arith=arith))
            res[arith + "_tail_pitch80"] = timeit(lambda: ops.posttrans(agg, 4 * F, W, [None, amp, att], b, col_scale=cs, col_shift=ct,
                                                                        relu=True, residual=x, out=y80, arith=arith))
            print("posttrans", arith, {k: v for k, v in res.items() if k.startswith(arith)}, flush=True)
        ya = ops.posttrans(agg, 4 * F, W, [None, amp, att], b, col_scale=cs, col_shift=ct, relu=True, residual=x, arith="bf16x3")
        yb = ops.posttrans(agg, 4 * F, W, [None, amp, att], b, col_scale=cs, col_shift=ct, relu=True, residual=x, arith="f32")
        res["max_abs_diff_x3_vs_f32"] = (ya - yb).abs().max().item()
        res["max_abs_y"] = yb.abs().max().item()
    out["posttrans"] = res

if "kscan" in sections:
    # the bf16x3 contraction at constant MFMA work and growing K: the epilogue's share shrinks as 1/K, so the time per unit
    # of work separates the per-chunk cost (MFMA stream + boundary) from the per-tile cost (epilogue)
    res = []
    with torch.no_grad():
        for M, K in ((1_000_000, 300), (468_750, 640), (234_375, 1280), (93_750, 3200)):
            a = torch.randn(M, K, device=dev)
            W = (torch.randn(F, 3 * K, generator=torch.Generator().manual_seed(1)) / 30).to(dev)
            sc = [None, torch.rand(M, device=dev) + 0.5, torch.rand(M, device=dev) + 0.5]
            for pl in (2, 3):
                t = timeit(lambda: ops.posttrans(a, K, W, sc, arith="bf16x3", pipeline=pl))
                chunks = M / 16 * ((K + 31) // 32)
                res.append(dict(M=M, K=K, pipeline=pl, ns_per_rowtile_chunk=t["min_ms"] * 1e6 / chunks * 1024, **t))
                print("kscan", res[-1], flush=True)
            del a, W, sc
    out["kscan"] = res

if "tower" in sections:
    with torch.no_grad():
        db = torch.zeros(V, 80, device=dev)
        d = db[:, :F]
        d.copy_(torch.randn(V, F, generator=torch.Generator().manual_seed(5)))
        res = {}
        res["simple_fast"] = timeit(lambda: PF.aggregate(g, x, F, AGGS))
        res["tower_fast"] = timeit(lambda: PF.aggregate(g, x, F, AGGS, dst_term=d))
        a1 = PF.aggregate(g, x, F, AGGS, dst_term=d)
        a2 = ops.segreduce(g.csr.rowptr, g.csr.col, x, F, AGGS, dst_term=d, heavy=g.heavy_schedule(), workspace=g.workspace,
                           items=g.work_items(), tune=dict(generic=1))
        res["tower_template"] = timeit(lambda: ops.segreduce(g.csr.rowptr, g.csr.col, x, F, AGGS, dst_term=d, heavy=g.heavy_schedule(),
                                                             workspace=g.workspace, items=g.work_items(), tune=dict(generic=1)))
        res["bit_identical_fast_vs_template"] = bool(torch.equal(a1, a2))
        print("tower", res, flush=True)
    out["tower"] = res

if "c5" in sections:
    del xb, x
    torch.cuda.empty_cache()
    V5, E5, F5 = 2_000_000, 20_000_000, 128
    s5, d5 = powerlaw_graph(V5, E5, seed=1234, device=dev)
    g5 = Graph(s5, d5, V5)
    x5 = torch.randn(V5, F5, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))
    amp5, att5 = g5.degree_scalers(2.2488)
    with torch.no_grad():
        res = {}
        res["segreduce"] = timeit(lambda: PF.aggregate(g5, x5, F5, AGGS), iters=5)
        agg5 = PF.aggregate(g5, x5, F5, AGGS)
        W5 = (torch.randn(F5, 12 * F5, generator=torch.Generator().manual_seed(1)) / 40).to(dev)
        b5 = torch.zeros(F5, device=dev)
        for arith in ("bf16x3", "f32"):
            res["posttrans_" + arith] = timeit(lambda: ops.posttrans(agg5, 4 * F5, W5, [None, amp5, att5], b5, arith=arith), iters=5)
        alg = E5 * (4 * F5 + 4) + 4 * (V5 + 1) + V5 * 16 * F5
        res["algorithmic_bytes"] = alg
        res["segreduce_frac_of_8TBs"] = alg / (res["segreduce"]["min_ms"] * 1e-3) / 8e12
        res["read_only_frac"] = (E5 * (4 * F5 + 4) + 4 * (V5 + 1)) / (res["segreduce"]["min_ms"] * 1e-3) / 8e12
        print("c5", res, flush=True)
    out["c5_shard_shape_one_gpu"] = res

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
tag = "_".join(sections)
with open(os.path.join(ROOT, "gpurun_out", f"exp_r02_{tag}.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
