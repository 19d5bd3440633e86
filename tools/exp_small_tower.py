#!/usr/bin/env python
"""The tower layer of molecule batches: pna_tower_layer_f32 (one C call, two launches) against the large-graph kernels (four
launches), eager and under hipGraph replay, over batch sizes -- where the row limit PF.SMALL_TOWER_ROWS should sit.

    python tools/exp_small_tower.py > gpurun_out/small_tower.json
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, functional as PF  # noqa: E402
from pna_amd.capture import GraphedForward  # noqa: E402
from pna_amd.dgl.pna_layer import PNALayer  # noqa: E402
from pna_amd.synth import molecule_batch  # noqa: E402

dev = torch.device("cuda:0")
AGG, SCA = "mean max min std", "identity amplification attenuation"


def gpu_ms(fn, iters=50, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


out = []
for graphs in ([int(a) for a in sys.argv[1:]] or [128, 512, 2048, 8192, 32768]):
    src, dst, sizes = molecule_batch(graphs, seed=41)
    V, E = sum(sizes), src.numel()
    g = Graph(src, dst, V, sizes).to(dev)
    avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
    for name, (fi, fo, div) in (("zinc_mid", (75, 75, False)), ("zinc_last", (75, 70, True))):
        layer = PNALayer(fi, fo, AGG, SCA, avg, 0.0, True, True, towers=5, divide_input=div, residual=True).eval().to(dev)
        h = torch.randn(V, fi, device=dev)
        sn = g.snorm_n()
        rec = dict(case=name, graphs=graphs, V=V, E=E)
        with torch.no_grad():
            for path, limit in (("small", 1 << 30), ("large", 0)):
                PF.SMALL_TOWER_ROWS = limit
                rec[path + "_eager_ms"] = gpu_ms(lambda: layer(g, h, None, sn), iters=30)
                gf = GraphedForward(lambda x: layer(g, x, None, sn), h)
                rec[path + "_hipgraph_ms"] = gpu_ms(lambda: gf(h), iters=30)
                y = gf(h).clone()
                rec[path + "_absmax"] = y.abs().max().item()
                if path == "small":
                    y_small = y
            rec["max_abs_diff"] = (y - y_small).abs().max().item()
        out.append(rec)
        print(rec, file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
