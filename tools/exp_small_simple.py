#!/usr/bin/env python
"""PNASimpleLayer (hidden 80, eval): the one-call small-batch path (pna_tower_layer_f32) vs the three-kernel path by batch size."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, functional as PF  # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer  # noqa: E402
from pna_amd.synth import molecule_batch  # noqa: E402

dev = torch.device("cuda:0")


def gpu_ms(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


out = []
for n_graphs in (64, 128, 256, 512, 1024, 2048):
    src, dst, sizes = molecule_batch(n_graphs, mean_nodes=25.5, sd_nodes=12, lo=6, hi=222, seed=41, lognormal=True)
    V = int(sum(sizes))
    g = Graph(src, dst, V, sizes).to(dev)
    avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
    layer = PNASimpleLayer(80, 80, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True).to(dev).eval()
    h = torch.randn(V, 80, device=dev)
    with torch.no_grad():
        PF.SMALL_SIMPLE_ROWS = 1 << 30
        assert layer._small_batch_path(g, h)
        y1 = layer(g, h)
        t_small = gpu_ms(lambda: layer(g, h))
        PF.SMALL_SIMPLE_ROWS = 0
        y2 = layer(g, h)
        t_large = gpu_ms(lambda: layer(g, h))
    out.append(dict(graphs=n_graphs, V=V, E=int(src.numel()), one_call_ms=t_small, three_kernel_ms=t_large,
                    max_abs_diff=(y1 - y2).abs().max().item()))
    print(out[-1], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "small_simple.json"), "w"), indent=1)
