#!/usr/bin/env python
"""Forward + backward of one PNASimpleLayer on the roofline workload (C3) -- not a bench.py line (the metric is the
forward layer); documents what the training path costs.  Prints one JSON object."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph  # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

dev = torch.device("cuda:0")
V, E, F = 1_000_000, 10_000_000, 75
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
avg_log = float(torch.log(g.in_degrees().float() + 1).mean())
layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", {"log": avg_log}, 0.0, True, True).to(dev).train()
h = torch.randn(V, F, device=dev, requires_grad=True)


def step():
    out = layer(g, h)
    out.sum().backward()


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def fwd_only():
    with torch.no_grad():
        layer(g, h)


res = {"workload": "C3 PNASimpleLayer, train mode (batch-stat BatchNorm), V=1M E=10M F=75",
       "fwd_bwd_ms": timed(step), "fwd_train_mode_nograd_ms": timed(fwd_only)}
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda r: -r.device_time_total)[:12]
res["top_kernels_us"] = {r.key[:70]: round(r.device_time_total, 1) for r in rows}
print(json.dumps(res, indent=1))
