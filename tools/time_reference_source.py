#!/usr/bin/env python
"""CPU timing of the REFERENCE'S OWN SOURCE on the roofline workload at 1/5 scale (SURVEY 8d "CPU baseline"): the unmodified
models/dgl/pna_layer.py PNASimpleLayer over the test-only DGL stand-in (degree-bucketed mailboxes, oracle/dgl_standin.py) on a
power-law graph of V = 200 k, E = 2 M, F = 75.  /root/reference does not exist on the GPU box, so this runs in the BUILD
CONTAINER and its result is committed as profiles/cpu_reference_source.json; bench.py attaches it to its JSON line under
"cpu_baseline_reference_source" with this provenance.  (The C/OpenMP port timed by bench.py on the GPU box's own cores is the
"cpu_baseline" object.)

    python tools/time_reference_source.py
"""
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dgl_standin  # noqa: E402

dgl_standin.install()
sys.path.insert(0, dgl_standin.REFERENCE_ROOT)
from models.dgl.pna_layer import PNASimpleLayer as RefSimple  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

V, E, F = 200_000, 2_000_000, 75
threads = os.cpu_count()
torch.set_num_threads(threads)
src, dst = powerlaw_graph(V, E, seed=1234)
deg = torch.bincount(dst, minlength=V)
avg_log = torch.log(deg.double() + 1).mean().float()
layer = RefSimple(F, F, "mean max min std", "identity amplification attenuation", {"log": avg_log}, 0.0, True, True).eval()
h = torch.randn(V, F, generator=torch.Generator().manual_seed(1234))
g = dgl_standin.StandinGraph(src.numpy(), dst.numpy(), V)
ts = []
with torch.no_grad():
    for _ in range(3):
        t0 = time.perf_counter()
        layer(g, h)
        ts.append(time.perf_counter() - t0)
best = min(ts)
rec = {"value": E / best, "unit": "edges/s", "cores": threads, "kind": "reference",
       "sample": f"models/dgl/pna_layer.py PNASimpleLayer UNMODIFIED over oracle/dgl_standin.py (DGL 0.4.2 stand-in), power-law graph V={V} "
                 f"E={E} F={F} (1/5 of C3), forward, eval, best of 3 ({best:.2f} s), torch {torch.__version__} CPU, {threads} threads",
       "host": f"build container ({platform.processor() or platform.machine()}, {threads} logical CPUs) -- NOT the GPU box: /root/reference "
               "cannot travel there",
       "distinct_in_degrees": int(torch.unique(deg).numel())}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", "cpu_reference_source.json"), "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec))
