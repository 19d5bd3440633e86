#!/usr/bin/env python
"""Many repeats of the one-call small-batch layers (gather + MFMA in one kernel): identical bits every run?  (development check
after the packed-fold finding of DESIGN.md 4.7 point 7)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, functional as PF  # noqa: E402
from pna_amd.dgl.pna_layer import PNALayer, PNASimpleLayer  # noqa: E402
from pna_amd.synth import molecule_batch  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
tower = PNALayer(75, 75, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.0)}, 0.0, True, True,
                 towers=5, divide_input=False, residual=True).to(dev).eval()
simple = PNASimpleLayer(75, 75, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.0)}, 0.0, True, True).to(dev).eval()
bad = 0
with torch.no_grad():
    for n_graphs, layer, name, limit in ((int(PF.SMALL_TOWER_ROWS / 24), tower, "tower", PF.SMALL_TOWER_ROWS), (160, simple, "simple", PF.SMALL_SIMPLE_ROWS)):
        s, d, sizes = molecule_batch(n_graphs, seed=5)
        g = Graph(s, d, int(sum(sizes)), sizes).to(dev)
        assert g.num_nodes <= limit, (g.num_nodes, limit)
        h = torch.randn(g.num_nodes, 75, device=dev)
        sn = torch.rand(g.num_nodes, 1, device=dev) + 0.5
        fn = (lambda: layer(g, h, None, sn)) if name == "tower" else (lambda: layer(g, h))
        y0 = fn().clone()
        diff = sum(0 if torch.equal(fn(), y0) else 1 for _ in range(200))
        bad += diff
        print(f"{name}: {g.num_nodes} rows, 200 repeats, {diff} differ", flush=True)
print("DETERMINISTIC" if bad == 0 else "NOT DETERMINISTIC", flush=True)
