#!/usr/bin/env python
"""Golden vectors for the degree-grouped contraction, produced by the REFERENCE's own source.  TEST INFRASTRUCTURE ONLY.

The fixtures of make_golden.py are molecule-sized: no in-degree value has the 128 rows a degree tile needs, so through the
grouped path they only reach its rest list.  These graphs have a few thousand nodes (hundreds of rows per frequent degree,
rare degrees and a few hubs as well); `models/dgl/pna_layer.py::PNASimpleLayer` runs unmodified over oracle/dgl_standin.py.
Stored: graph, features, parameters and the layer OUTPUT (the (V, 12F) reduce tensor would be megabytes; the small fixtures
pin it).  Kind "dgl_simple_groups".

Run in the build container only (the GPU box has no /root/reference):
    python oracle/make_golden_degree_groups.py            # writes tests/golden/groups_*.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import make_golden as MG  # noqa: E402  (installs the DGL stand-in, imports the reference layers)
from oracle import dgl_standin  # noqa: E402


def golden_groups(name, seed, N, E, F, out_dim, scalers=MG.SCA3, residual=True, hubs=3):
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    src, dst = MG.powerlaw_graph(rng, N, E)
    # a few hub rows far above every other in-degree (the hub rows of the gather's heavy path at full size)
    hub_dst = np.repeat(rng.choice(N, size=hubs, replace=False), 300)
    src = np.concatenate([src, rng.integers(0, N, size=hub_dst.size)]).astype(np.int64)
    dst = np.concatenate([dst, hub_dst]).astype(np.int64)
    deg = np.bincount(dst, minlength=N)
    avg_log = torch.tensor(float(np.mean(np.log(deg + 1))), dtype=torch.float32)
    torch.manual_seed(seed)
    layer = MG.RefSimpleLayer(F, out_dim, MG.AGG4, scalers, {"log": avg_log}, 0.0, True, residual, posttrans_layers=1).eval()
    MG.randomise(layer, gen)
    h = torch.randn(N, F, generator=gen)
    g = dgl_standin.StandinGraph(src, dst, N)
    with torch.no_grad():
        out = layer(g, h)
    cnt = np.bincount(deg)
    meta = dict(kind="dgl_simple_groups", seed=seed, N=N, F=F, out_dim=out_dim, aggregators=MG.AGG4, scalers=scalers,
                residual=residual, batch_norm=True, posttrans_layers=1, degrees_with_128_rows=int((cnt >= 128).sum()),
                max_in_degree=int(deg.max()))
    MG.save(name, meta, dict(src=src.astype(np.int32), dst=dst.astype(np.int32), h=h, avg_log=avg_log, out=out), layer)


def golden_tower_groups(name, seed, N, E, in_dim, out_dim, towers, divide_input, graph_norm=True, residual=True, hubs=3):
    """The reference's tower layer (models/dgl/pna_layer.py::PNALayer: pretrans per edge, reduce_func, posttrans, graph norm,
    BatchNorm, mixing network) on a graph with degree tiles.  Kind "dgl_tower_groups"."""
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    src, dst = MG.powerlaw_graph(rng, N, E)
    hub_dst = np.repeat(rng.choice(N, size=hubs, replace=False), 300)
    src = np.concatenate([src, rng.integers(0, N, size=hub_dst.size)]).astype(np.int64)
    dst = np.concatenate([dst, hub_dst]).astype(np.int64)
    deg = np.bincount(dst, minlength=N)
    avg_log = torch.tensor(float(np.mean(np.log(deg + 1))), dtype=torch.float32)
    torch.manual_seed(seed)
    layer = MG.RefDGLLayer(in_dim, out_dim, MG.AGG4, MG.SCA3, {"log": avg_log}, 0.0, graph_norm, True, towers=towers, pretrans_layers=1,
                           posttrans_layers=1, divide_input=divide_input, residual=residual, edge_features=False, edge_dim=0).eval()
    MG.randomise(layer, gen)
    h = torch.randn(N, in_dim, generator=gen)
    sizes = [N // 3, N - N // 3]                                   # two member graphs (graph norm: 1 / sqrt(nodes of the node's graph))
    snorm_n = torch.cat([torch.full((s_, 1), 1.0 / s_) for s_ in sizes]).sqrt()
    g = dgl_standin.StandinGraph(src, dst, N)
    with torch.no_grad():
        out = layer(g, h, None, snorm_n)
    cnt = np.bincount(deg)
    meta = dict(kind="dgl_tower_groups", seed=seed, N=N, in_dim=in_dim, out_dim=out_dim, towers=towers, divide_input=divide_input, edge_dim=0,
                pretrans_layers=1, posttrans_layers=1, graph_norm=graph_norm, batch_norm=True, residual=residual, aggregators=MG.AGG4,
                scalers=MG.SCA3, sizes=sizes, degrees_with_128_rows=int((cnt >= 128).sum()), max_in_degree=int(deg.max()))
    MG.save(name, meta, dict(src=src.astype(np.int32), dst=dst.astype(np.int32), h=h, snorm_n=snorm_n, avg_log=avg_log, out=out), layer)


def main():
    os.makedirs(MG.OUT, exist_ok=True)
    torch.set_num_threads(1)
    golden_groups("groups_f44", 11, N=2600, E=9000, F=44, out_dim=44)                                  # 80-column block, 3 scalers
    golden_groups("groups_f96_two_scalers", 12, N=1500, E=5000, F=96, out_dim=96, scalers="identity amplification")   # 128-column block
    # round 3: the TIMED shape (F = 75, three scalers: BASELINE configs[2]; the one-kernel path's 2 full + half feature
    # blocks) and the C5 shape (out_dim = 128, three scalers: the 128-column grouped block), plus a width whose last block is full
    golden_groups("groups_f75", 13, N=2400, E=8500, F=75, out_dim=75)
    golden_groups("groups_f128", 14, N=1400, E=4800, F=128, out_dim=128)
    golden_groups("groups_f64_n72", 15, N=2000, E=7000, F=64, out_dim=72, residual=False)
    # round 3: the tower layer through the degree-grouped contraction -- the SURVEY 8d C3 (iii) shape (one tower,
    # F = 75), the ZINC shape (5 towers, hidden 75, inputs not divided) and a divided-input layer without graph norm
    golden_tower_groups("tower_groups_t1_f75", 16, N=2200, E=8000, in_dim=75, out_dim=75, towers=1, divide_input=False)
    golden_tower_groups("tower_groups_t5_f75", 17, N=1500, E=5200, in_dim=75, out_dim=75, towers=5, divide_input=False)
    golden_tower_groups("tower_groups_t4_div", 18, N=1800, E=6500, in_dim=64, out_dim=64, towers=4, divide_input=True, graph_norm=False)


if __name__ == "__main__":
    main()
