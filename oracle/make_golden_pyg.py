#!/usr/bin/env python
"""Golden vectors for the PyG formulation: runs the REFERENCE's `models/pytorch_geometric/pna.py` (unmodified, over
oracle/pyg_standin.py) on CPU.  TEST INFRASTRUCTURE ONLY; build container only (the GPU box has no /root/reference).
    python oracle/make_golden_pyg.py        # writes tests/golden/pyg_*.npz
Produced with torch 2.10.0 (CPU), seeds below.  The graphs deliberately contain nodes WITHOUT in-edges: the PyG
variant is the one place where the reference defines them (scalers.py:16-19,:26-29).
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyg_standin  # noqa: E402

pyg_standin.install()
from models.pytorch_geometric.pna import PNAConv as RefConv, PNAConvSimple as RefConvSimple  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def graph(rng, N, E, isolated):
    """Random multigraph; the last `isolated` nodes receive no edge."""
    w = (np.arange(N) + 1.0) ** -0.6
    w /= w.sum()
    src = rng.choice(N, size=E, p=w)
    dst = rng.choice(N - isolated, size=E)
    return torch.from_numpy(np.stack([src, dst]).astype(np.int64))


def randomise(module, gen):
    with torch.no_grad():
        for _, p in module.named_parameters():
            if p.dim() == 2:
                p.copy_(torch.randn(p.shape, generator=gen) / p.shape[1] ** 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.3)


def save(name, meta, arrays, module):
    sd = {"sd/" + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}
    arrays = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, meta=json.dumps(meta), **arrays, **sd)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  out={arrays['out'].shape}")


def hist(edge_index, N):
    d = np.bincount(edge_index[1].numpy(), minlength=N)
    return torch.from_numpy(np.bincount(d)).long()


def golden_simple(name, seed, N, E, F, out, aggregators, scalers, post_layers=1, isolated=3):
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    ei = graph(rng, N, E, isolated)
    deg_hist = hist(ei, N)
    layer = RefConvSimple(F, out, list(aggregators), list(scalers), deg_hist, post_layers=post_layers).eval()
    randomise(layer, gen)
    x = torch.randn(N, F, generator=gen)
    with torch.no_grad():
        y = layer(x, ei)
        agg = layer.propagate(ei, x=x, size=None)                  # the (V, A*S*F) tensor of aggregate()
    meta = dict(kind="pyg_simple", seed=seed, N=N, F=F, out=out, aggregators=list(aggregators), scalers=list(scalers),
                post_layers=post_layers)
    save(name, meta, dict(edge_index=ei, deg_hist=deg_hist, x=x, agg=agg, out=y), layer)


def golden_conv(name, seed, N, E, in_c, out_c, aggregators, scalers, towers, divide_input, edge_dim=None, pre_layers=1,
                post_layers=1, isolated=2):
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    ei = graph(rng, N, E, isolated)
    deg_hist = hist(ei, N)
    layer = RefConv(in_c, out_c, list(aggregators), list(scalers), deg_hist, edge_dim=edge_dim, towers=towers,
                    pre_layers=pre_layers, post_layers=post_layers, divide_input=divide_input).eval()
    randomise(layer, gen)
    x = torch.randn(N, in_c, generator=gen)
    ea = torch.randn(E, edge_dim, generator=gen) if edge_dim else None
    with torch.no_grad():
        y = layer(x, ei, ea)
    meta = dict(kind="pyg_conv", seed=seed, N=N, in_c=in_c, out_c=out_c, aggregators=list(aggregators), scalers=list(scalers),
                towers=towers, divide_input=divide_input, edge_dim=edge_dim or 0, pre_layers=pre_layers, post_layers=post_layers)
    save(name, meta, dict(edge_index=ei, deg_hist=deg_hist, x=x, edge_attr=ea if ea is not None else torch.zeros(E, 0), out=y), layer)


def main():
    torch.set_num_threads(1)
    A4, S3 = ("mean", "min", "max", "std"), ("identity", "amplification", "attenuation")
    golden_simple("pyg_simple_hiv", 11, N=80, E=400, F=80, out=80, aggregators=A4, scalers=S3)        # example.py:31-33
    golden_simple("pyg_simple_all", 12, N=60, E=300, F=20, out=12, aggregators=("sum", "mean", "min", "max", "var", "std"),
                  scalers=("identity", "amplification", "attenuation", "linear", "inverse_linear"), post_layers=2)
    golden_conv("pyg_conv_towers", 13, N=70, E=350, in_c=30, out_c=25, aggregators=A4, scalers=S3, towers=5, divide_input=False)
    golden_conv("pyg_conv_edge_divide", 14, N=50, E=260, in_c=24, out_c=24, aggregators=A4, scalers=S3, towers=4,
                divide_input=True, edge_dim=6)
    golden_conv("pyg_conv_deep", 15, N=40, E=200, in_c=16, out_c=16, aggregators=("mean", "max", "var"),
                scalers=("identity", "linear"), towers=2, divide_input=False, pre_layers=2, post_layers=2)


if __name__ == "__main__":
    main()
