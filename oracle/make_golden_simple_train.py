#!/usr/bin/env python
"""Golden vectors of ONE TRAINING STEP of the reference's PNASimpleLayer (models/dgl/pna_layer.py:151-216 in train mode: batch
statistics in `batchnorm_h`, F.relu, the residual) produced by the reference's own source over oracle/dgl_standin.py.
TEST INFRASTRUCTURE ONLY.  Per fixture: the layer's state before the step, the graph, the input, the output, the gradient of
`(out * R).sum()` w.r.t. the input and every parameter, and the BatchNorm running statistics after the step.

Run in the build container only (the GPU box has no /root/reference):
    python oracle/make_golden_simple_train.py        # writes tests/golden/simple_train_*.npz
The committed fixtures were produced with torch 2.10.0 (CPU path, one thread), seeds below.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dgl_standin  # noqa: E402
from oracle.make_golden import AGG4, SCA3, RefDGLLayer, RefSimpleLayer, molecule_batch, powerlaw_graph, randomise, save  # noqa: E402


def golden_simple_train(name, seed, N, E, F, out_dim, scalers=SCA3, residual=True):
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    src, dst = powerlaw_graph(rng, N, E)
    deg = np.bincount(dst, minlength=N)
    avg_log = torch.tensor(float(np.mean(np.log(deg + 1))), dtype=torch.float32)
    torch.manual_seed(seed)
    layer = RefSimpleLayer(F, out_dim, AGG4, scalers, {"log": avg_log}, 0.0, True, residual).train()
    randomise(layer, gen)
    before = {k: v.clone() for k, v in layer.state_dict().items()}
    h = (torch.randn(N, F, generator=gen) * 1.5 + 0.3).requires_grad_(True)
    R = torch.randn(N, out_dim, generator=gen)
    g = dgl_standin.StandinGraph(src, dst, N)
    out = layer(g, h)
    (out * R).sum().backward()
    arrays = dict(src=src, dst=dst, h=h.detach(), avg_log=avg_log, R=R, out=out.detach(), grad_h=h.grad,
                  running_mean_after=layer.batchnorm_h.running_mean.clone(), running_var_after=layer.batchnorm_h.running_var.clone())
    for k, p in layer.named_parameters():
        arrays["grad/" + k] = p.grad.clone()
    meta = dict(kind="dgl_simple_train", seed=seed, N=N, F=F, out_dim=out_dim, aggregators=AGG4, scalers=scalers, residual=residual,
                batch_norm=True, posttrans_layers=1)
    layer.load_state_dict(before)                            # the fixture stores the state BEFORE the step
    save(name, meta, arrays, layer)


def golden_tower_train(name, seed, in_dim, out_dim, towers, divide_input, edge_dim=0, n_graphs=8):
    """The same for PNALayer with towers (models/dgl/pna_layer.py:130-148 over :55-76, train mode: per-tower batch-statistics
    BatchNorm behind the graph norm, LeakyReLU mixing network, residual), on a batch of molecule-shaped graphs."""
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    src, dst, sizes = molecule_batch(rng, n_graphs)
    N = int(sum(sizes))
    deg = np.bincount(dst, minlength=N)
    avg_log = torch.tensor(float(np.mean(np.log(deg + 1))), dtype=torch.float32)
    layer = RefDGLLayer(in_dim, out_dim, AGG4, SCA3, {"log": avg_log}, 0.0, True, True, towers=towers, pretrans_layers=1,
                        posttrans_layers=1, divide_input=divide_input, residual=True, edge_features=edge_dim > 0, edge_dim=edge_dim).train()
    randomise(layer, gen)
    before = {k: v.clone() for k, v in layer.state_dict().items()}
    h = (torch.randn(N, in_dim, generator=gen) * 1.2).requires_grad_(True)
    e = torch.randn(src.size, edge_dim, generator=gen).requires_grad_(edge_dim > 0) if edge_dim > 0 else torch.zeros(src.size, 0)
    R = torch.randn(N, out_dim, generator=gen)
    snorm_n = torch.cat([torch.full((s, 1), 1.0 / s) for s in sizes]).sqrt()
    g = dgl_standin.StandinGraph(src, dst, N)
    out = layer(g, h, e if edge_dim > 0 else None, snorm_n)
    (out * R).sum().backward()
    arrays = dict(src=src, dst=dst, h=h.detach(), e=e.detach(), snorm_n=snorm_n, avg_log=avg_log, R=R, out=out.detach(), grad_h=h.grad.clone())
    if edge_dim > 0:
        arrays["grad_e"] = e.grad.clone()
    for k, p in layer.named_parameters():
        arrays["grad/" + k] = p.grad.clone()
    for k, b in layer.named_buffers():
        if "running" in k:
            arrays["after/" + k] = b.clone()
    meta = dict(kind="dgl_tower_train", seed=seed, N=N, in_dim=in_dim, out_dim=out_dim, towers=towers, divide_input=divide_input,
                edge_dim=edge_dim, pretrans_layers=1, posttrans_layers=1, graph_norm=True, batch_norm=True, residual=True,
                aggregators=AGG4, scalers=SCA3, sizes=sizes)
    layer.load_state_dict(before)
    save(name, meta, arrays, layer)


def main():
    torch.set_num_threads(1)
    golden_tower_train("tower_train_t4_div", 61, in_dim=24, out_dim=24, towers=4, divide_input=True)
    golden_tower_train("tower_train_t3_edgefeat", 62, in_dim=18, out_dim=18, towers=3, divide_input=False, edge_dim=5)
    golden_simple_train("simple_train_f75", 75, N=400, E=3600, F=75, out_dim=75)
    golden_simple_train("simple_train_f20", 20, N=250, E=1500, F=20, out_dim=12, scalers="identity amplification", residual=False)


if __name__ == "__main__":
    main()
