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
from oracle.make_golden import AGG4, SCA3, RefSimpleLayer, powerlaw_graph, randomise, save  # noqa: E402


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


def main():
    torch.set_num_threads(1)
    golden_simple_train("simple_train_f75", 75, N=400, E=3600, F=75, out_dim=75)
    golden_simple_train("simple_train_f20", 20, N=250, E=1500, F=20, out_dim=12, scalers="identity amplification", residual=False)


if __name__ == "__main__":
    main()
