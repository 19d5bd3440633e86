#!/usr/bin/env python
"""Diagnostic: the fp16 x 2 one-kernel layer and the two-kernel (bf16 x 3) path against a float64 contraction of the SAME fp32 statistics
(the kernel's agg_out dump), on inputs with a wide dynamic range (tests/test_gpu_fused_degree.py::..._wide_dynamic_range)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pna_amd import Graph, degree_groups as DG, functional as PF
from pna_amd.synth import powerlaw_graph
import test_gpu_fused_degree as T

dev = torch.device("cuda:0")
V, E, F, N = 150_000, 1_200_000, 75, 75
src, dst = powerlaw_graph(V, E, seed=11, device=dev)
g = Graph(src, dst, V)
layer = T._layer(F, N, dev, residual=False, seed=4)
gen = torch.Generator(device=dev).manual_seed(5)
h = T._features(V, F, dev, seed=6)
with torch.no_grad():
    h.mul_(10.0 ** torch.empty(V, 1, device=dev).uniform_(-15, 15, generator=gen))
    h.mul_(10.0 ** torch.empty(V, F, device=dev).uniform_(-6, 0, generator=gen))
    lin = layer.posttrans.fully_connected[0].linear
    lin.weight.mul_((10.0 ** (torch.arange(N, device=dev) % 9 - 4).float())[:, None])
    with T._Knobs(fused=True, small_graphs=True):
        y_f = layer(g, h)
        plan = DG.plan_of(g)
        dump = torch.zeros(plan.NV, 4 * F, device=dev)
        PF.simple_layer_degree_fused(layer, g, h, agg_out=dump)
    with T._Knobs(fused=False, small_graphs=True):
        y_g = layer(g, h)
    live = plan.perm >= 0
    nodes = plan.perm[live].long()
    a = dump[live].double()                                   # [mean | max | min | std] x F, fp32 statistics
    amp, att = g.degree_scalers(2.3)
    bn = layer.batchnorm_h
    W, b = lin.weight.double(), lin.bias.double()
    K = 4 * F
    WD = (W[:, :K][None] + amp[nodes].double()[:, None, None] * W[:, K:2 * K][None] + att[nodes].double()[:, None, None] * W[:, 2 * K:][None]) if False else None
    # per row (memory): z = W_id a + amp W_amp a + att W_att a
    z = a @ W[:, :K].t() + amp[nodes].double()[:, None] * (a @ W[:, K:2 * K].t()) + att[nodes].double()[:, None] * (a @ W[:, 2 * K:].t())
    mass = a.abs() @ W[:, :K].abs().t() + amp[nodes].double().abs()[:, None] * (a.abs() @ W[:, K:2 * K].abs().t()) + att[nodes].double().abs()[:, None] * (a.abs() @ W[:, 2 * K:].abs().t())
    bn_scale = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
    ref = torch.relu((z + b - bn.running_mean.double()) * bn_scale + bn.bias.double())
    floor = mass * bn_scale.abs() + 1e-30
    for name, y in (("one-kernel fp16x2", y_f), ("two-kernel bf16x3", y_g)):
        err = (y[nodes].double() - ref).abs()
        rel = err / (floor + 1e-6 * (1 + ref.abs()) / 1e-7 * 1e-7)     # relative to sum |a||w| (+ the O(1) epilogue floor)
        r2 = err / (floor + 10.0)
        print(f"{name}: max err / (sum|a||w| bn) = {(err / floor).max().item():.3e}   max err / (sum|a||w| bn + 10) = {r2.max().item():.3e}   "
              f"rms = {((err / (floor + 10.0)) ** 2).mean().sqrt().item():.3e}")
    d = (y_f[nodes] - y_g[nodes]).abs().double()
    print("max |fp16x2 - bf16x3| / (sum|a||w| bn + 10) =", (d / (floor + 10.0)).max().item())
