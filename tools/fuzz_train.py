#!/usr/bin/env python
"""Randomised sweep of the plan-order training step (autograd.SimpleLayerPlanFn + the weight-gradient kernels + the in-place packed
pull rows) against the node-order route (AggregateFn + PosttransFn with the library weight gradient) on random graphs / shapes:
output, input gradient, weight and bias gradients.    python tools/fuzz_train.py [seconds] [seed]"""
import os, sys, time, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, autograd as AG, degree_groups as DG, functional as PF
from pna_amd.dgl.pna_layer import PNASimpleLayer
from pna_amd.synth import powerlaw_graph

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
DG.MIN_ROWS, DG.MIN_OUT, PF.SMALL_SIMPLE_ROWS = 1, 1, 0
t0, n_ok, n_skip, worst = time.time(), 0, 0, [0.0, 0.0, 0.0, 0.0]
while time.time() - t0 < budget:
    F = rnd.randint(17, 80)
    N = F if rnd.random() < 0.5 else rnd.randint(4, 80)
    V = rnd.choice([20000, 60000, 150000])
    E = int(V * rnd.choice([4, 8, 14]))
    src, dst = powerlaw_graph(V, E, seed=rnd.randint(0, 10 ** 6), device=dev)
    if rnd.random() < 0.5:
        keep = dst >= rnd.randint(1, 200)
        src, dst = src[keep], dst[keep]
    g = Graph(src, dst, V)
    bn = rnd.random() < 0.5
    seed = rnd.randint(0, 10 ** 6)
    res = {}
    for plan_route in (True, False):
        AG.PLAN_TRAIN, AG.DW_KERNEL = plan_route, plan_route
        torch.manual_seed(seed)
        layer = PNASimpleLayer(F, N, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.1)}, 0.0, bn, F == N).to(dev).train()
        with torch.no_grad():
            for p in layer.parameters():
                if p.dim() == 2:
                    p.copy_(torch.randn_like(p) / p.shape[1] ** 0.5)
        h = torch.randn(V, F, device=dev, generator=torch.Generator(device=dev).manual_seed(seed)).requires_grad_(True)
        if plan_route and not AG.simple_layer_plan_applies(layer, g, h):
            break
        out = layer(g, h)
        (out * torch.linspace(0.5, 1.5, N, device=dev)).sum().backward()
        lin = layer.posttrans.fully_connected[0].linear
        res[plan_route] = (out.detach(), h.grad, lin.weight.grad, lin.bias.grad)
    AG.PLAN_TRAIN = AG.DW_KERNEL = True
    if len(res) < 2:
        n_skip += 1
        continue
    rel = lambda a, b: (a - b).abs().max().item() / max(1e-30, b.abs().max().item())
    (o1, gh1, gw1, gb1), (o0, gh0, gw0, gb0) = res[True], res[False]
    e_o, e_w = rel(o1, o0), rel(gw1, gw0)
    off = ((gh1 - gh0).abs() > 1e-4 * gh0.abs().max()).float().mean().item()       # (ReLU flips at pre-activations within rounding of 0)
    e_b = 0.0 if bn else rel(gb1, gb0)                    # (a bias in front of batch-statistics BatchNorm: true gradient 0, rounding noise)
    assert torch.isfinite(o1).all() and torch.isfinite(gh1).all() and torch.isfinite(gw1).all()
    # (with BatchNorm the weight gradient is a small difference of large sums -- the component along the batch mean / scale cancels --
    #  and every ReLU flip moves it: 5e-2 of its largest entry there, 1e-2 without)
    assert e_o <= 3e-6 and off <= 2e-3 and e_w <= (5e-2 if bn else 1e-2) and e_b <= 1e-2, ("differs", V, src.numel(), F, N, bn, e_o, off, e_w, e_b)
    worst = [max(a, b) for a, b in zip(worst, (e_o, off, e_w, e_b))]
    n_ok += 1
    print(f"ok V={V} E={src.numel()} F={F} N={N} bn={int(bn)} out={e_o:.1e} gh_off={off:.1e} gw={e_w:.1e} gb={e_b:.1e}", flush=True)
print(f"SUMMARY {n_ok} cases passed, {n_skip} skipped (plan route did not apply); worst: output {worst[0]:.2e} of max, input-gradient entries off by > 1e-4 of max {worst[1]:.2e}, weight gradient {worst[2]:.2e}, bias gradient {worst[3]:.2e}; {time.time() - t0:.0f} s")
