#!/usr/bin/env python
"""Same-box A/B of the training step's pull: per-edge rows (round 6, default) against the ranked pull of rounds 3-5 -- step time at C3 and
bit-equality of every gradient."""
import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, autograd as AG                        # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer                 # noqa: E402
from pna_amd.synth import powerlaw_graph                         # noqa: E402
dev = torch.device("cuda:0")
V, E, F = 1_000_000, 10_000_000, 75
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
avg_log = float(torch.log(g.in_degrees().float() + 1).mean())
torch.manual_seed(0)
layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", {"log": avg_log}, 0.0, True, True).to(dev).train()
h = torch.randn(V, F, device=dev, requires_grad=True)
R = torch.randn(V, F, device=dev)


def step():
    h.grad = None
    layer.zero_grad(set_to_none=True)
    (layer(g, h) * R).sum().backward()


def timed(n=20, warmup=12, repeats=2):
    for _ in range(warmup):
        step()
    best = 1e9
    for _ in range(repeats):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / n * 1e3)
    return best


out, grads = {}, {}
for rnd in range(2):
    for mode in (True, False):
        AG.PULL_EDGE_ROWS = mode
        name = "edge_rows" if mode else "ranked"
        out.setdefault(name, []).append(round(timed(), 4))
        step()
        grads[name] = [h.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
# hub SOURCE rows (out-degree above the heavy threshold) are summed from segments with atomics in both forms: their order varies from run to run
gT = g._pna_amd_transposed
hs = gT.heavy_schedule()
light = torch.ones(V, dtype=torch.bool, device=dev)
if hs.n_heavy > 0:
    light[hs.heavy_rows.long()] = False
step()
again = h.grad.clone()
d = (grads["edge_rows"][0] - grads["ranked"][0]).abs()
print(json.dumps({"fwd_bwd_ms": out, "grad_h_bit_identical_on_rows_without_atomics": bool(torch.equal(grads["edge_rows"][0][light], grads["ranked"][0][light])),
                  "hub_source_rows": int((~light).sum()), "max_abs_diff_grad_h_hub_rows": d[~light].max().item() if hs.n_heavy > 0 else 0.0,
                  "same_mode_repeat_differs_on_hub_rows": bool(not torch.equal(again[~light], grads["ranked"][0][~light])),
                  "same_mode_repeat_identical_elsewhere": bool(torch.equal(again[light], grads["ranked"][0][light])),
                  "weight_gradients_max_rel_diff": max(((a - b).abs().max() / b.abs().max().clamp(min=1e-30)).item() for a, b in zip(grads["edge_rows"][1:], grads["ranked"][1:]))}))
