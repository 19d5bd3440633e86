#!/usr/bin/env python
"""hipGraph replay of the C3 step against the eager step, one box: the layer's two halves recorded beside one another
(fork / join: eager mode's arrangement), one behind the other, the small launches first -- and a 4-layer stack the same ways."""
import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, functional as PF                         # noqa: E402
from pna_amd.capture import GraphedForward                          # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer                    # noqa: E402
from pna_amd.synth import powerlaw_graph                            # noqa: E402
dev = torch.device("cuda:0")
V, E, F = 1_000_000, 10_000_000, 75
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
torch.manual_seed(0)
layers = [PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True).to(dev).eval() for _ in range(4)]
h = torch.randn(V, 80, device=dev)[:, :F]


def wall(fn, n=30, reps=4):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / n * 1e3)
    return best


def stack(x, n):
    for l in layers[:n]:
        x = l(g, x)
    return x


out = {}
with torch.no_grad():
    for n in (1, 4):
        rec = {"eager": wall(lambda: stack(h, n))}
        for mode in ("beside", "behind", "rest_first"):
            PF.CAPTURE_OVERLAP = mode
            gf = GraphedForward(lambda x: stack(x, n), h, alias_inputs=True)
            rec["replay_" + mode] = wall(gf.graph.replay)
            ok = torch.equal(gf.static_out, stack(h, n))
            rec["same_bits_" + mode] = bool(ok)
            del gf
        out[f"{n}_layers"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in rec.items()}
        print(n, out[f"{n}_layers"], flush=True)
print("RESULT " + json.dumps(out))
