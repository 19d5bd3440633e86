import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph
from pna_amd.dgl.pna_layer import PNALayer
from pna_amd.synth import powerlaw_graph
dev = torch.device("cuda:0")
V, E, F = 1_000_000, 10_000_000, 75
s, d = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(s, d, V)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
h = torch.randn(V, F, device=dev)
sn = torch.ones(V, 1, device=dev)
layer = PNALayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True, towers=1, divide_input=False, residual=True).to(dev).eval()
with torch.no_grad():
    for _ in range(5):
        y = layer(g, h, None, sn)
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
