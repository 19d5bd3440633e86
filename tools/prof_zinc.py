import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph
from pna_amd.dgl.pna_layer import PNALayer
from pna_amd.synth import molecule_batch
dev = torch.device("cuda:0")
src, dst, sizes = molecule_batch(128, seed=41)
V = sum(sizes)
g = Graph(src, dst, V, sizes).to(dev)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
h = torch.randn(V, 75, device=dev)
layer = PNALayer(75, 75, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True, towers=5, divide_input=False, residual=True).to(dev).eval()
sn = g.snorm_n()
with torch.no_grad():
    for _ in range(20):
        y = layer(g, h, None, sn)
torch.cuda.synchronize()
print("ok")
