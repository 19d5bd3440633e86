#!/usr/bin/env python
"""Development: where does the C1 loss trace of tests/test_gpu_train_trace.py drift?  Net A = pna_amd dense layer (HIP forward /
backward); net B = the same modules, but every conv forward goes through oracle/torch_oracle.py's plain torch ops ON THE GPU
(autograd backward).  Step-1 gradients A vs B per parameter, then 8 Adam steps of each against the CPU trace of the reference."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
import test_gpu_train_trace as T
from oracle import torch_oracle as TO
from pna_amd.pytorch.pna.layer import PNALayer
dev = torch.device("cuda:0")
meta, a, sd = load_golden("c1_train_trace_n15")
avg_d = {k: a["avg_" + k].to(dev) for k in ("lin", "log", "exp")}


class OracleLayer(PNALayer):
    def forward(self, x, adj):
        return TO.dense_layer_forward(dict(self.named_parameters()), x, adj, meta["aggregators"], meta["scalers"], avg_d, meta["towers"], self._div)


def make(layer_type):
    torch.manual_seed(0)
    net = T._GNN(layer_type, meta, avg_d, dev)
    net.load_state_dict(sd, strict=True)
    return net.to(dev)


A, Bn = make(PNALayer), make(OracleLayer)
Bn.conv_layers[0]._div, Bn.conv_layers[1]._div = False, True
Bsz = meta["B"]
adj, x, nl, gl = (a[k].to(dev).split(Bsz) for k in ("adj", "x", "node_labels", "graph_labels"))
for net in (A, Bn):
    net.train(); net.zero_grad()
    T._total_loss(net(x[0], adj[0]), (nl[0], gl[0])).backward()
worst = []
for (n, p), (_, q) in zip(A.named_parameters(), Bn.named_parameters()):
    if p.grad is None or q.grad is None:
        print("no grad", n); continue
    d = (p.grad - q.grad).abs().max().item(); s = q.grad.abs().max().item()
    worst.append((d / max(s, 1e-30), n, d, s))
for r in sorted(worst, reverse=True)[:12]:
    print("grad rel diff %.2e  %-60s abs %.2e of max %.2e" % r)
want = a["out"].double().tolist()


def make_hybrid(layer_type):
    torch.manual_seed(0)
    torch.set_num_threads(1)
    net = T._GNN(layer_type, meta, avg_d, dev)
    net.load_state_dict(sd, strict=True)
    net.conv_layers.to(dev)
    return net


for name, net in (("pna_amd, everything on the GPU", make(PNALayer)), ("torch-oracle, everything on the GPU", make(OracleLayer)),
                  ("pna_amd layers on the GPU, assembly on the CPU", make_hybrid(PNALayer)),
                  ("torch-oracle layers on the GPU, assembly on the CPU", make_hybrid(OracleLayer))):
    hybrid = "assembly" in name
    if hybrid:
        adj, x, nl, gl = (a[k].split(Bsz) for k in ("adj", "x", "node_labels", "graph_labels"))
    if name.startswith("torch"):
        net.conv_layers[0]._div, net.conv_layers[1]._div = False, True
    opt = torch.optim.Adam(net.parameters(), lr=meta["lr"], weight_decay=meta["weight_decay"])
    got = []
    for ep in range(2):
        net.train()
        for b in range(4):
            opt.zero_grad(); loss = T._total_loss(net(x[b], adj[b]), (nl[b], gl[b])); loss.backward(); opt.step(); got.append(loss.item())
    print(name, "rel deviation per step:", " ".join("%.1e" % (abs(g - w) / abs(w)) for g, w in zip(got, want)))
