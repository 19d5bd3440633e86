#!/usr/bin/env python
"""Golden 2-epoch LOSS TRACE of the multitask benchmark's training loop, produced by the REFERENCE's own source.
TEST INFRASTRUCTURE ONLY.

What runs here is the reference, unmodified, on CPU: `models/pytorch/gnn_framework.py::GNN` (the README model: hidden 16,
4 towers, fixed + variable N/2 iterations, shared GRU, Set2Set readout; 8350 parameters) over `models/pytorch/pna/layer.py`,
the data and labels of `multitask_dataset.py` (generate_graph(RANDOM) with its seed scheme; eccentricity | graph_laplacian_features
| sssp node labels, is_connected | diameter | spectral_radius graph labels through `graph_algorithms.py`), the loss of
`util/util.py::total_loss('mse')` and the training step of `util/train.py:143-149` (zero_grad, forward, loss, backward,
Adam(lr=0.003, weight_decay=1e-6: the README's flags) step), seed 42 (util/train.py:32,74-77).  A 64-graph slice of the train
split (N = 15: the first 64 graphs of the seed-1234 sequence) as 4 batches of 16, two epochs = 8 steps.

Stored: the graphs, features, labels, avg_d, the INITIAL state_dict and the per-step training loss; the GPU test
(tests/test_gpu_train_trace.py) rebuilds the network around pna_amd.pytorch.pna.layer.PNALayer, loads the state_dict strictly and
replays the eight steps.  Kind "c1_train_trace".

    python oracle/make_golden_c1_train.py            # build container only (the GPU box has no /root/reference)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import make_golden_c1_hiv as C1  # noqa: E402  (installs the stand-ins, imports the reference modules)
from oracle.make_golden import save  # noqa: E402
from multitask_benchmark.datasets_generation import graph_algorithms  # noqa: E402
from multitask_benchmark.datasets_generation.graph_generation import GraphType, generate_graph  # noqa: E402
from multitask_benchmark.util.util import total_loss  # noqa: E402

NODE_LABELS = ["eccentricity", "graph_laplacian_features"]          # + sssp (multitask_dataset.py:97-98,:107-111)
GRAPH_LABELS = ["is_connected", "diameter", "spectral_radius"]


def reference_dataset(n_graphs, N, seed):
    """multitask_dataset.py:42-68 for one batch size / node count: adj, features [one-hot source | value], node and graph labels."""
    from inspect import signature
    nla = [getattr(graph_algorithms, s) for s in NODE_LABELS]
    gla = [getattr(graph_algorithms, s) for s in GRAPH_LABELS]
    adjs, feats, nls, gls = [], [], [], []
    for _ in range(n_graphs):
        seed += 1
        adj, features, gtype = generate_graph(N, GraphType.RANDOM, seed=seed)
        while np.min(np.max(adj, 0)) == 0.0:
            seed += 1
            adj, features, _ = generate_graph(N, gtype, seed=seed)
        source = np.random.randint(0, N)
        labels = [graph_algorithms.all_pairs_shortest_paths(adj, 0)[source]]
        for f in nla:
            labels.append(f(adj, features) if "F" in signature(f).parameters else f(adj))
        nls.append(np.swapaxes(np.stack(labels), 0, 1))
        gls.append(np.asarray([f(adj, features) if "F" in signature(f).parameters else f(adj) for f in gla]).flatten())
        onehot = np.zeros(N)
        onehot[source] = 1
        feats.append(np.stack([onehot, features], axis=1))
        adjs.append(adj)
    t = lambda xs: torch.from_numpy(np.asarray(xs)).float()   # noqa: E731
    return t(adjs), t(feats), t(nls), t(gls)


def main():
    torch.set_num_threads(1)
    adj, x, nl, gl = reference_dataset(64, 15, 1234)
    # (same graphs as make_golden_c1_hiv.py's train slice: the seed scheme is the dataset's)
    adj_b, x_b, nl_b, gl_b = (list(t.split(16)) for t in (adj, x, nl, gl))
    avg_d = C1.avg_d_of(adj_b)
    np.random.seed(42)
    torch.manual_seed(42)
    scalers = ["identity", "amplification", "attenuation"]
    conv = dict(aggregators=C1.AGG, scalers=scalers, avg_d=avg_d, towers=4, self_loop=False, pretrans_layers=1, posttrans_layers=1)
    gnn = C1.RefGNN(nfeat=2, nhid=16, nodes_out=3, graph_out=3, dropout=0.0, conv_layers=lambda a: a.shape[1] // 2, fc_layers=3,
                    first_conv_descr=dict(layer_type=C1.RefDenseLayer, args=dict(conv, divide_input=False)),
                    middle_conv_descr=dict(layer_type=C1.RefDenseLayer, args=dict(conv, divide_input=True)),
                    skip=False, gru=True, fixed=True, variable=True, device="cpu")
    assert sum(p.numel() for p in gnn.parameters()) == 8350
    init_sd = {k: v.clone() for k, v in gnn.state_dict().items()}
    opt = torch.optim.Adam(gnn.parameters(), lr=0.003, weight_decay=1e-6)
    losses = []
    for epoch in range(2):
        gnn.train()                                               # util/train.py:142-149
        for b in range(len(adj_b)):
            opt.zero_grad()
            out = gnn(x_b[b], adj_b[b])
            loss = total_loss(out, (nl_b[b], gl_b[b]), loss="mse", only_nodes=False, only_graph=False)
            loss.backward()
            opt.step()
            losses.append(float(loss.item()))
    print("losses", losses)
    meta = dict(kind="c1_train_trace", seed=42, dataset_seed=1234, B=16, batches=4, epochs=2, N=15, hidden=16, towers=4, scalers=scalers,
                aggregators=C1.AGG, lr=0.003, weight_decay=1e-6, dropout=0.0, n_parameters=8350)
    arrays = dict(adj=adj, x=x, node_labels=nl, graph_labels=gl, avg_lin=avg_d["lin"], avg_log=avg_d["log"], avg_exp=avg_d["exp"],
                  out=np.asarray(losses, dtype=np.float64))       # `out` = the per-step training loss

    class _SD:                                                    # save() stores module.state_dict(): hand it the INITIAL one
        def state_dict(self):
            return init_sd
    save("c1_train_trace_n15", meta, arrays, _SD())


if __name__ == "__main__":
    main()
