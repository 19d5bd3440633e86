#!/usr/bin/env python
"""Generate golden input/output vectors by running the REFERENCE's own source
(/root/reference, read-only) on CPU.  TEST INFRASTRUCTURE ONLY.

The reference ships no tests or known-answer vectors, so these
files are the pins: they are produced by `models/dgl/pna_layer.py` (executed unmodified
over oracle/dgl_standin.py) and by `models/pytorch/pna/layer.py` (imported unmodified).

Run in the build container only (the GPU box has no /root/reference):
    python oracle/make_golden.py            # writes tests/golden/*.npz
The committed fixtures were produced with torch 2.10.0 (CPU path), seeds below.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dgl_standin  # noqa: E402

dgl_standin.install()
from models.dgl.pna_layer import PNALayer as RefDGLLayer, PNASimpleLayer as RefSimpleLayer  # noqa: E402
from models.pytorch.pna.layer import PNALayer as RefDenseLayer  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
AGG4 = "mean max min std"
SCA3 = "identity amplification attenuation"


def powerlaw_graph(rng, N, E, min_in_degree=1):
    """Small multigraph with a skewed in/out-degree profile; every node gets >= min_in_degree
    in-edges (the reference datasets have no isolated nodes)."""
    w = (np.arange(N) + 1.0) ** -0.6
    w /= w.sum()
    perm = rng.permutation(N)
    src = perm[rng.choice(N, size=E, p=w)]
    dst = perm[rng.choice(N, size=E, p=w)]
    extra_dst = np.repeat(np.arange(N), min_in_degree)
    extra_src = rng.integers(0, N, size=extra_dst.size)
    src = np.concatenate([src, extra_src])
    dst = np.concatenate([dst, extra_dst])
    p = rng.permutation(src.size)
    return src[p].astype(np.int64), dst[p].astype(np.int64)


def molecule_batch(rng, n_graphs, mean_nodes=23.2):
    """ZINC-like batch: random spanning tree + ~10% ring closures, symmetrised."""
    srcs, dsts, sizes, off = [], [], [], 0
    for _ in range(n_graphs):
        n = int(np.clip(round(rng.normal(mean_nodes, 4.3)), 9, 38))
        u = np.arange(1, n)
        v = np.array([rng.integers(0, i) for i in u])
        k = max(1, n // 10)
        ru, rv = rng.integers(0, n, k), rng.integers(0, n, k)
        keep = ru != rv
        a = np.concatenate([u, ru[keep]]) + off
        b = np.concatenate([v, rv[keep]]) + off
        srcs += [a, b]
        dsts += [b, a]
        sizes.append(n)
        off += n
    return np.concatenate(srcs).astype(np.int64), np.concatenate(dsts).astype(np.int64), sizes


def randomise(module, gen):
    """Replace the reference's tiny xavier(gain=1/in) init with weights ~ N(0, 1/fan_in) (activations
    stay O(1)) and O(0.3) biases so that parity checks have teeth; BatchNorm affine parameters and
    running stats are made non-trivial too."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if "batchnorm" in name:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.3 + (1.0 if name.endswith("weight") else 0.0))
            elif p.dim() == 2:
                p.copy_(torch.randn(p.shape, generator=gen) / p.shape[1] ** 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.3)
        for name, b in module.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=gen) * 0.2)
            if name.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=gen) + 0.5)


def save(name, meta, arrays, module):
    sd = {"sd/" + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}
    arrays = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, meta=json.dumps(meta), **arrays, **sd)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  out={arrays['out'].shape}")


def golden_simple(name, seed, N, E, F, out_dim, aggregators=AGG4, scalers=SCA3, residual=True,
                  default_init=False, posttrans_layers=1):
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    src, dst = powerlaw_graph(rng, N, E)
    deg = np.bincount(dst, minlength=N)
    avg_log = torch.tensor(float(np.mean(np.log(deg + 1))), dtype=torch.float32)
    torch.manual_seed(seed)                      # the reference's default init draws from the GLOBAL generator: pin it
    layer = RefSimpleLayer(F, out_dim, aggregators, scalers, {"log": avg_log}, 0.0, True, residual,
                           posttrans_layers=posttrans_layers).eval()
    if not default_init:
        randomise(layer, gen)
    h = torch.randn(N, F, generator=gen)
    g = dgl_standin.StandinGraph(src, dst, N)
    with torch.no_grad():
        out = layer(g, h)
        # the (V, A*S*F) tensor the reduce step leaves in ndata['h'] (pna_layer.py:194,:203)
        g2 = dgl_standin.StandinGraph(src, dst, N)
        g2.ndata["h"] = h
        import dgl.function as fn
        g2.update_all(fn.copy_u("h", "m"), layer.reduce_func)
        agg = g2.ndata["h"]
    meta = dict(kind="dgl_simple", seed=seed, N=N, F=F, out_dim=out_dim, aggregators=aggregators,
                scalers=scalers, residual=residual, batch_norm=True, posttrans_layers=posttrans_layers)
    save(name, meta, dict(src=src, dst=dst, h=h, avg_log=avg_log, agg=agg, out=out), layer)


def golden_tower(name, seed, in_dim, out_dim, towers, divide_input, edge_dim=0, pretrans_layers=1,
                 posttrans_layers=1, n_graphs=6, graph_norm=True, batch_norm=True, residual=True,
                 aggregators=AGG4, scalers=SCA3, edge_types=0):
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    src, dst, sizes = molecule_batch(rng, n_graphs)
    N = int(sum(sizes))
    deg = np.bincount(dst, minlength=N)
    avg_log = torch.tensor(float(np.mean(np.log(deg + 1))), dtype=torch.float32)
    layer = RefDGLLayer(in_dim, out_dim, aggregators, scalers, {"log": avg_log}, 0.0, graph_norm, batch_norm,
                        towers=towers, pretrans_layers=pretrans_layers, posttrans_layers=posttrans_layers,
                        divide_input=divide_input, residual=residual, edge_features=edge_dim > 0,
                        edge_dim=edge_dim).eval()
    randomise(layer, gen)
    h = torch.randn(N, in_dim, generator=gen)
    e = torch.randn(src.size, edge_dim, generator=gen) if edge_dim > 0 else torch.zeros(src.size, 0)
    if edge_types > 0:        # the molecule nets' edge features: an embedding of the bond type (nets/molecules_graph_regression/pna_net.py)
        emb = torch.nn.Embedding(edge_types, edge_dim)
        with torch.no_grad():
            emb.weight.copy_(torch.randn(edge_types, edge_dim, generator=gen))
            e = emb(torch.randint(0, edge_types, (src.size,), generator=gen))
    snorm_n = torch.cat([torch.full((s, 1), 1.0 / s) for s in sizes]).sqrt()   # data/molecules.py:157-159
    g = dgl_standin.StandinGraph(src, dst, N)
    with torch.no_grad():
        out = layer(g, h, e if edge_dim > 0 else None, snorm_n)
    meta = dict(kind="dgl_tower", seed=seed, N=N, in_dim=in_dim, out_dim=out_dim, towers=towers,
                divide_input=divide_input, edge_dim=edge_dim, pretrans_layers=pretrans_layers,
                posttrans_layers=posttrans_layers, graph_norm=graph_norm, batch_norm=batch_norm,
                residual=residual, aggregators=aggregators, scalers=scalers, sizes=sizes)
    save(name, meta, dict(src=src, dst=dst, h=h, e=e, snorm_n=snorm_n, avg_log=avg_log, out=out), layer)


def golden_dense(name, seed, B, N, in_f, out_f, towers, divide_input, scalers, aggregators=("mean", "max", "min", "std"),
                 self_loop=False, symmetric=True, p=0.3):
    gen = torch.Generator().manual_seed(seed)
    adj = (torch.rand(B, N, N, generator=gen) < p).float()
    adj = adj * (1 - torch.eye(N))
    if symmetric:                                           # benchmark graphs are symmetric 0/1
        adj = torch.maximum(adj, adj.transpose(1, 2))
    ring = torch.zeros(N, N)
    idx = torch.arange(N)
    ring[idx, (idx + 1) % N] = 1
    ring[(idx + 1) % N, idx] = 1                            # no isolated nodes, rows AND columns non-empty
    adj = torch.maximum(adj, ring.unsqueeze(0))
    D = adj.sum(-1)
    avg_d = dict(lin=torch.mean(D), exp=torch.mean(torch.exp(torch.div(1, D)) - 1), log=torch.mean(torch.log(D + 1)))
    layer = RefDenseLayer(in_f, out_f, list(aggregators), list(scalers), avg_d, towers=towers, self_loop=self_loop,
                          divide_input=divide_input).eval()
    randomise(layer, gen)
    x = torch.randn(B, N, in_f, generator=gen)
    with torch.no_grad():
        out = layer(x, adj)
    meta = dict(kind="dense", seed=seed, B=B, N=N, in_features=in_f, out_features=out_f, towers=towers,
                divide_input=divide_input, scalers=list(scalers), aggregators=list(aggregators),
                self_loop=self_loop, symmetric=symmetric)
    save(name, meta, dict(x=x, adj=adj, avg_lin=avg_d["lin"], avg_log=avg_d["log"], out=out), layer)


def golden_net(name, seed, hidden, out_dim, L, towers, edge_dim, readout, gru=False, n_graphs=7):
    """The whole molecules PNANet (realworld_benchmark/nets/molecules_graph_regression/pna_net.py) -- SURVEY 8f N2."""
    sys.path.insert(0, os.path.join(dgl_standin.REFERENCE_ROOT, "realworld_benchmark"))
    from nets.molecules_graph_regression.pna_net import PNANet as RefNet
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    src, dst, sizes = molecule_batch(rng, n_graphs)
    N = int(sum(sizes))
    deg = np.bincount(dst, minlength=N)
    avg_log = torch.tensor(float(np.mean(np.log(deg + 1))), dtype=torch.float32)
    params = dict(num_atom_type=28, num_bond_type=4, hidden_dim=hidden, out_dim=out_dim, in_feat_dropout=0.0, dropout=0.0,
                  L=L, readout=readout, graph_norm=True, batch_norm=True, residual=True, aggregators=AGG4, scalers=SCA3,
                  avg_d={"log": avg_log}, towers=towers, divide_input_first=False, divide_input_last=True,
                  edge_feat=edge_dim > 0, edge_dim=edge_dim, pretrans_layers=1, posttrans_layers=1, gru=gru, device="cpu")
    net = RefNet(params).eval()
    randomise(net, gen)
    with torch.no_grad():
        net.embedding_h.weight.copy_(torch.randn(net.embedding_h.weight.shape, generator=gen))
        if edge_dim > 0:
            net.embedding_e.weight.copy_(torch.randn(net.embedding_e.weight.shape, generator=gen))
    atoms = torch.randint(0, 28, (N,), generator=gen)
    bonds = torch.randint(0, 4, (src.size,), generator=gen)
    snorm_n = torch.cat([torch.full((s, 1), 1.0 / s) for s in sizes]).sqrt()
    g = dgl_standin.StandinGraph(src, dst, N, sizes)
    with torch.no_grad():
        out = net(g, atoms, bonds, snorm_n, None)
    meta = dict(kind="net_molecules", seed=seed, N=N, sizes=sizes, hidden_dim=hidden, out_dim=out_dim, L=L, towers=towers,
                edge_dim=edge_dim, readout=readout, gru=gru, aggregators=AGG4, scalers=SCA3)
    save(name, meta, dict(src=src, dst=dst, atoms=atoms, bonds=bonds, snorm_n=snorm_n, avg_log=avg_log, out=out), net)


def knn_batch(rng, n_graphs, mean_nodes, k):
    """Superpixel-like batch: per graph n random 2-D positions, every node receives an edge from each of its k nearest neighbours
    (data/superpixels.py builds its graphs the same way: k = 8 nearest superpixel centres).  -> src, dst, sizes, pos."""
    srcs, dsts, sizes, poss, off = [], [], [], [], 0
    for _ in range(n_graphs):
        n = int(np.clip(round(rng.normal(mean_nodes, mean_nodes * 0.08)), k + 2, None))
        pos = rng.random((n, 2))
        d2 = ((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1)
        np.fill_diagonal(d2, np.inf)
        nb = np.argsort(d2, axis=1)[:, :k]
        dsts.append(np.repeat(np.arange(n), k) + off)
        srcs.append(nb.reshape(-1) + off)
        sizes.append(n)
        poss.append(pos)
        off += n
    return np.concatenate(srcs).astype(np.int64), np.concatenate(dsts).astype(np.int64), sizes, np.concatenate(poss)


def golden_net_superpixels(name, seed, hidden, out_dim, L, towers, edge_feat, readout, n_graphs=6, mean_nodes=24, k=8, in_dim=5, n_classes=10,
                           divide_first=True, divide_last=False, gru=False):
    """The whole superpixels PNANet (realworld_benchmark/nets/superpixels_graph_classification/pna_net.py:17-104; CIFAR10 json:
    5 towers, divide_input_first=True, divide_input_last=False, readout sum) -- VERDICT r4 item 9."""
    sys.path.insert(0, os.path.join(dgl_standin.REFERENCE_ROOT, "realworld_benchmark"))
    from nets.superpixels_graph_classification.pna_net import PNANet as RefNet
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    src, dst, sizes, pos = knn_batch(rng, n_graphs, mean_nodes, k)
    N = int(sum(sizes))
    deg = np.bincount(dst, minlength=N)
    avg_log = torch.tensor(float(np.mean(np.log(deg + 1))), dtype=torch.float32)
    edge_dim = 6 if edge_feat else 0
    params = dict(in_dim=in_dim, in_dim_edge=1, hidden_dim=hidden, out_dim=out_dim, n_classes=n_classes, in_feat_dropout=0.0, dropout=0.0,
                  L=L, readout=readout, graph_norm=True, batch_norm=True, residual=True, aggregators=AGG4, scalers=SCA3,
                  avg_d={"log": avg_log}, towers=towers, divide_input_first=divide_first, divide_input_last=divide_last,
                  edge_feat=edge_feat, edge_dim=edge_dim, pretrans_layers=1, posttrans_layers=1, gru=gru, device="cpu")
    net = RefNet(params).eval()
    randomise(net, gen)
    x = torch.cat([torch.rand(N, in_dim - 2, generator=gen), torch.from_numpy(pos).float()], dim=1)       # (mean colour | x, y)
    e = torch.from_numpy(np.sqrt(((pos[src] - pos[dst]) ** 2).sum(-1, keepdims=True))).float()         # edge feature: the distance
    snorm_n = torch.cat([torch.full((s, 1), 1.0 / s) for s in sizes]).sqrt()
    g = dgl_standin.StandinGraph(src, dst, N, sizes)
    with torch.no_grad():
        out = net(g, x, e, snorm_n, None)
    meta = dict(kind="net_superpixels", seed=seed, N=N, sizes=sizes, in_dim=in_dim, n_classes=n_classes, hidden_dim=hidden, out_dim=out_dim, L=L,
                towers=towers, edge_feat=edge_feat, edge_dim=edge_dim, readout=readout, gru=gru, divide_input_first=divide_first,
                divide_input_last=divide_last, aggregators=AGG4, scalers=SCA3)
    save(name, meta, dict(src=src, dst=dst, x=x, e=e, snorm_n=snorm_n, avg_log=avg_log, out=out), net)


def golden_dense_registry(name, seed, B=3, N=7, F=5):
    """Every entry of the dense operator registries (models/pytorch/pna/aggregators.py:149-152, scalers.py:41-42)
    evaluated by the reference itself on one random message tensor -- SURVEY 8f N4."""
    from models.pytorch.pna.aggregators import AGGREGATORS as REF_AGG
    from models.pytorch.pna.scalers import SCALERS as REF_SCA
    gen = torch.Generator().manual_seed(seed)
    X = torch.randn(B, N, N, F, generator=gen)
    adj = (torch.rand(B, N, N, generator=gen) < 0.4).float() * (1 - torch.eye(N))
    idx = torch.arange(N)
    ring = torch.zeros(N, N)
    ring[idx, (idx + 1) % N] = 1
    ring[(idx + 1) % N, idx] = 1
    adj = torch.maximum(adj, ring.unsqueeze(0))            # directed otherwise: rows and columns non-empty
    D = adj.sum(-1)
    avg_d = dict(lin=torch.mean(D), log=torch.mean(torch.log(D + 1)))
    arrays = dict(X=X, adj=adj, avg_lin=avg_d["lin"], avg_log=avg_d["log"], out=torch.zeros(1))
    for k, f in REF_AGG.items():
        for sl in (False, True):
            arrays[f"agg/{k}/{int(sl)}"] = f(X, adj, self_loop=sl)
    m = torch.randn(B, N, 2 * F, generator=gen)
    arrays["m"] = m
    for k, f in REF_SCA.items():
        arrays[f"sca/{k}"] = f(m, adj, avg_d=avg_d)
    meta = dict(kind="dense_registry", seed=seed, B=B, N=N, F=F, aggregators=list(REF_AGG), scalers=list(REF_SCA))
    save(name, meta, arrays, torch.nn.Module())


def golden_dgl_registry(name, seed, n=6, d=5, F=7):
    """Every entry of the DGL operator registries (models/dgl/aggregators.py:54-56, scalers.py:22) evaluated by the reference
    itself on one random mailbox -- the moment entries included, whole-tensor mean and all."""
    from models.dgl.aggregators import AGGREGATORS as REF_AGG
    from models.dgl.scalers import SCALERS as REF_SCA
    gen = torch.Generator().manual_seed(seed)
    h = torch.randn(n, d, F, generator=gen) * 1.5 + 0.3
    avg_d = dict(log=torch.tensor(1.37))
    arrays = dict(h=h, avg_log=avg_d["log"], out=torch.zeros(1))
    for k, f in REF_AGG.items():
        arrays[f"agg/{k}"] = f(h)
    m = torch.randn(n, 2 * F, generator=gen)
    arrays["m"] = m
    for k, f in REF_SCA.items():
        arrays[f"sca/{k}"] = f(m, d, avg_d)
    meta = dict(kind="dgl_registry", seed=seed, n=n, d=d, F=F, aggregators=list(REF_AGG), scalers=list(REF_SCA))
    save(name, meta, arrays, torch.nn.Module())


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    # --- PNASimpleLayer (MolHIV path, models/dgl/pna_layer.py:151-216) ---
    golden_simple("simple_f75", 1234, N=96, E=900, F=75, out_dim=75)               # the roofline width
    golden_simple("simple_f80_hiv", 42, N=64, E=200, F=80, out_dim=80)               # README.md:45 widths
    golden_simple("simple_f20_order", 7, N=90, E=700, F=20, out_dim=12,
                  aggregators="max std sum mean var min", scalers="attenuation identity", residual=False)
    golden_simple("simple_f16_default_init", 41, N=50, E=300, F=16, out_dim=16, default_init=True)
    golden_simple("simple_f33_post2", 3, N=70, E=400, F=33, out_dim=20, posttrans_layers=2, residual=False)
    # --- PNALayer with towers (ZINC path, models/dgl/pna_layer.py:17-148) ---
    golden_tower("tower_zinc_first", 41, in_dim=30, out_dim=30, towers=5, divide_input=False)
    golden_tower("tower_zinc_last", 43, in_dim=30, out_dim=25, towers=5, divide_input=True)
    golden_tower("tower_edgefeat", 44, in_dim=24, out_dim=24, towers=4, divide_input=True, edge_dim=6)
    golden_tower("tower_edgetype", 47, in_dim=30, out_dim=30, towers=5, divide_input=False, edge_dim=8, edge_types=4, n_graphs=8)
    golden_tower("tower_edgetype_div", 48, in_dim=32, out_dim=32, towers=4, divide_input=True, edge_dim=5, edge_types=3)
    golden_tower("tower_deep_mlps", 45, in_dim=16, out_dim=16, towers=2, divide_input=False, pretrans_layers=2,
                 posttrans_layers=2, graph_norm=False)
    golden_tower("tower_f75", 46, in_dim=75, out_dim=70, towers=5, divide_input=False, n_graphs=3)
    # BASELINE.json configs[3] words the MolHIV net as "8 towers" at hidden 80: the reference's HIV net has no towers
    # (nets/HIV_graph_classification/pna_net.py:30-38), so this is an EXTENSION: DGL PNALayer(towers=8, divide_input=True) at 80 -> 80
    # (10 features per tower), SURVEY 8d C4
    golden_tower("tower_hiv_t8_div", 49, in_dim=80, out_dim=80, towers=8, divide_input=True, n_graphs=12)
    # --- whole molecules net incl. embeddings, graph readout, MLPReadout ---
    golden_net("net_zinc_sum_edgefeat", 41, hidden=20, out_dim=20, L=3, towers=5, edge_dim=6, readout="sum")
    golden_net("net_zinc_mean_gru", 42, hidden=16, out_dim=16, L=2, towers=4, edge_dim=0, readout="mean", gru=True)
    golden_net("net_zinc_max", 43, hidden=12, out_dim=8, L=2, towers=2, edge_dim=0, readout="max")
    # --- whole superpixels net (CIFAR10 / MNIST configs): Linear embeddings of float features, n_classes readout ---
    golden_net_superpixels("net_superpixels_cifar", 41, hidden=30, out_dim=25, L=3, towers=5, edge_feat=False, readout="sum")
    golden_net_superpixels("net_superpixels_edgefeat_gru", 42, hidden=16, out_dim=16, L=2, towers=4, edge_feat=True, readout="mean", in_dim=3,
                           divide_first=False, divide_last=True, gru=True)
    golden_dense_registry("dense_registry_all", 9)
    golden_dgl_registry("dgl_registry_all", 11)
    # --- dense variant (multitask path, models/pytorch/pna/layer.py) ---
    golden_dense("dense_multitask_mid", 1234, B=6, N=14, in_f=16, out_f=16, towers=4, divide_input=True,
                 scalers=("identity",))
    golden_dense("dense_multitask_first", 42, B=5, N=11, in_f=2, out_f=16, towers=4, divide_input=False,
                 scalers=("identity", "amplification", "attenuation"))
    golden_dense("dense_linear_scalers", 5, B=3, N=9, in_f=8, out_f=8, towers=2, divide_input=True,
                 scalers=("identity", "linear", "inverse_linear"), aggregators=("mean", "sum", "max", "min", "std", "var"))
    golden_dense("dense_directed", 6, B=4, N=10, in_f=8, out_f=8, towers=2, divide_input=True,
                 scalers=("identity", "amplification", "attenuation"), symmetric=False)
    golden_dense("dense_self_loop", 8, B=3, N=8, in_f=8, out_f=4, towers=1, divide_input=True,
                 scalers=("identity", "amplification"), self_loop=True)


if __name__ == "__main__":
    main()
