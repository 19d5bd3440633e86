"""Deterministic synthetic workloads of the shapes BASELINE.json / SURVEY.md section 8(d) name.

The real datasets (ZINC, MolHIV) are not available offline; these generators produce graphs with
the same shape statistics.  Used by bench.py and the tests only -- the layers never import this.
"""
import numpy as np
import torch


def powerlaw_graph(V, E, seed=1234, device="cpu"):
    """Chung-Lu power-law multigraph of SURVEY.md 8(d) "C3 -- roofline config".

    weights w_k = (k+1)^-0.5 over a random permutation of node ids; (E - 2V)/2 undirected pairs
    with both endpoints ~ categorical(w), each stored in both directions, plus a ring v <-> v+1
    (2V directed edges) so that every node has in-degree >= 2 (the reference rejects graphs with
    singleton nodes, multitask_benchmark/datasets_generation/multitask_dataset.py:46-49).
    Multi-edges and self pairs are kept (DGL graphs allow them).  Returns int64 (src, dst) of
    exactly E directed edges, on `device`.
    """
    assert E >= 2 * V and (E - 2 * V) % 2 == 0
    g = torch.Generator(device="cpu").manual_seed(seed)
    n_pairs = (E - 2 * V) // 2
    w = (torch.arange(V, dtype=torch.float64) + 1.0) ** -0.5
    cdf = torch.cumsum(w, 0)
    cdf = (cdf / cdf[-1]).to(device)
    perm = torch.randperm(V, generator=g).to(device)
    # inverse-CDF sampling; the uniforms come from the CPU generator so that the graph is identical
    # on every rank and every device type
    u = torch.rand(2, n_pairs, generator=g, dtype=torch.float64).to(device)
    a = perm[torch.searchsorted(cdf, u[0]).clamp_(max=V - 1)]
    b = perm[torch.searchsorted(cdf, u[1]).clamp_(max=V - 1)]
    ring = torch.arange(V, device=device)
    nxt = (ring + 1) % V
    src = torch.cat([a, b, ring, nxt])
    dst = torch.cat([b, a, nxt, ring])
    return src, dst


def molecule_batch(n_graphs, mean_nodes=23.2, sd_nodes=4.3, lo=9, hi=38, seed=41, lognormal=False):
    """Batch of molecule-like graphs (random spanning tree + ~10% ring-closing bonds, symmetrised),
    the ZINC / MolHIV stand-in of SURVEY.md 8(d) C2 / C4.  Returns (src, dst, sizes) with node ids
    offset per graph exactly like dgl.batch (realworld_benchmark/data/molecules.py:163)."""
    rng = np.random.default_rng(seed)
    if lognormal:   # MolHIV: mean 25.5, sd 12, graphs with > 5 nodes (data/HIV.py:17)
        sigma2 = np.log(1 + (sd_nodes / mean_nodes) ** 2)
        n = rng.lognormal(np.log(mean_nodes) - sigma2 / 2, np.sqrt(sigma2), n_graphs)
    else:
        n = rng.normal(mean_nodes, sd_nodes, n_graphs)
    sizes = np.clip(np.rint(n), lo, hi).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    srcs, dsts = [], []
    for n_i, off in zip(sizes, offs):
        u = np.arange(1, n_i)
        v = (rng.random(n_i - 1) * u).astype(np.int64)          # parent in a random recursive tree
        k = max(1, n_i // 10)
        ru, rv = rng.integers(0, n_i, k), rng.integers(0, n_i, k)
        keep = ru != rv
        a = np.concatenate([u, ru[keep]]) + off
        b = np.concatenate([v, rv[keep]]) + off
        srcs += [a, b]
        dsts += [b, a]
    return (torch.from_numpy(np.concatenate(srcs)), torch.from_numpy(np.concatenate(dsts)),
            [int(s) for s in sizes])
