"""Dense (B, N, N) adjacency -> the two CSR groupings the dense PNA variant needs.

The reference's dense aggregators reduce mean/sum/std/var over the SECOND node index (node i
aggregates m[b,i,j] over j, weighted by adj[b,i,j]) but max/min over the FIRST (node j takes the
max of m[b,i,j] over i with adj[b,i,j] > 0) -- models/pytorch/pna/aggregators.py:24-26 vs :37-38,
SURVEY.md A.1.  So one edge set is needed in two destination orders:
  `by_row`: destination (b,i), sources j in ascending order, weight adj[b,i,j]
  `by_col`: destination (b,j), sources i in ascending order, weight adj[b,i,j]
Zero entries carry no weight and are masked out of max/min, so they are dropped.
"""
from typing import NamedTuple

import torch

from ...graph import Graph


class DenseGraphs(NamedTuple):
    by_row: Graph
    by_col: Graph
    w_row: torch.Tensor        # fp32 [E] weights in by_row CSR order
    w_col: torch.Tensor        # fp32 [E] weights in by_col CSR order
    row_to_col: torch.Tensor   # int32 [E]: by_col CSR edge k is by_row CSR edge row_to_col[k]
    binary: bool               # every kept weight is exactly 1 (then the weights need not be passed)


def sparsify(adj: torch.Tensor, self_loop: bool) -> DenseGraphs:
    # cached on the adjacency tensor object (the N/2 layer calls of one GNN.forward pass the same object,
    # gnn_framework.py:93-95) per in-place version; never keyed by address
    cache = getattr(adj, "_pna_amd_sparse", None)
    if cache is None or cache[0] != adj._version:
        cache = (adj._version, {})
        try:
            adj._pna_amd_sparse = cache
        except AttributeError:
            pass
    if self_loop in cache[1]:
        return cache[1][self_loop]
    B, N, _ = adj.shape
    a = adj + torch.eye(N, device=adj.device, dtype=adj.dtype).unsqueeze(0) if self_loop else adj
    b, i, j = torch.nonzero(a, as_tuple=True)                 # lexicographic in (b, i, j)
    w_row = a[b, i, j].contiguous()
    V = B * N
    # nonzero() output is already grouped by (b,i) with j ascending: a stable sort by dst keeps it
    g_row = Graph(b * N + j, b * N + i, V, [N] * B)
    # by_col: same edges, destination (b,j); inside a destination order by i
    key_col = (b * N + j) * N + i
    perm = torch.sort(key_col, stable=True).indices           # by_col position -> by_row position
    g_col = Graph((b * N + i)[perm], (b * N + j)[perm], V, [N] * B)
    out = DenseGraphs(g_row, g_col, w_row, w_row[perm].contiguous(), perm.to(torch.int32),
                      bool((w_row == 1).all().item()) if w_row.numel() else True)
    cache[1][self_loop] = out
    return out
