"""Operator registry of the dense variant:
AGGREGATORS[name](X:(B,N,N,F), adj:(B,N,N), self_loop=False, device='cpu') -> (B,N,F).

Interface of models/pytorch/pna/aggregators.py:149-152; each call is one launch of the HIP
segment-reduce kernel over the (B*N*N, F) message tensor with the adjacency as edge weight
(mean/sum/std/var: node i reduces over j) or as mask (max/min: node j reduces over i, :37-38,:49-50).
GPU tensors only.  The remaining names of the reference dict (identity, normalised_mean, softmax,
softmin, moment3-5) are SURVEY.md 8f row N4 ("next") and raise KeyError here.
"""
import torch

from ... import ops

_IDX = {}


def _dense_index(B, N, device):
    key = (B, N, str(device))
    if key not in _IDX:
        rowptr = torch.arange(0, (B * N + 1) * N, N, dtype=torch.int32, device=device)
        b = torch.arange(B, device=device).view(B, 1, 1)
        jj = torch.arange(N, device=device).view(1, N, 1)
        ii = torch.arange(N, device=device).view(1, 1, N)
        col_t = (b * N * N + ii * N + jj).reshape(-1).to(torch.int32)      # row (b,j): messages (b,i,j), i ascending
        if len(_IDX) > 16:
            _IDX.clear()
        _IDX[key] = (rowptr, col_t)
    return _IDX[key]


def _reduce(name, X, adj, self_loop):
    B, N, N2, F = X.shape
    if self_loop:
        adj = adj + torch.eye(N, device=adj.device, dtype=adj.dtype).unsqueeze(0)
    rowptr, col_t = _dense_index(B, N, X.device)
    x = X.contiguous().view(B * N * N, F)
    if name in ("max", "min"):
        w = adj.transpose(1, 2).contiguous().view(-1)
        out = ops.segreduce(rowptr, col_t, x, F, [name], edge_weight=w)
    else:
        out = ops.segreduce(rowptr, None, x, F, [name], edge_weight=adj.contiguous().view(-1))
    return out.view(B, N, F)


def _aggregator(name):
    def aggregate(X, adj, self_loop=False, device="cpu"):
        return _reduce(name, X, adj, self_loop)
    aggregate.__name__ = "aggregate_" + name
    return aggregate


AGGREGATORS = {n: _aggregator(n) for n in ("mean", "sum", "max", "min", "std", "var")}
