"""Operator registry of the dense variant:
AGGREGATORS[name](X:(B,N,N,F), adj:(B,N,N), self_loop=False, device='cpu') -> (B,N,F).

Interface of models/pytorch/pna/aggregators.py:149-152; each call is one launch of the HIP
segment-reduce kernel over the (B*N*N, F) message tensor with the adjacency as edge weight
(mean/sum/std/var: node i reduces over j) or as mask (max/min: node j reduces over i, :37-38,:49-50).
GPU tensors only.  All 13 names of the reference dict are provided: identity, normalised_mean, softmax, softmin and
moment3-5 prepare messages / weights elementwise and reduce through the same kernel.
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


def _eye_like(adj):
    return torch.eye(adj.shape[-1], device=adj.device, dtype=adj.dtype).unsqueeze(0)


class _DenseIndex:
    """The (B*N rows x N edges) CSR of a dense batch, in the shape pna_amd.autograd.AggregateFn expects of a graph."""

    def __init__(self, B, N, device, transposed):
        from ...graph import CSR, HeavySchedule
        rowptr, col_t = _dense_index(B, N, device)
        row = torch.arange(B * N, device=device, dtype=torch.int32).repeat_interleave(N)
        self.csr = CSR(rowptr, col_t if transposed else None, None, row, N)
        self._hs = HeavySchedule(0, 0, 0, 0, None, None, None)
        self.num_nodes = B * N

    def heavy_schedule(self):
        return self._hs

    def workspace(self, nbytes):
        return None

    def work_items(self):
        return None

    def finish_exchange(self):
        pass


def _reduce_w(name, X, w):
    """One kernel launch: aggregator `name` over the (B,N,N,F) message tensor with (B,N,N) edge weights `w`
    (mean/sum/std/var: node i reduces over j with weight w[b,i,j]; max/min: node j over i where w[b,i,j] > 0).
    Differentiable in X (training through the registry functions, like the reference's torch ops): under autograd the call
    goes through AggregateFn, whose weighted backward is pna_amd.autograd._backward_edges_torch."""
    B, N, N2, F = X.shape
    rowptr, col_t = _dense_index(B, N, X.device)
    x = X.contiguous().view(B * N * N, F)
    tr = name in ("max", "min")
    ew = (w.transpose(1, 2) if tr else w).contiguous().view(-1)
    if torch.is_grad_enabled() and X.requires_grad:
        if ew.requires_grad:
            raise NotImplementedError("pna_amd: the gradient w.r.t. the adjacency WEIGHTS of a dense aggregator is not implemented "
                                      "(the reference's benchmark never differentiates the adjacency)")
        from ...autograd import AggregateFn
        g = _DenseIndex(B, N, X.device, tr)
        out = AggregateFn.apply(g, x, None, None, F, (name,), 1, (None,), not tr, ew, None)
        return out[:, :F].reshape(B, N, F)
    if tr:
        out = ops.segreduce(rowptr, col_t, x, F, [name], edge_weight=ew)
    else:
        out = ops.segreduce(rowptr, None, x, F, [name], edge_weight=ew)
    return out.view(B, N, F)


def _reduce(name, X, adj, self_loop):
    return _reduce_w(name, X, adj + _eye_like(adj) if self_loop else adj)


def _aggregator(name):
    def aggregate(X, adj, self_loop=False, device="cpu"):
        return _reduce(name, X, adj, self_loop)
    aggregate.__name__ = "aggregate_" + name
    return aggregate


AGGREGATORS = {n: _aggregator(n) for n in ("mean", "sum", "max", "min", "std", "var")}


# ---- the remaining registry entries of the reference (aggregators.py:11-14, :85-146): elementwise preparation of
# ---- the messages / weights in torch, every reduction over the neighbourhood through the HIP kernel ------------
def aggregate_identity(X, adj, self_loop=False, device="cpu"):
    """Y[b,i] = X[b,i,i] (aggregators.py:11-14): a 1-edge-per-row gather through the kernel."""
    B, N, _, F = X.shape
    i = torch.arange(N, device=X.device)
    col = (torch.arange(B, device=X.device).view(B, 1) * N * N + (i * N + i).view(1, N)).reshape(-1).to(torch.int32)
    if torch.is_grad_enabled() and X.requires_grad:           # the diagonal: a differentiable view is all it takes
        return torch.diagonal(X, dim1=1, dim2=2).permute(0, 2, 1).contiguous()
    rowptr = torch.arange(B * N + 1, dtype=torch.int32, device=X.device)
    return ops.segreduce(rowptr, col, X.contiguous().view(B * N * N, F), F, ["sum"]).view(B, N, F)


def aggregate_normalised_mean(X, adj, self_loop=False, device="cpu"):
    """D^-1/2 A D^-1/2 X (aggregators.py:85-97): a weighted sum with w_ij = a_ij / sqrt(D_i D_j)."""
    a = adj + _eye_like(adj) if self_loop else adj
    r = torch.pow(torch.sum(a, -1), -0.5)
    return _reduce_w("sum", X, r.unsqueeze(-1) * a * r.unsqueeze(-2))


def aggregate_softmax(X, adj, self_loop=False, device="cpu"):
    """sum_j x_j e^{x_j} a_ij / sum_j e^{x_j} a_ij (aggregators.py:100-112): both sums in ONE launch (width 2F)."""
    a = adj + _eye_like(adj) if self_loop else adj
    e = torch.exp(X)
    F = X.shape[-1]
    both = _reduce_w("sum", torch.cat([e * X, e], dim=-1), a)
    return both[..., :F] / both[..., F:]


def aggregate_softmin(X, adj, self_loop=False, device="cpu"):
    return -aggregate_softmax(-X, adj, self_loop=self_loop, device=device)      # aggregators.py:115-117


def _moment(n):
    def aggregate(X, adj, self_loop=False, device="cpu"):
        """sign(m) (|m| + eps)^(1/n), m = E[(x - E x)^n] (aggregators.py:120-132).  As in the reference, with
        self_loop=True the inner mean sees the self loop twice (aggregate_mean adds it again, :129)."""
        a = adj + _eye_like(adj) if self_loop else adj
        mean = _reduce_w("mean", X, a + _eye_like(adj) if self_loop else a)
        m = _reduce_w("sum", torch.pow(X - mean.unsqueeze(2), n), a) / torch.sum(a, -1, keepdim=True)
        return torch.sign(m) * torch.pow(torch.abs(m) + 1e-5, 1.0 / n)
    aggregate.__name__ = f"aggregate_moment_{n}"
    return aggregate


AGGREGATORS.update({"identity": aggregate_identity, "normalised_mean": aggregate_normalised_mean,
                    "softmax": aggregate_softmax, "softmin": aggregate_softmin,
                    "moment3": _moment(3), "moment4": _moment(4), "moment5": _moment(5)})
