"""AGGREGATORS of the PyG formulation -- models/pytorch_geometric/aggregators.py: `fn(src, index, dim_size)` reduces
the rows of `src` (E, ...) into `dim_size` segments given by `index` (E,).  Same names, same semantics: segments
without entries give 0 (torch_scatter's fill, also for min / max), `var` = E[x^2] - E[x]^2 is NOT clamped (:25-28),
`std` = sqrt(relu(var) + 1e-5) (:31-32) -- hence sqrt(1e-5) for an empty segment.

Each call runs the HIP segment-reduce kernel over a CSR built from `index` (the layers in pna.py do that once per
`edge_index` and ask for all aggregators in ONE launch; these entries are for code that uses the registry directly).
"""
import torch

from .. import functional as PF
from ..graph import Graph

_KERNEL_NAME = {"sum": "sum", "mean": "mean", "min": "min", "max": "max", "var": "var_raw", "std": "std"}
EMPTY_STD = 1e-5 ** 0.5


def fix_empty_std(out2d, names, n_scaler, F, deg, factors=None):
    """The kernel writes 0 for every aggregator of a row without in-edges (the DGL variants leave such rows
    undefined); the PyG `std` of an empty segment is sqrt(0 + 1e-5).  Patches those blocks in place:
    out2d (V, n_tower * n_scaler * A * F) with `factors[s]` the per-row scaler (None = identity)."""
    if "std" not in names:
        return out2d
    empty = (deg == 0).nonzero().flatten()
    if empty.numel() == 0:
        return out2d
    A = len(names)
    if out2d.requires_grad or out2d.grad_fn is not None:
        # under autograd the tensor may be one the aggregation saved for its backward (AggregateFn returns the very tensor it
        # saves when there is one identity scaler): patch a copy, out of place
        out2d = out2d.clone()
    view = out2d.view(out2d.shape[0], -1, n_scaler, A, F)
    for a, n in enumerate(names):
        if n != "std":
            continue
        for s in range(n_scaler):
            f = 1.0 if factors is None or factors[s] is None else factors[s][empty].view(-1, 1, 1)
            view[empty, :, s, a, :] = EMPTY_STD * f
    return out2d


def _make(name):
    def aggregate(src, index, dim_size=None):
        if not src.is_cuda:
            raise RuntimeError("pna_amd: aggregators run on the GPU only; there is no CPU path")
        n = int(index.max()) + 1 if dim_size is None else int(dim_size)
        E = src.shape[0]
        flat = src.reshape(E, -1).float()
        g = Graph(torch.arange(E, device=src.device), index, n)       # message k is "edge k"
        g.num_src = E
        msgs = flat[g.csr.eid]                                         # CSR (destination-sorted) order
        out = PF.aggregate(g, msgs, flat.shape[1], [_KERNEL_NAME[name]], edge_resident=True)
        if name == "std":
            out = fix_empty_std(out, ["std"], 1, flat.shape[1], g.in_degrees())
        return out.reshape((n,) + tuple(src.shape[1:]))
    aggregate.__name__ = "aggregate_" + name
    return aggregate


AGGREGATORS = {name: _make(name) for name in ("sum", "mean", "min", "max", "var", "std")}
