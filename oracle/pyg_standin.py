"""Test-only stand-ins for `torch_geometric` (1.6-era API) and `torch_scatter`, so that the reference's
`models/pytorch_geometric/{aggregators,scalers,pna}.py` run UNMODIFIED in the build container (neither
package is installed).  TEST INFRASTRUCTURE ONLY.

Exactly what the reference touches, restated from the packages' published behaviour (parity for these
third-party pieces is "unpinned": SURVEY.md 8c (iii)):
  * `torch_scatter.scatter(src, index, dim, out, dim_size, reduce)`  (aggregators.py:9-22), reduce in
    {sum, mean, min, max}; segments without entries give 0 (also for min / max), mean divides by
    max(count, 1)
  * `torch_geometric.utils.degree(index, num_nodes, dtype)`           (pna.py:157,:247)
  * `torch_geometric.nn.conv.MessagePassing` with `aggr=None, node_dim=0`, flow source_to_target:
    `propagate(edge_index, size=None, **kw)` -> `message(...)` (arguments `<name>_i` = target rows
    `edge_index[1]`, `<name>_j` = source rows `edge_index[0]`) -> `aggregate(inputs, index=edge_index[1],
    dim_size=N)` -> `update` (identity)                                (pna.py:129,:135-151,:238-251)
  * `torch_geometric.nn.inits.reset`, `torch_geometric.typing.{Adj, OptTensor}`

Nothing here is used on the GPU box: /root/reference does not exist there.
"""
import inspect
import sys
import types
from typing import Optional

import torch

REFERENCE_ROOT = "/root/reference"


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    assert dim == 0 and out is None
    n = int(index.max()) + 1 if dim_size is None else int(dim_size)
    shape = (n,) + tuple(src.shape[1:])
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    if reduce in ("sum", "add"):
        return torch.zeros(shape, dtype=src.dtype).scatter_add_(0, idx, src)
    if reduce == "mean":
        s = torch.zeros(shape, dtype=src.dtype).scatter_add_(0, idx, src)
        cnt = torch.zeros(n, dtype=src.dtype).scatter_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        return s / cnt.clamp(min=1).view((-1,) + (1,) * (src.dim() - 1))
    if reduce in ("min", "max"):
        r = torch.zeros(shape, dtype=src.dtype).scatter_reduce_(0, idx, src, "amin" if reduce == "min" else "amax",
                                                                include_self=False)
        return r                     # untouched (empty) segments keep the zero fill, like torch_scatter
    raise ValueError(reduce)


def degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else int(num_nodes)
    out = torch.zeros(n, dtype=dtype or torch.float32)
    return out.scatter_add_(0, index, torch.ones(index.numel(), dtype=out.dtype))


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=0, **kwargs):
        super().__init__()
        assert flow == "source_to_target" and node_dim == 0
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        j, i = edge_index[0], edge_index[1]
        n = None
        for v in kwargs.values():
            if torch.is_tensor(v):
                n = v.shape[0]
                break
        args = {}
        for name in inspect.signature(self.message).parameters:
            if name.endswith("_i") or name.endswith("_j"):
                base = kwargs[name[:-2]]
                args[name] = None if base is None else base[i if name.endswith("_i") else j]
            else:
                args[name] = kwargs.get(name)
        out = self.message(**args)
        out = self.aggregate(out, index=i, dim_size=n)
        return self.update(out)

    def message(self, x_j):
        return x_j

    def update(self, inputs):
        return inputs


def reset(nn):
    def _reset(item):
        if hasattr(item, "reset_parameters"):
            item.reset_parameters()

    if nn is not None:
        if hasattr(nn, "children") and len(list(nn.children())) > 0:
            for item in nn.children():
                _reset(item)
        else:
            _reset(nn)


def install():
    """Register the stub modules and put the reference on sys.path."""
    if "torch_scatter" not in sys.modules:
        ts = types.ModuleType("torch_scatter")
        ts.scatter = scatter
        sys.modules["torch_scatter"] = ts
    if "torch_geometric" not in sys.modules:
        tg = types.ModuleType("torch_geometric")
        typing_m = types.ModuleType("torch_geometric.typing")
        typing_m.Adj = torch.Tensor
        typing_m.OptTensor = Optional[torch.Tensor]
        nn_m = types.ModuleType("torch_geometric.nn")
        conv_m = types.ModuleType("torch_geometric.nn.conv")
        conv_m.MessagePassing = MessagePassing
        inits_m = types.ModuleType("torch_geometric.nn.inits")
        inits_m.reset = reset
        utils_m = types.ModuleType("torch_geometric.utils")
        utils_m.degree = degree
        nn_m.conv, nn_m.inits = conv_m, inits_m
        tg.typing, tg.nn, tg.utils = typing_m, nn_m, utils_m
        for name, mod in (("torch_geometric", tg), ("torch_geometric.typing", typing_m), ("torch_geometric.nn", nn_m),
                          ("torch_geometric.nn.conv", conv_m), ("torch_geometric.nn.inits", inits_m),
                          ("torch_geometric.utils", utils_m)):
            sys.modules[name] = mod
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
