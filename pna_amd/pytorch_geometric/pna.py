"""PNAConv / PNAConvSimple -- drop-in for models/pytorch_geometric/pna.py (the PyG formulation of the layer).

Same constructor and `forward(x, edge_index, edge_attr=None)` signatures, same assertions, same state_dict keys
(`pre_nns.{t}.{0,2,..}`, `post_nns.{t}.{0,2,..}`, `edge_encoder`, `lin`; `post_nn.{0,2,..}` for the simple layer) and
the same `avg_deg` dictionary computed from the degree histogram `deg` (pna.py:84-91,:214-221).  Inside forward:

  reference (MessagePassing.propagate)                       here
  ---------------------------------------------------------  --------------------------------------------------
  message: pre_nn(cat[x_i, x_j, enc(e)]) per edge, per tower  a 1-layer pre_nn is affine: node-level projections
    (pna.py:135-149)                                          P = W_i x (+b), Q = W_j x, R = W_e enc(e); the
                                                              per-edge message P[i] + Q[j] + R[k] is formed inside
                                                              the gather kernel, never in HBM
  aggregate: one torch_scatter call per aggregator, then      ONE pna_segreduce_fwd_f32 launch for all towers and
    the scalers on the concatenation (:151-158, :241-251)     aggregators; scalers enter the contraction per row
  cat[x, out] -> post_nn Linear (:131-133)                    pna_posttrans_*: the 12F tensor is never built

`edge_index[0]` = source j, `edge_index[1]` = target i (flow source_to_target).  Nodes without in-edges follow the
reference's PyG rules (scalers.py:16-19,:26-29; 0 for sum/mean/min/max/var, sqrt(1e-5) for std).  The CSR of an
`edge_index` tensor is cached on the tensor object.  GPU tensors only.
"""
from typing import Dict, List, Optional

import torch
from torch import Tensor
from torch.nn import Linear, ModuleList, ReLU, Sequential

from .. import functional as PF
from ..graph import Graph
from .aggregators import AGGREGATORS, _KERNEL_NAME, fix_empty_std
from .scalers import SCALERS, row_factor


def _avg_deg(deg: Tensor) -> Dict[str, float]:
    deg = deg.to(torch.float)
    total_no_vertices = deg.sum()
    bin_degrees = torch.arange(len(deg), device=deg.device)
    return {
        "lin": ((bin_degrees * deg).sum() / total_no_vertices).item(),
        "log": (((bin_degrees + 1).log() * deg).sum() / total_no_vertices).item(),
        "exp": ((bin_degrees.exp() * deg).sum() / total_no_vertices).item(),
    }


def _mlp(n_in, n_out, layers):
    modules = [Linear(n_in, n_out)]
    for _ in range(layers - 1):
        modules += [ReLU()]
        modules += [Linear(n_out, n_out)]
    return Sequential(*modules)


def _graph_of(edge_index: Tensor, num_nodes: int) -> Graph:
    """CSR of `edge_index`, cached on the tensor object per (version, num_nodes) -- never keyed by address."""
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError("edge_index must have shape (2, E)")
    key = (edge_index._version, num_nodes)
    hit = getattr(edge_index, "_pna_amd_graph", None)
    if hit is None or hit[0] != key:
        hit = (key, Graph(edge_index[0], edge_index[1], num_nodes))
        try:
            edge_index._pna_amd_graph = hit
        except AttributeError:
            pass
    return hit[1]


def _row_factors(graph: Graph, scalers: List[str], avg_deg):
    """Per-row multipliers of the scalers (None = identity) and the in-degree vector."""
    deg = graph.in_degrees().to(torch.float32)
    return [None if s == "identity" else row_factor(s, deg, avg_deg).contiguous() for s in scalers], deg


class PNAConv(torch.nn.Module):
    """The full layer (pre_nns on [x_i, x_j, edge], towers, post_nns on [x, aggregate], final mixing Linear)."""

    def __init__(self, in_channels: int, out_channels: int, aggregators: List[str], scalers: List[str], deg: Tensor,
                 edge_dim: Optional[int] = None, towers: int = 1, pre_layers: int = 1, post_layers: int = 1,
                 divide_input: bool = False, **kwargs):
        super().__init__()
        if divide_input:
            assert in_channels % towers == 0
        assert out_channels % towers == 0
        self.in_channels, self.out_channels = in_channels, out_channels
        self.aggregator_names = list(aggregators)
        self.scaler_names = list(scalers)
        self.aggregators = [AGGREGATORS[a] for a in aggregators]       # KeyError on unknown names, like the reference
        self.scalers = [SCALERS[s] for s in scalers]
        self.edge_dim = edge_dim
        self.towers = towers
        self.divide_input = divide_input
        self.F_in = in_channels // towers if divide_input else in_channels
        self.F_out = out_channels // towers
        self.avg_deg = _avg_deg(deg)
        if self.edge_dim is not None:
            self.edge_encoder = Linear(edge_dim, self.F_in)
        self.pre_nns = ModuleList()
        self.post_nns = ModuleList()
        for _ in range(towers):
            self.pre_nns.append(_mlp((3 if edge_dim else 2) * self.F_in, self.F_in, pre_layers))
            self.post_nns.append(_mlp((len(aggregators) * len(scalers) + 1) * self.F_in, self.F_out, post_layers))
        self.lin = Linear(out_channels, out_channels)

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, Linear):
                m.reset_parameters()

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr: Optional[Tensor] = None) -> Tensor:
        V, T, Fi = x.shape[0], self.towers, self.F_in
        graph = _graph_of(edge_index, V)
        csr = graph.csr
        names = [_KERNEL_NAME[a] for a in self.aggregator_names]
        A, S = len(names), len(self.scaler_names)
        xs = [x[:, t * Fi:(t + 1) * Fi] if self.divide_input else x for t in range(T)]
        if (edge_attr is None) != (self.edge_dim is None):
            raise RuntimeError("edge_attr must be given exactly when the layer was built with edge_dim")   # the reference fails too
        use_edge = edge_attr is not None
        enc = self.edge_encoder(edge_attr)[csr.eid] if use_edge else None      # (E, F_in), CSR order, shared by towers
        if all(len(nn) == 1 for nn in self.pre_nns):
            # message(j -> i) = W [x_i | x_j | enc(e)] + b = (W_i x_i + b) + W_j x_j + W_e enc(e)
            W = torch.stack([nn[0].weight for nn in self.pre_nns])              # (T, Fi, (2|3) Fi)
            b = torch.stack([nn[0].bias for nn in self.pre_nns])
            Wi, Wj, We = W[:, :, :Fi], W[:, :, Fi:2 * Fi], W[:, :, 2 * Fi:]
            if self.divide_input:
                xv = x.reshape(V, T, Fi)
                x_dst = (torch.einsum("vti,tfi->vtf", xv, Wi) + b).reshape(V, T * Fi)
                x_src = torch.einsum("vti,tfi->vtf", xv, Wj).reshape(V, T * Fi)
            else:
                x_dst = torch.addmm(b.reshape(-1), x, Wi.reshape(T * Fi, Fi).t())
                x_src = x @ Wj.reshape(T * Fi, Fi).t()
            x_edge = enc @ We.reshape(T * Fi, Fi).t() if use_edge else None
            agg = PF.aggregate(graph, x_src, Fi, names, n_tower=T, dst_term=x_dst, edge_term=x_edge)
        else:
            src, dst = csr.col.long(), csr.row.long()
            msgs = []
            for t, nn in enumerate(self.pre_nns):
                z = [xs[t][dst], xs[t][src]] + ([enc] if use_edge else [])
                msgs.append(nn(torch.cat(z, dim=1)))
            agg = PF.aggregate(graph, torch.cat(msgs, dim=1) if T > 1 else msgs[0], Fi, names, n_tower=T, edge_resident=True)
        factors, deg = _row_factors(graph, self.scaler_names, self.avg_deg)
        agg = fix_empty_std(agg, names, 1, Fi, deg)
        K = A * Fi
        outs = []
        for t, nn in enumerate(self.post_nns):
            y = PF.posttrans(agg[:, t * K:(t + 1) * K], K, nn[0].weight, nn[0].bias, factors, h_self=xs[t])
            for m in list(nn)[1:]:
                y = m(y)
            outs.append(y)
        return self.lin(torch.cat(outs, dim=1) if T > 1 else outs[0])

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels}, towers={self.towers})"


class PNAConvSimple(torch.nn.Module):
    """The simple layer of the MolHIV example (pna.py:167-253): messages are the raw source features."""

    def __init__(self, in_channels: int, out_channels: int, aggregators: List[str], scalers: List[str], deg: Tensor,
                 post_layers: int = 1, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.aggregator_names = list(aggregators)
        self.scaler_names = list(scalers)
        self.aggregators = [AGGREGATORS[a] for a in aggregators]
        self.scalers = [SCALERS[s] for s in scalers]
        self.F_in, self.F_out = in_channels, out_channels
        self.avg_deg = _avg_deg(deg)
        self.post_nn = _mlp(len(aggregators) * len(scalers) * self.F_in, self.F_out, post_layers)

    def reset_parameters(self):
        for m in self.post_nn:
            if isinstance(m, Linear):
                m.reset_parameters()

    def aggregate(self, x: Tensor, edge_index: Tensor) -> Tensor:
        """The (V, S*A*F) tensor `propagate` hands to post_nn (pna.py:241-251), materialised (forward() does not)."""
        graph = _graph_of(edge_index, x.shape[0])
        names = [_KERNEL_NAME[a] for a in self.aggregator_names]
        factors, deg = _row_factors(graph, self.scaler_names, self.avg_deg)
        out = PF.aggregate(graph, graph.source_features(x), self.F_in, names, row_scales=factors)
        return fix_empty_std(out, names, len(factors), self.F_in, deg, factors)

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr: Optional[Tensor] = None) -> Tensor:
        graph = _graph_of(edge_index, x.shape[0])
        names = [_KERNEL_NAME[a] for a in self.aggregator_names]
        agg = PF.aggregate(graph, graph.source_features(x), self.F_in, names)       # identity scaler only
        factors, deg = _row_factors(graph, self.scaler_names, self.avg_deg)
        agg = fix_empty_std(agg, names, 1, self.F_in, deg)
        y = PF.posttrans(agg, len(names) * self.F_in, self.post_nn[0].weight, self.post_nn[0].bias, factors)
        for m in list(self.post_nn)[1:]:
            y = m(y)
        return y

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels})"
