"""PNATower / PNALayer of the dense-adjacency variant -- drop-in for models/pytorch/pna/layer.py.

Same constructor / forward signatures, assertions and state_dict keys
(`towers.{t}.pretrans...`, `towers.{t}.posttrans...`, `mixing_network.linear.*`).  Instead of
materialising the (B, N, N, 2F) pair tensor and reducing it four times (layer.py:37-44), forward
sparsifies `adj` once (cached across the N/2 layer calls that share it, gnn_framework.py:93-95),
factorises the 1-layer pretrans to node level and runs the HIP segment-reduce kernel over the two
CSR groupings the dense aggregators imply (see sparsify.py), then the MFMA posttrans kernel.
input:(B,N,F_in) fp32 and adj:(B,N,N) fp32 must be GPU tensors.
"""
import torch
import torch.nn as nn

from ... import functional as PF
from ...layers import MLP, FCLayer
from .aggregators import AGGREGATORS
from .scalers import SCALERS, row_factor
from .sparsify import sparsify

_BY_ROW = ("mean", "sum", "std", "var")     # reduce over j, adjacency as weight
_BY_COL = ("max", "min")                    # reduce over i, adjacency as mask


class PNATower(nn.Module):
    def __init__(self, in_features, out_features, aggregators, scalers, avg_d, self_loop, pretrans_layers,
                 posttrans_layers, device):
        super().__init__()
        self.device = device
        self.in_features, self.out_features = in_features, out_features
        self.aggregators = [a if isinstance(a, str) else a.__name__.replace("aggregate_", "") for a in aggregators]
        self.scalers = [s if isinstance(s, str) else s.__name__.replace("scale_", "") for s in scalers]
        self.self_loop = self_loop
        self.avg_d = avg_d
        self.pretrans = MLP(in_size=2 * in_features, hidden_size=in_features, out_size=in_features,
                            layers=pretrans_layers, mid_activation="relu", last_activation="none")
        self.posttrans = MLP(in_size=(len(self.aggregators) * len(self.scalers) + 1) * in_features,
                             hidden_size=out_features, out_size=out_features, layers=posttrans_layers,
                             mid_activation="relu", last_activation="none")

    def forward(self, input, adj):
        return _dense_towers_forward([self], input, adj, divide_input=False)

    def __repr__(self):
        return f"{self.__class__.__name__} ({self.in_features} -> {self.out_features})"


def _dense_towers_forward(towers, x, adj, divide_input):
    t0 = towers[0]
    B, N, _ = adj.shape
    T, Fi = len(towers), t0.in_features
    V = B * N
    dg = sparsify(adj, t0.self_loop)
    xf = x.reshape(V, -1)
    hs = [xf[:, t * Fi:(t + 1) * Fi] if divide_input else xf for t in range(T)]
    rows = [a for a in t0.aggregators if a in _BY_ROW]
    cols = [a for a in t0.aggregators if a in _BY_COL]
    w_row = None if dg.binary else dg.w_row
    w_col = None if dg.binary else dg.w_col

    exotic = [a for a in t0.aggregators if a not in _BY_ROW and a not in _BY_COL]
    if exotic:
        return _dense_towers_forward_registry(towers, x, adj, divide_input)
    out_row = out_col = None
    if all(t.pretrans.is_affine for t in towers):
        # m[b,i,j] = W [x_i | x_j] + b = P[i] + Q[j]   with P = W[:, :F] x,  Q = W[:, F:] x + b
        W = torch.stack([t.pretrans.fully_connected[0].linear.weight for t in towers])
        b = torch.stack([t.pretrans.fully_connected[0].linear.bias for t in towers])
        Wp, Wq = W[:, :, :Fi], W[:, :, Fi:]
        if divide_input:
            hv = xf.reshape(V, T, Fi)
            P = torch.einsum("vti,tfi->vtf", hv, Wp).reshape(V, T * Fi)
            Q = (torch.einsum("vti,tfi->vtf", hv, Wq) + b).reshape(V, T * Fi)
        else:
            P = xf @ Wp.reshape(T * Fi, Fi).t()
            Q = torch.addmm(b.reshape(-1), xf, Wq.reshape(T * Fi, Fi).t())
        if rows:    # node i over its row: gather Q[j], add P[i]
            out_row = PF.aggregate(dg.by_row, Q, Fi, rows, n_tower=T, dst_term=P, edge_weight=w_row)
        if cols:    # node j over its column: gather P[i], add Q[j]
            out_col = PF.aggregate(dg.by_col, P, Fi, cols, n_tower=T, dst_term=Q, edge_weight=w_col)
    else:
        ci, cj = dg.by_row.csr.row.long(), dg.by_row.csr.col.long()         # edge (b,i,j): dst i, src j
        msgs = torch.cat([t.pretrans(torch.cat([hs[k][ci], hs[k][cj]], dim=1)) for k, t in enumerate(towers)], dim=1)
        if rows:
            out_row = PF.aggregate(dg.by_row, msgs, Fi, rows, n_tower=T, edge_resident=True, edge_weight=w_row)
        if cols:    # same messages, visited in by_col order through the row->col permutation
            out_col = PF.aggregate(dg.by_col, msgs, Fi, cols, n_tower=T, edge_weight=w_col, col_override=dg.row_to_col)

    # interleave the two results into the caller's aggregator order: (V, T, A, Fi)
    parts = []
    if out_row is not None:
        out_row = out_row.view(V, T, len(rows), Fi)
    if out_col is not None:
        out_col = out_col.view(V, T, len(cols), Fi)
    for a in t0.aggregators:
        parts.append(out_row[:, :, rows.index(a)] if a in rows else out_col[:, :, cols.index(a)])
    A = len(parts)
    agg = torch.stack(parts, dim=2).reshape(V, T * A * Fi)

    scales = []
    for s in t0.scalers:
        f = row_factor(s, adj, t0.avg_d)
        scales.append(None if f is None else f.reshape(V).contiguous())
    K = A * Fi
    outs = []
    for t, tower in enumerate(towers):
        lin = tower.posttrans.fully_connected[0].linear
        y = PF.posttrans(agg[:, t * K:(t + 1) * K], K, lin.weight, lin.bias, scales, h_self=hs[t])
        outs.append(tower.posttrans.tail(y))
    y = torch.cat(outs, dim=1) if T > 1 else outs[0]
    return y.view(B, N, -1)


def _dense_towers_forward_registry(towers, x, adj, divide_input):
    """Aggregator sets beyond mean/sum/max/min/std/var (softmax, softmin, moments, normalised_mean, identity -- used
    by no shipped configuration): the (B,N,N,F) message tensor is materialised like the reference does
    (layer.py:37-40) and every registry operator reduces it through the HIP kernel."""
    B, N, _ = adj.shape
    outs = []
    for t, tower in enumerate(towers):
        Fi = tower.in_features
        xt = x[:, :, t * Fi:(t + 1) * Fi] if divide_input else x
        h_cat = torch.cat([xt.unsqueeze(2).expand(B, N, N, Fi), xt.unsqueeze(1).expand(B, N, N, Fi)], dim=3)
        h_mod = tower.pretrans(h_cat)
        m = torch.cat([AGGREGATORS[a](h_mod, adj, self_loop=tower.self_loop, device=x.device) for a in tower.aggregators], dim=2)
        m = torch.cat([SCALERS[s](m, adj, avg_d=tower.avg_d) for s in tower.scalers], dim=2)
        outs.append(tower.posttrans(torch.cat([xt, m], dim=2)))
    return torch.cat(outs, dim=2) if len(outs) > 1 else outs[0]


class PNALayer(nn.Module):
    """A single PNA convolution on dense adjacency (https://arxiv.org/abs/2004.05718)."""

    def __init__(self, in_features, out_features, aggregators, scalers, avg_d, towers=1, self_loop=False,
                 pretrans_layers=1, posttrans_layers=1, divide_input=True, device="cpu"):
        super().__init__()
        assert ((not divide_input) or in_features % towers == 0), "if divide_input is set the number of towers has to divide in_features"
        assert (out_features % towers == 0), "the number of towers has to divide the out_features"
        for a in aggregators:
            AGGREGATORS[a]                  # KeyError on unknown / not-yet-supported names
        for s in scalers:
            SCALERS[s]
        self.in_features, self.out_features = in_features, out_features
        self.divide_input = divide_input
        self.input_tower = in_features // towers if divide_input else in_features
        self.output_tower = out_features // towers
        self.towers = nn.ModuleList(
            PNATower(in_features=self.input_tower, out_features=self.output_tower, aggregators=list(aggregators),
                     scalers=list(scalers), avg_d=avg_d, self_loop=self_loop, pretrans_layers=pretrans_layers,
                     posttrans_layers=posttrans_layers, device=device) for _ in range(towers))
        self.mixing_network = FCLayer(out_features, out_features, activation="LeakyReLU")

    def forward(self, input, adj):
        y = _dense_towers_forward(list(self.towers), input, adj, self.divide_input)
        return self.mixing_network(y)

    def __repr__(self):
        return f"{self.__class__.__name__} ({self.in_features} -> {self.out_features})"
