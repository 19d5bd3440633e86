"""Degree scalers of the dense variant: SCALERS[name](X:(B,N,D), adj:(B,N,N), avg_d) -> (B,N,D).

Registry interface of models/pytorch/pna/scalers.py:41-42.  Degrees are row sums of the (weighted)
adjacency, D = adj.sum(-1) (:13,:21,:29,:36).  `row_factor` returns the per-node multiplier; the
layer hands it to the MFMA posttrans kernel as a per-row scale so the scaled copies of the
aggregate are never materialised.
"""
import torch


def row_factor(name, adj, avg_d):
    """(B, N) multiplier of scaler `name`, or None for identity."""
    if name == "identity":
        return None
    D = torch.sum(adj, -1)
    if name == "amplification":
        return torch.log(D + 1) / avg_d["log"]
    if name == "attenuation":
        return avg_d["log"] / torch.log(D + 1)
    if name == "linear":
        return D / avg_d["lin"]
    if name == "inverse_linear":
        return avg_d["lin"] / D
    raise KeyError(name)


def _scaler(name):
    def scale(X, adj, avg_d=None):
        f = row_factor(name, adj, avg_d)
        return X if f is None else f.unsqueeze(-1) * X
    scale.__name__ = "scale_" + name
    return scale


SCALERS = {n: _scaler(n) for n in ("identity", "amplification", "attenuation", "linear", "inverse_linear")}
