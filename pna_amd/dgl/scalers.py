"""Degree scalers of the DGL variant: SCALERS[name](h, D:int, avg_d) -> h * factor(D).

Keeps the registry interface of the reference (models/dgl/scalers.py:22) and its rounding sequence
(models/dgl/scalers.py:12-19: log(D+1) is taken in float64 by numpy and meets the 0-dim fp32 tensor
avg_d['log'] through Tensor.__rtruediv__ / __truediv__).  The fused layers never call these per
degree bucket: they use the per-row factor arrays of Graph.degree_scalers
(pna_degree_scalers_f32), which reproduce exactly these values for every in-degree.
"""
import numpy as np


def degree_factor(name, D, avg_d):
    """Scalar (0-dim fp32 tensor) the scaler `name` multiplies a bucket of in-degree D with."""
    if name == "amplification":
        return np.log(D + 1) / avg_d["log"]       # -> avg.reciprocal() * fp32(log(D+1))
    if name == "attenuation":
        return avg_d["log"] / np.log(D + 1)       # -> avg / fp32(log(D+1))
    raise KeyError(name)


def _scaler(name):
    def scale(h, D=None, avg_d=None):
        return h if name == "identity" else h * degree_factor(name, D, avg_d)
    scale.__name__ = "scale_" + name
    return scale


SCALERS = {name: _scaler(name) for name in ("identity", "amplification", "attenuation")}
