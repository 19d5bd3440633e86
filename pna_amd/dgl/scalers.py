"""Degree scalers of the DGL variant: SCALERS[name](h, D:int, avg_d) -> h * factor(D).

Keeps the registry interface of the reference (models/dgl/scalers.py:22) and its rounding sequence
(models/dgl/scalers.py:12-19: log(D+1) is taken in float64 by numpy and meets the 0-dim fp32 tensor
avg_d['log'] through Tensor.__rtruediv__ / __truediv__).  The fused layers never call these per
degree bucket: they use the per-row factor arrays of Graph.degree_scalers
(pna_degree_scalers_f32), which reproduce exactly these values for every in-degree.
"""
import numpy as np


def degree_factor(name, D, avg_d):
    """The fp32 value (as a Python float) the scaler `name` multiplies a bucket of in-degree D with.

    The reference evaluates `np.log(D+1) / avg` as avg.reciprocal() * fp32(log(D+1)) and
    `avg / np.log(D+1)` as a true fp32 division (torch's scalar dispatch).  The same sequence is done
    here in numpy float32 on the host so that the factor is identical whether avg_d['log'] lives on the
    CPU or on a GPU (a device-side reciprocal may differ from the CPU's by one ulp)."""
    avg = np.float32(float(avg_d["log"]))
    lg = np.float32(np.log(D + 1))
    if name == "amplification":
        return float(np.float32(np.float32(1.0) / avg) * lg)
    if name == "attenuation":
        return float(avg / lg)
    raise KeyError(name)


def _scaler(name):
    def scale(h, D=None, avg_d=None):
        return h if name == "identity" else h * degree_factor(name, D, avg_d)
    scale.__name__ = "scale_" + name
    return scale


SCALERS = {name: _scaler(name) for name in ("identity", "amplification", "attenuation")}
