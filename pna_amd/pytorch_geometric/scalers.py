"""SCALERS of the PyG formulation -- models/pytorch_geometric/scalers.py: `fn(src, deg, avg_deg)` with `deg` a float
tensor broadcastable against `src` and avg_deg = {'lin', 'log', 'exp'}.  The only place where the reference defines
nodes without in-edges: attenuation / inverse_linear scale 1 there (:16-19, :26-29), amplification / linear 0."""
import torch


def row_factor(name, deg, avg_deg):
    """Per-row multiplier (same shape as `deg`), None for the identity -- the expression order of the reference."""
    if name == "identity":
        return None
    if name == "amplification":
        return torch.log(deg + 1) / avg_deg["log"]
    if name == "attenuation":
        scale = avg_deg["log"] / torch.log(deg + 1)
        scale[deg == 0] = 1
        return scale
    if name == "linear":
        return deg / avg_deg["lin"]
    if name == "inverse_linear":
        scale = avg_deg["lin"] / deg
        scale[deg == 0] = 1
        return scale
    raise KeyError(name)


def _make(name):
    def scale(src, deg, avg_deg):
        f = row_factor(name, deg, avg_deg)
        return src if f is None else src * f
    scale.__name__ = "scale_" + name
    return scale


SCALERS = {name: _make(name) for name in ("identity", "amplification", "attenuation", "linear", "inverse_linear")}
