"""Dense blocks the PNA layers are assembled from: `FCLayer` and `MLP`.

Same constructor arguments, forward order (linear -> activation -> dropout -> batch-norm), weight
initialisation and state_dict keys (`fully_connected.{k}.linear.{weight,bias}`) as the reference's
models/layers.py:101-234, so checkpoints are interchangeable.  These are the plain library GEMMs
of the path (torch.nn.Linear -> rocBLAS/hipBLASLt); the hot contraction after the aggregation goes
through pna_amd.ops.posttrans instead.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

_ACTIVATIONS = {
    "relu": nn.ReLU, "sigmoid": nn.Sigmoid, "tanh": nn.Tanh, "elu": nn.ELU, "selu": nn.SELU, "glu": nn.GLU,
    "leakyrelu": nn.LeakyReLU, "softplus": nn.Softplus,
}


def get_activation(activation):
    """String -> module, case-insensitive; None / 'none' -> None; callables pass through (models/layers.py:8-19)."""
    if activation is None or callable(activation):
        return activation
    key = str(activation).lower()
    if key == "none":
        return None
    assert key in _ACTIVATIONS, 'Unhandled activation function'   # same failure mode as the reference
    return _ACTIVATIONS[key]()


class FCLayer(nn.Module):
    def __init__(self, in_size, out_size, activation="relu", dropout=0., b_norm=False, bias=True, init_fn=None,
                 device="cpu"):
        super().__init__()
        self.in_size, self.out_size, self.bias = in_size, out_size, bias
        self.linear = nn.Linear(in_size, out_size, bias=bias).to(device)
        self.dropout = nn.Dropout(p=dropout) if dropout else None
        self.b_norm = nn.BatchNorm1d(out_size).to(device) if b_norm else None
        self.activation = get_activation(activation)
        self.init_fn = init_fn or nn.init.xavier_uniform_
        self.reset_parameters()

    def reset_parameters(self, init_fn=None):
        (init_fn or self.init_fn)(self.linear.weight, 1 / self.in_size)     # gain = 1/in_size (:170-179)
        if self.bias:
            self.linear.bias.data.zero_()

    def forward(self, x):
        h = self.linear(x)
        if self.activation is not None:
            h = self.activation(h)
        if self.dropout is not None:
            h = self.dropout(h)
        if self.b_norm is not None:
            h = self.b_norm(h.transpose(1, 2)).transpose(1, 2) if h.shape[1] != self.out_size else self.b_norm(h)
        return h

    def __repr__(self):
        return f"{self.__class__.__name__} ({self.in_size} -> {self.out_size})"


class MLP(nn.Module):
    """`layers` FCLayers; a single layer uses `last_activation` (models/layers.py:214-216)."""

    def __init__(self, in_size, hidden_size, out_size, layers, mid_activation="relu", last_activation="none",
                 dropout=0., mid_b_norm=False, last_b_norm=False, device="cpu"):
        super().__init__()
        self.in_size, self.hidden_size, self.out_size = in_size, hidden_size, out_size
        sizes = [in_size] + [hidden_size] * (max(layers, 1) - 1) + [out_size]
        self.fully_connected = nn.ModuleList()
        for i in range(len(sizes) - 1):
            last = i == len(sizes) - 2
            self.fully_connected.append(FCLayer(sizes[i], sizes[i + 1],
                                                activation=last_activation if last else mid_activation,
                                                b_norm=last_b_norm if last else mid_b_norm, device=device,
                                                dropout=dropout))

    def forward(self, x):
        for fc in self.fully_connected:
            x = fc(x)
        return x

    def __repr__(self):
        return f"{self.__class__.__name__} ({self.in_size} -> {self.out_size})"

    # -- helpers for the fused paths -----------------------------------------------------------
    @property
    def is_affine(self):
        """True when the MLP is a single Linear with no activation / dropout / batch-norm, i.e. the
        form a 1-layer pretrans / posttrans takes (last_activation='none')."""
        if len(self.fully_connected) != 1:
            return False
        fc = self.fully_connected[0]
        return fc.activation is None and fc.b_norm is None and (fc.dropout is None or not self.training)

    def tail(self, x):
        """Everything after the first Linear (its activation, dropout, batch-norm and the later layers)."""
        fc = self.fully_connected[0]
        if fc.activation is not None:
            x = fc.activation(x)
        if fc.dropout is not None:
            x = fc.dropout(x)
        if fc.b_norm is not None:
            x = fc.b_norm(x)
        for fc in self.fully_connected[1:]:
            x = fc(x)
        return x
