"""PyG-compatible front end (models/pytorch_geometric of the reference): `PNAConv`, `PNAConvSimple` and the two
registries, taking `(x, edge_index[, edge_attr])` instead of a DGL graph.  No torch_geometric dependency."""
from .aggregators import AGGREGATORS  # noqa: F401
from .pna import PNAConv, PNAConvSimple  # noqa: F401
from .scalers import SCALERS  # noqa: F401
