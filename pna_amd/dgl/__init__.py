"""MI355X-native replacements for the reference's DGL-variant PNA modules (models/dgl/)."""
from .pna_layer import PNALayer, PNASimpleLayer, PNATower  # noqa: F401
from .aggregators import AGGREGATORS  # noqa: F401
from .scalers import SCALERS  # noqa: F401
