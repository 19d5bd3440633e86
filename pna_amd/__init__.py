"""pna_amd -- MI355X (gfx950) native implementation of the PNA message-passing layer of
lukecavabarrett/pna: hand-written HIP kernels behind the reference's own layer API.

    from pna_amd.dgl.pna_layer import PNALayer, PNASimpleLayer        # <- models/dgl/pna_layer.py
    from pna_amd.pytorch.pna.layer import PNALayer                     # <- models/pytorch/pna/layer.py
    from pna_amd import Graph                                          # <- the DGLGraph the layers consume
"""
from .graph import Graph, avg_d_from_adjacency, avg_d_from_degrees  # noqa: F401

__version__ = "0.1.0"
