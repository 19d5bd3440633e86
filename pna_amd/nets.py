"""Network assembly of the realworld (sparse) benchmarks -- SURVEY.md 8f row N2.

`PNANet` mirrors realworld_benchmark/nets/molecules_graph_regression/pna_net.py:16-96 (same `net_params`
dict, forward signature and state_dict keys): atom/bond embeddings -> L x PNALayer -> per-graph readout ->
MLPReadout.  `PNANetHIV` mirrors nets/HIV_graph_classification/pna_net.py:9-64 with an in-tree stand-in for
ogb's AtomEncoder (a sum of per-feature embeddings; ogb is not available offline).  `PNANetSuperpixels` mirrors
nets/superpixels_graph_classification/pna_net.py:17-104 (CIFAR10 / MNIST: Linear embeddings of float node / edge features,
MLPReadout to n_classes, cross-entropy).

The per-graph readout (`dgl.sum_nodes` / `mean_nodes` / `max_nodes`, pna_net.py:83-90) is the same HIP
segment-reduce kernel as the message passing: segments = batch_num_nodes, messages = the node rows themselves.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as PF
from .dgl.pna_layer import PNALayer, PNASimpleLayer, as_graph
from .graph import Graph


class MLPReadout(nn.Module):
    """nets/mlp_readout_layer.py:14-29: L halving Linear+ReLU layers, then Linear to output_dim."""

    def __init__(self, input_dim, output_dim, L=2):
        super().__init__()
        dims = [input_dim // 2 ** l for l in range(L + 1)]
        self.FC_layers = nn.ModuleList([nn.Linear(dims[l], dims[l + 1], bias=True) for l in range(L)] +
                                       [nn.Linear(dims[L], output_dim, bias=True)])
        self.L = L

    def forward(self, x):
        for l in range(self.L):
            x = F.relu(self.FC_layers[l](x))
        return self.FC_layers[self.L](x)


class GRU(nn.Module):
    """nets/gru.py:5-30: nn.GRU over a length-1 sequence with the previous features as hidden state."""

    def __init__(self, input_size, hidden_size, device):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.gru = nn.GRU(input_size=input_size, hidden_size=hidden_size).to(device)

    def forward(self, x, y):
        assert x.shape[-1] <= self.input_size and y.shape[-1] <= self.hidden_size
        return self.gru(x.unsqueeze(0), y.unsqueeze(0))[1].squeeze()


def readout_nodes(g, h, op):
    """(n_graphs, F): sum / mean / max of the node rows of every graph of the batch -- one segment-reduce launch."""
    graph = as_graph(g)
    key = ("readout", h.device)
    rg = graph._heavy.get(key)
    if rg is None:
        sizes = torch.tensor(graph.batch_num_nodes, dtype=torch.long, device=h.device)
        owner = torch.repeat_interleave(torch.arange(sizes.numel(), device=h.device), sizes)
        rg = Graph(torch.arange(graph.num_nodes, device=h.device), owner, sizes.numel())   # node v -> its graph
        graph._heavy[key] = rg
    return PF.aggregate(rg, h, h.shape[1], [op], edge_resident=True)


class PNANet(nn.Module):
    def __init__(self, net_params):
        super().__init__()
        p = net_params
        hidden_dim, out_dim, n_layers = p["hidden_dim"], p["out_dim"], p["L"]
        self.readout = p["readout"]
        self.edge_feat = p["edge_feat"]
        self.gru_enable = p["gru"]
        self.in_feat_dropout = nn.Dropout(p["in_feat_dropout"])
        self.embedding_h = nn.Embedding(p["num_atom_type"], hidden_dim)
        if self.edge_feat:
            self.embedding_e = nn.Embedding(p["num_bond_type"], p["edge_dim"])
        common = dict(dropout=p["dropout"], graph_norm=p["graph_norm"], batch_norm=p["batch_norm"], residual=p["residual"],
                      aggregators=p["aggregators"], scalers=p["scalers"], avg_d=p["avg_d"], towers=p["towers"],
                      edge_features=self.edge_feat, edge_dim=p["edge_dim"], pretrans_layers=p["pretrans_layers"],
                      posttrans_layers=p["posttrans_layers"])
        self.layers = nn.ModuleList([PNALayer(in_dim=hidden_dim, out_dim=hidden_dim, divide_input=p["divide_input_first"],
                                              **common) for _ in range(n_layers - 1)])
        self.layers.append(PNALayer(in_dim=hidden_dim, out_dim=out_dim, divide_input=p["divide_input_last"], **common))
        if self.gru_enable:
            self.gru = GRU(hidden_dim, hidden_dim, p["device"])
        self.MLP_layer = MLPReadout(out_dim, 1)

    def forward(self, g, h, e, snorm_n, snorm_e=None):
        graph = as_graph(g)
        h = self.in_feat_dropout(embed(self.embedding_h, h) if isinstance(self.embedding_h, nn.Embedding) else self.embedding_h(h))
        if self.edge_feat:
            e_idx = e
            e = embed(self.embedding_e, e)
            if e_idx.dim() == 1 and not torch.is_floating_point(e_idx) and hasattr(graph, "register_edge_types"):
                # the layers' edge-type fast paths (<= 4 bond types: ZINC) get the types they would otherwise have to FIND in e's rows
                graph.register_edge_types(e, e_idx, self.embedding_e.weight)
        for i, conv in enumerate(self.layers):
            h_t = conv(graph, h, e, snorm_n)
            if self.gru_enable and i != len(self.layers) - 1:
                h_t = self.gru(h, h_t)
            h = h_t
        hg = readout_nodes(graph, h, self.readout if self.readout in ("sum", "max", "mean") else "mean")
        return self.MLP_layer(hg)

    def loss(self, scores, targets):
        return nn.L1Loss()(scores, targets)


class PNANetSuperpixels(nn.Module):
    """nets/superpixels_graph_classification/pna_net.py:17-104: the same stack as the molecules net behind LINEAR embeddings of the
    float node features (mean colour | position: in_dim 3 for MNIST, 5 for CIFAR10) and edge features, `n_classes` outputs and a
    cross-entropy loss.  (Like the reference's forward :72-98, `in_feat_dropout` is read from the params and never applied.)
    Same `net_params` keys, forward signature and state_dict keys."""

    def __init__(self, net_params):
        super().__init__()
        p = net_params
        hidden_dim, out_dim, n_layers = p["hidden_dim"], p["out_dim"], p["L"]
        self.readout = p["readout"]
        self.edge_feat = p["edge_feat"]
        self.gru_enable = p["gru"]
        self.embedding_h = nn.Linear(p["in_dim"], hidden_dim)
        if self.edge_feat:
            self.embedding_e = nn.Linear(p["in_dim_edge"], p["edge_dim"])
        common = dict(dropout=p["dropout"], graph_norm=p["graph_norm"], batch_norm=p["batch_norm"], residual=p["residual"],
                      aggregators=p["aggregators"], scalers=p["scalers"], avg_d=p["avg_d"], towers=p["towers"],
                      edge_features=self.edge_feat, edge_dim=p["edge_dim"], pretrans_layers=p["pretrans_layers"],
                      posttrans_layers=p["posttrans_layers"])
        self.layers = nn.ModuleList([PNALayer(in_dim=hidden_dim, out_dim=hidden_dim, divide_input=p["divide_input_first"],
                                              **common) for _ in range(n_layers - 1)])
        self.layers.append(PNALayer(in_dim=hidden_dim, out_dim=out_dim, divide_input=p["divide_input_last"], **common))
        if self.gru_enable:
            self.gru = GRU(hidden_dim, hidden_dim, p["device"])
        self.MLP_layer = MLPReadout(out_dim, p["n_classes"])

    def forward(self, g, h, e, snorm_n, snorm_e=None):
        graph = as_graph(g)
        h = self.embedding_h(h)
        if self.edge_feat:
            e = self.embedding_e(e)
        for i, conv in enumerate(self.layers):
            h_t = conv(graph, h, e, snorm_n)
            if self.gru_enable and i != len(self.layers) - 1:
                h_t = self.gru(h, h_t)
            h = h_t
        hg = readout_nodes(graph, h, self.readout if self.readout in ("sum", "max", "mean") else "mean")
        return self.MLP_layer(hg)

    def loss(self, pred, label):
        return nn.CrossEntropyLoss()(pred, label)


class _SmallTableEmbedding(torch.autograd.Function):
    """weight[idx] whose backward is a one-hot GEMM, grad_w = onehot(idx)^T @ grad.  torch's embedding backward sorts the
    indices and runs a segmented scatter (0.4 ms per table for 52 k atoms -- with the nine tables of the atom encoder
    that was 44 % of a MolHIV training step) and `index_add_` into a table of 2..119 rows is one long chain of
    contended atomics (0.3 ms per table); the GEMM is ~0.03 ms."""

    @staticmethod
    def forward(ctx, weight, idx):
        ctx.save_for_backward(idx)
        ctx.rows = weight.shape[0]
        return weight.index_select(0, idx)

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        if ctx.rows <= 1024:
            gw = F.one_hot(idx, ctx.rows).to(grad.dtype).t() @ grad
        else:
            gw = torch.zeros(ctx.rows, grad.shape[1], dtype=grad.dtype, device=grad.device).index_add_(0, idx, grad.contiguous())
        return gw, None


def embed(emb: nn.Embedding, idx):
    """emb(idx) for the small categorical vocabularies of the molecule nets (same values, faster backward)."""
    if emb.padding_idx is not None or emb.max_norm is not None or idx.dim() != 1:
        return emb(idx)
    return _SmallTableEmbedding.apply(emb.weight, idx)


class AtomEncoderStandIn(nn.Module):
    """Sum of one embedding per categorical atom feature -- what ogb.graphproppred.mol_encoder.AtomEncoder computes
    (ogb 1.2.2 is pinned by the reference but not installable offline; dims are the OGB molecule vocabulary sizes)."""
    DIMS = (119, 4, 12, 12, 10, 6, 6, 2, 2)

    def __init__(self, emb_dim, dims=DIMS):
        super().__init__()
        self.atom_embedding_list = nn.ModuleList(nn.Embedding(d, emb_dim) for d in dims)
        for emb in self.atom_embedding_list:
            nn.init.xavier_uniform_(emb.weight.data)

    def forward(self, x):
        out = 0
        for i, emb in enumerate(self.atom_embedding_list):
            out = out + embed(emb, x[:, i])
        return out


class PNANetHIV(nn.Module):
    """nets/HIV_graph_classification/pna_net.py:9-64 (PNASimpleLayer stack, mean readout by default)."""

    def __init__(self, net_params):
        super().__init__()
        p = net_params
        hidden_dim, out_dim, n_layers = p["hidden_dim"], p["out_dim"], p["L"]
        self.readout = p["readout"]
        self.in_feat_dropout = nn.Dropout(p["in_feat_dropout"])
        self.embedding_h = AtomEncoderStandIn(emb_dim=hidden_dim)
        common = dict(dropout=p["dropout"], batch_norm=p["batch_norm"], residual=p["residual"], aggregators=p["aggregators"],
                      scalers=p["scalers"], avg_d=p["avg_d"], posttrans_layers=p["posttrans_layers"])
        self.layers = nn.ModuleList([PNASimpleLayer(in_dim=hidden_dim, out_dim=hidden_dim, **common)
                                     for _ in range(n_layers - 1)])
        self.layers.append(PNASimpleLayer(in_dim=hidden_dim, out_dim=out_dim, **common))
        self.MLP_layer = MLPReadout(out_dim, 1)

    def forward(self, g, h):
        graph = as_graph(g)
        h = self.in_feat_dropout(embed(self.embedding_h, h) if isinstance(self.embedding_h, nn.Embedding) else self.embedding_h(h))
        for conv in self.layers:
            h = conv(graph, h)
        hg = readout_nodes(graph, h, self.readout if self.readout in ("sum", "max", "mean") else "mean")
        return self.MLP_layer(hg)

    def loss(self, scores, labels):
        return nn.BCEWithLogitsLoss()(scores, labels.float().to(scores.device).unsqueeze(-1))
