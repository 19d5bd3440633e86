"""PNATower / PNALayer / PNASimpleLayer -- drop-in replacements for models/dgl/pna_layer.py.

Constructor signatures, forward signatures, assertion behaviour and state_dict keys are the
reference's; what happens inside forward is not:

  reference (pna_layer.py)                                this module
  ------------------------------------------------------  ------------------------------------------
  apply_edges(pretrans) on cat[h_src|h_dst|ef]  (:35-40)  1-layer pretrans is affine, so it is split
    -> an (E, 2F+ed)x(.,F) GEMM per tower                 into node-level projections A=W_a h,
                                                          B=W_b h+b (and C=W_e ef per edge); the
                                                          per-edge message A[src]+B[dst]+C is formed
                                                          inside the gather kernel, never in HBM
  update_all -> per in-degree bucket, per aggregator,     ONE pna_segreduce_fwd_f32 launch for all
    per scaler torch ops, per tower (:45-50,:64)          towers, aggregators and degrees
  cat[h, 12F aggregate] -> posttrans Linear (:65-68)      pna_posttrans_f32: fp32 MFMA contraction of
                                                          the 4F identity aggregate with the three
                                                          scaler weight blocks, scalers applied per
                                                          row in the epilogue (12F never materialised)

`g` is a pna_amd.Graph (or anything with .edges()/.number_of_nodes(), e.g. a DGLGraph, which is
adapted and cached).  GPU tensors only: there is no CPU path.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as PF
from .. import ops
from ..graph import Graph
from ..layers import MLP, FCLayer
from .aggregators import AGGREGATORS
from .scalers import SCALERS


def _names(items, registry, kind):
    """Accept registry names or the registry's own callables (the reference hands functions to PNATower)."""
    out = []
    for it in items:
        if isinstance(it, str):
            registry[it]                       # KeyError for unknown names, like the reference's lookup
            out.append(it)
            continue
        hit = [k for k, v in registry.items() if v is it]
        if not hit:
            name = getattr(it, "__name__", "")
            hit = [k for k in registry if name in (f"aggregate_{k}", f"scale_{k}")]
        if not hit:
            raise KeyError(f"unknown {kind}: {it!r}")
        out.append(hit[0])
    return out


def as_graph(g) -> Graph:
    if isinstance(g, Graph):
        return g
    cached = getattr(g, "_pna_amd_graph", None)
    if cached is None:
        cached = Graph.from_dgl(g)
        try:
            g._pna_amd_graph = cached
        except AttributeError:
            pass
    return cached


def _avg_log_value(avg_d):
    """avg_d['log'] as a Python float, read from the device ONCE per tensor object/version (a .item() per forward
    would put a device->host synchronisation into every layer call and make the forward un-capturable in a
    hipGraph).  The cache lives on the tensor object itself, never keyed by address (addresses are recycled)."""
    t = avg_d["log"]
    if not torch.is_tensor(t):
        return float(t)
    hit = getattr(t, "_pna_amd_float", None)
    if hit is None or hit[0] != (t._version, t.data_ptr(), str(t.device)):
        hit = ((t._version, t.data_ptr(), str(t.device)), float(t))
        t._pna_amd_float = hit
    return hit[1]


def _row_scales(graph, scalers, avg_d, device):
    if all(s == "identity" for s in scalers):
        return [None] * len(scalers)
    if graph.device != device:
        raise RuntimeError("graph and features live on different devices; call graph.to(device) first")
    amp, att = graph.degree_scalers(_avg_log_value(avg_d))
    return [{"identity": None, "amplification": amp, "attenuation": att}[s] for s in scalers]


class PNATower(nn.Module):
    def __init__(self, in_dim, out_dim, dropout, graph_norm, batch_norm, aggregators, scalers, avg_d,
                 pretrans_layers, posttrans_layers, edge_features, edge_dim):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.dropout = dropout
        self.graph_norm = graph_norm
        self.batch_norm = batch_norm
        self.edge_features = edge_features
        self.edge_dim = edge_dim if edge_features else 0
        self.aggregators = _names(aggregators, AGGREGATORS, "aggregator")
        self.scalers = _names(scalers, SCALERS, "scaler")
        self.avg_d = avg_d

        self.batchnorm_h = nn.BatchNorm1d(out_dim)
        self.pretrans = MLP(in_size=2 * in_dim + self.edge_dim, hidden_size=in_dim, out_size=in_dim,
                            layers=pretrans_layers, mid_activation="relu", last_activation="none")
        self.posttrans = MLP(in_size=(len(self.aggregators) * len(self.scalers) + 1) * in_dim, hidden_size=out_dim,
                             out_size=out_dim, layers=posttrans_layers, mid_activation="relu",
                             last_activation="none")

    # the stages below are shared with PNALayer, which runs them for all towers at once -------------
    def tail(self, y, snorm_n):
        """Everything after the first posttrans Linear (pna_layer.py:68-75)."""
        y = self.posttrans.tail(y)
        if self.graph_norm:
            y = y * snorm_n
        if self.batch_norm:
            y = self.batchnorm_h(y)
        return F.dropout(y, self.dropout, training=self.training)

    def forward(self, g, h, e, snorm_n):
        return _towers_forward([self], as_graph(g), h, e, snorm_n, divide_input=False)


def _projection_cache(towers, Fi):
    """([W_a ; W_b] (2*T*Fi, Fi), [0 ; b] (2*T*Fi)) of the towers' 1-layer pretrans, cached on the first tower per parameter
    (version, address): one GEMM then gives the source-side rows W_a h (columns [0, T*Fi)) and the destination-side rows
    W_b h + b (columns [T*Fi, 2*T*Fi)) of every tower."""
    lins = [t.pretrans.fully_connected[0].linear for t in towers]
    key = tuple((p._version, p.data_ptr(), str(p.device)) for l in lins for p in (l.weight, l.bias))
    hit = towers[0].__dict__.get("_pna_amd_proj")
    if hit is None or hit[0] != key:
        with torch.no_grad():
            W = torch.stack([l.weight for l in lins])                                        # (T, Fi, 2Fi + ed)
            T = len(lins)
            Wcat = torch.cat([W[:, :, :Fi].reshape(T * Fi, Fi), W[:, :, Fi:2 * Fi].reshape(T * Fi, Fi)], dim=0).contiguous()
            b = torch.stack([l.bias for l in lins]).reshape(-1)
            bcat = torch.cat([torch.zeros_like(b), b]).contiguous()
        hit = (key, Wcat, bcat)
        towers[0].__dict__["_pna_amd_proj"] = hit
    return hit[1], hit[2]


def _projection_cache_padded(tower, Fi, P):
    """_projection_cache for ONE tower with each half padded to P columns (zero weight rows): x_cat = [x_src | 0 | x_dst | 0],
    rows and halves 16-byte aligned -- the table the one-kernel tower layer reads in 16-byte strips."""
    lin = tower.pretrans.fully_connected[0].linear
    key = tuple((p._version, p.data_ptr(), str(p.device)) for p in (lin.weight, lin.bias)) + (P,)
    hit = tower.__dict__.get("_pna_amd_proj_pad")
    if hit is None or hit[0] != key:
        with torch.no_grad():
            W = torch.zeros(2 * P, Fi, dtype=lin.weight.dtype, device=lin.weight.device)
            b = torch.zeros(2 * P, dtype=lin.weight.dtype, device=lin.weight.device)
            W[:Fi], W[P:P + Fi], b[P:P + Fi] = lin.weight[:, :Fi], lin.weight[:, Fi:2 * Fi], lin.bias
        hit = (key, W.contiguous(), b.contiguous())
        tower.__dict__["_pna_amd_proj_pad"] = hit
    return hit[1], hit[2]


def _projection_cache_padded_div(towers, Fi, P):
    """_projection_cache_padded for T towers with divide_input=True (models/dgl/pna_layer.py:133-136): tower t projects the input
    slice [t Fi, (t+1) Fi) -- ONE block-diagonal GEMM gives x_cat = [x_src (T Fi) | 0 | x_dst (T Fi) | 0] with tower t's rows in
    columns [t Fi, (t+1) Fi) of each half."""
    lins = [t.pretrans.fully_connected[0].linear for t in towers]
    key = tuple((p._version, p.data_ptr(), str(p.device)) for l in lins for p in (l.weight, l.bias)) + (P,)
    hit = towers[0].__dict__.get("_pna_amd_proj_pad_div")
    if hit is None or hit[0] != key:
        T = len(lins)
        with torch.no_grad():
            W = torch.zeros(2 * P, T * Fi, dtype=lins[0].weight.dtype, device=lins[0].weight.device)
            b = torch.zeros(2 * P, dtype=W.dtype, device=W.device)
            for t, lin in enumerate(lins):
                r = slice(t * Fi, (t + 1) * Fi)
                W[r, r] = lin.weight[:, :Fi]
                W[P + t * Fi:P + (t + 1) * Fi, r] = lin.weight[:, Fi:2 * Fi]
                b[P + t * Fi:P + (t + 1) * Fi] = lin.bias
        hit = (key, W.contiguous(), b.contiguous())
        towers[0].__dict__["_pna_amd_proj_pad_div"] = hit
    return hit[1], hit[2]


def _projection_cache_padded_multi(towers, Fi, P):
    """The SOURCE half of _projection_cache for T towers over the whole input (divide_input=False) with every tower's block padded to P
    columns: x_src = [W_a,0 h | 0 | .. | W_a,T-1 h | 0], blocks 16-byte aligned -- the table FusedMultiTowerCall's launches read their
    tower's 16-byte strips from (the destination half W_b h + b is linear in the row's own features: folded into the dense term)."""
    lins = [t.pretrans.fully_connected[0].linear for t in towers]
    key = tuple((p._version, p.data_ptr(), str(p.device)) for l in lins for p in (l.weight, l.bias)) + (P,)
    hit = towers[0].__dict__.get("_pna_amd_proj_pad_multi")
    if hit is None or hit[0] != key:
        T = len(lins)
        with torch.no_grad():
            W = torch.zeros(T * P, Fi, dtype=lins[0].weight.dtype, device=lins[0].weight.device)
            for t, lin in enumerate(lins):
                W[t * P:t * P + Fi] = lin.weight[:, :Fi]
        hit = (key, W.t().contiguous(), W)                     # (Fi, T P): x_src = h @ it; (T P, Fi): pna_project_f32's layout
        towers[0].__dict__["_pna_amd_proj_pad_multi"] = hit
    return hit[1], hit[2]


def _towers_forward(towers, graph, h, e, snorm_n, divide_input):
    """Shared body of PNATower.forward / PNALayer.forward: all towers through one gather kernel."""
    t0 = towers[0]
    T, Fi, ed = len(towers), t0.in_dim, t0.edge_dim
    A, S = len(t0.aggregators), len(t0.scalers)
    V = h.shape[0]
    csr = graph.csr
    etab = None
    if t0.edge_features:
        if e is None:
            raise ValueError("edge_features=True but no edge features were given")
        no_grad = not torch.is_grad_enabled() or not (h.requires_grad or e.requires_grad or any(p.requires_grad for t in towers for p in t.parameters()))
        # (not while a hipGraph is being captured -- capture.GraphedForward with `e` as an input: the table is built from the VALUES of
        # e with host syncs, a replay with other values would gather through the captured example's types; the per-edge product
        # below is captured as kernels and replays correctly -- ADVICE r3)
        if (no_grad and e.is_cuda and type(graph) is Graph and all(t.pretrans.is_affine for t in towers)
                and (not torch.cuda.is_current_stream_capturing() or graph.edge_types_registered(e))):
            etab = graph.edge_type_table(e)                   # edge features that are an embedding of <= 4 edge types: a table
        e_csr = e[csr.eid] if etab is None else None          # per-edge features in CSR (dst-sorted) order
    hs = [h[:, t * Fi:(t + 1) * Fi] if divide_input else h for t in range(T)]

    if all(t.pretrans.is_affine for t in towers):
        # factorised pretrans: message(u->v) = W_a h_u + (W_b h_v + b) + W_e ef
        lins = [t.pretrans.fully_connected[0].linear for t in towers]
        inference = not torch.is_grad_enabled() or not (h.requires_grad or any(p.requires_grad for l in lins for p in l.parameters()))
        if inference and not divide_input and not t0.edge_features:
            W = b = Wa = Wb = We = None                     # (no per-call stacking launches: everything comes from the cache)
        else:
            W = torch.stack([l.weight for l in lins])       # (T, Fi, 2Fi+ed)
            b = torch.stack([l.bias for l in lins])         # (T, Fi)
            Wa, Wb, We = W[:, :, :Fi], W[:, :, Fi:2 * Fi], W[:, :, 2 * Fi:]
        if divide_input:
            hv = h.reshape(V, T, Fi)
            x_src = torch.einsum("vti,tfi->vtf", hv, Wa).reshape(V, T * Fi)
            x_dst = (torch.einsum("vti,tfi->vtf", hv, Wb) + b).reshape(V, T * Fi)
        elif inference:
            # inference: both projections of all towers as ONE GEMM against the cached [W_a ; W_b] (2*T*Fi x Fi) weight
            Wcat, bcat = _projection_cache(towers, Fi)
            if h.is_cuda and h.shape[0] >= PF.ops.X3_MIN_ROWS and Fi >= 4:
                x_cat = PF.linear_act(h, Wcat, bcat)      # large graphs: the contraction kernel beats the library GEMM 3x here
            else:
                x_cat = torch.addmm(bcat, h, Wcat.t())
            x_src, x_dst = x_cat[:, :T * Fi], x_cat[:, T * Fi:]
        else:
            x_src = h @ Wa.reshape(T * Fi, Fi).t()
            x_dst = torch.addmm(b.reshape(-1), h, Wb.reshape(T * Fi, Fi).t())
        x_edge = edge_type = None
        if t0.edge_features and etab is not None:             # W_e . ef of the <= 4 distinct feature rows; the gather indexes it by type
            edge_type, x_edge = etab[0], etab[1] @ We.reshape(T * Fi, ed).t()
        elif t0.edge_features:
            x_edge = e_csr @ We.reshape(T * Fi, ed).t()
        x_src = graph.source_features(x_src, defer=True)      # multi-GPU: halo exchange of the PROJECTED rows, overlapped
        agg = PF.aggregate(graph, x_src, Fi, t0.aggregators, n_tower=T, dst_term=x_dst, edge_term=x_edge, edge_type=edge_type)   # (waits for it)
    else:
        # general pretrans (MLP with hidden layers): per-edge messages are materialised in CSR order
        src, dst = csr.col.long(), csr.row.long()
        msgs = []
        for t, tower in enumerate(towers):
            z = [graph.source_features(hs[t])[src], hs[t][dst]] + ([e_csr] if tower.edge_features else [])
            msgs.append(tower.pretrans(torch.cat(z, dim=1)))
        agg = PF.aggregate(graph, torch.cat(msgs, dim=1) if T > 1 else msgs[0], Fi, t0.aggregators, n_tower=T,
                           edge_resident=True)

    # posttrans: y_t = W_t [h_t | id*agg_t | amp*agg_t | att*agg_t] + b_t, scalers applied per row in the epilogue
    scales = _row_scales(graph, t0.scalers, t0.avg_d, h.device)
    K = A * Fi
    # inference: graph-norm, eval BatchNorm (and dropout = identity) fold into the contraction's epilogue and each
    # tower writes its slice of the concatenated output directly
    fuse = (not torch.is_grad_enabled() or not any(p.requires_grad for t in towers for p in t.parameters())) and \
        all((not t.training) and t.posttrans.is_affine for t in towers) and not h.requires_grad
    if fuse and (not divide_input or Fi >= 4) and all(t.graph_norm == t0.graph_norm and t.batch_norm == t0.batch_norm for t in towers):
        # all towers' contractions in one call (one launch for the batches the exact-f32 kernel serves)
        No = t0.out_dim
        h_cat = torch.empty(V, T * No, dtype=torch.float32, device=h.device)
        PF.posttrans_towers(agg, K, [t.posttrans.fully_connected[0].linear for t in towers], scales, h, not divide_input, h_cat,
                            row_post=snorm_n if t0.graph_norm else None,
                            bns=[t.batchnorm_h for t in towers] if t0.batch_norm else None)
        return h_cat
    if fuse:
        No = t0.out_dim
        h_cat = torch.empty(V, T * No, dtype=torch.float32, device=h.device)
        for t, tower in enumerate(towers):
            lin = tower.posttrans.fully_connected[0].linear
            PF.posttrans(agg[:, t * K:(t + 1) * K], K, lin.weight, lin.bias, scales, h_self=hs[t],
                         row_post=snorm_n if tower.graph_norm else None,
                         bn=tower.batchnorm_h if tower.batch_norm else None, out=h_cat[:, t * No:(t + 1) * No])
        return h_cat
    outs = []
    for t, tower in enumerate(towers):
        lin = tower.posttrans.fully_connected[0].linear
        y = PF.posttrans(agg[:, t * K:(t + 1) * K], K, lin.weight, lin.bias, scales, h_self=hs[t])
        outs.append(tower.tail(y, snorm_n))
    return torch.cat(outs, dim=1) if T > 1 else outs[0]


class PNALayer(nn.Module):
    def __init__(self, in_dim, out_dim, aggregators, scalers, avg_d, dropout, graph_norm, batch_norm, towers=1,
                 pretrans_layers=1, posttrans_layers=1, divide_input=True, residual=False, edge_features=False,
                 edge_dim=0):
        super().__init__()
        assert ((not divide_input) or in_dim % towers == 0), "if divide_input is set the number of towers has to divide in_dim"
        assert (out_dim % towers == 0), "the number of towers has to divide the out_dim"
        assert avg_d is not None

        aggregators = aggregators.split() if isinstance(aggregators, str) else list(aggregators)
        scalers = scalers.split() if isinstance(scalers, str) else list(scalers)
        for a in aggregators:
            AGGREGATORS[a]                      # KeyError on unknown names (pna_layer.py:107-108)
        for s in scalers:
            SCALERS[s]

        self.divide_input = divide_input
        self.input_tower = in_dim // towers if divide_input else in_dim
        self.output_tower = out_dim // towers
        self.in_dim, self.out_dim = in_dim, out_dim
        self.edge_features = edge_features
        self.residual = residual and in_dim == out_dim      # silently disabled otherwise (:117-118)

        self.towers = nn.ModuleList(
            PNATower(in_dim=self.input_tower, out_dim=self.output_tower, aggregators=aggregators, scalers=scalers,
                     avg_d=avg_d, pretrans_layers=pretrans_layers, posttrans_layers=posttrans_layers,
                     batch_norm=batch_norm, dropout=dropout, graph_norm=graph_norm, edge_features=edge_features,
                     edge_dim=edge_dim) for _ in range(towers))
        self.mixing_network = FCLayer(out_dim, out_dim, activation="LeakyReLU")

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_pna_amd_small", None)             # cached weight images follow the parameters' device / dtype
        return super()._apply(fn, *args, **kwargs)

    def _small_structure_ok(self):
        """The structural conditions of the one-call path (they do not change after construction): 1-layer pretrans and
        posttrans, the four standard aggregators, <= 3 scalers, a mixing network without batch-norm.  Edge features: served when the
        call's feature rows are an embedding of <= 4 types (forward checks that per call: Graph.edge_type_table)."""
        towers = list(self.towers)
        t0, mix = towers[0], self.mixing_network
        return (all(t.edge_features == t0.edge_features and t.edge_dim == t0.edge_dim for t in towers) and tuple(t0.aggregators) == ("mean", "max", "min", "std") and len(t0.scalers) <= 3
                and all(len(t.pretrans.fully_connected) == 1 and len(t.posttrans.fully_connected) == 1
                        and t.pretrans.fully_connected[0].activation is None and t.posttrans.fully_connected[0].activation is None
                        and t.pretrans.fully_connected[0].b_norm is None and t.posttrans.fully_connected[0].b_norm is None
                        and t.graph_norm == t0.graph_norm and t.batch_norm == t0.batch_norm for t in towers)
                and mix.b_norm is None and (mix.activation is None or isinstance(mix.activation, (nn.LeakyReLU, nn.ReLU)))
                and PF.small_tower_fits(len(towers), t0.in_dim, t0.out_dim, self.divide_input))

    def _small_batch_path(self, graph, h):
        """Whether this call is served by pna_tower_layer_f32 (one C call for the whole layer): inference on a whole
        (unsharded) graph of molecule-batch size, with the structure _small_structure_ok() describes."""
        if self.training or not h.is_cuda or h.dtype != torch.float32 or type(graph) is not Graph:
            return False
        ok = self.__dict__.get("_pna_amd_small_ok")
        if ok is None:
            ok = self.__dict__["_pna_amd_small_ok"] = self._small_structure_ok()
        if not ok or not 0 < h.shape[0] <= PF.SMALL_TOWER_ROWS * (8 if self.in_dim <= 128 and self.divide_input else 1):
            return False
        if any(t.training for t in self._modules["towers"]._modules.values()):
            return False
        return not (torch.is_grad_enabled() and (h.requires_grad or any(p.requires_grad for p in self.parameters())))

    def forward(self, g, h, e, snorm_n):
        graph = as_graph(g)
        if self._small_batch_path(graph, h):
            t0 = self.towers[0]
            etab = None
            if self.edge_features:
                # ZINC with --edge_feat True (realworld_benchmark/README.md:62): e = embedding_e(bond type), <= 4 distinct rows -> the
                # W_e . ef term is a 4-row table the one-call kernel indexes by edge type (not while a hipGraph is being captured: the
                # table is read from the VALUES of e with host syncs; the general route below is captured instead)
                if e is None:
                    raise ValueError("edge_features=True but no edge features were given")
                if e.is_cuda and not e.requires_grad and (not torch.cuda.is_current_stream_capturing() or graph.edge_types_registered(e)):
                    etab = graph.edge_type_table(e)          # (registered types -- PNANet -- are device-side results: capture-safe)
            if etab is not None or not self.edge_features:
                return PF.tower_layer_small(self, list(self.towers), self.mixing_network, graph, h, snorm_n,
                                            _row_scales(graph, t0.scalers, t0.avg_d, h.device), self.divide_input, self.residual, etab)
        if PF.tower_layer_degree_grouped_applies(self, graph, h):
            # large whole graphs, inference: node-level projections, gather in degree order, ONE grouped contraction for posttrans,
            # graph norm, BatchNorm and the mixing network (functional.tower_layer_degree_grouped)
            towers = list(self.towers)
            T, Fi = len(towers), towers[0].in_dim
            if PF.tower_layer_degree_fused_applies(self, graph, h):
                # one tower, or T towers over slices of the input (divide_input): everything after the projection in ONE kernel
                # (functional.tower_layer_degree_fused)
                if T == 1:
                    Wpad, bpad = _projection_cache_padded(towers[0], Fi, PF.tower_projection_pitch(Fi))
                elif self.divide_input:
                    Wpad, bpad = _projection_cache_padded_div(towers, Fi, PF.tower_projection_pitch(T * Fi))
                else:
                    # T projections of the whole input: one launch per tower over its own block of the projection table.  The table itself:
                    # pna_project_f32 (the weight resident in LDS, h read once: 0.70 ms at 1 M rows x 75 -> 400 columns against the library
                    # GEMM's 1.03 and the contraction kernels' 1.06 / 1.21), the library GEMM where the weight does not fit the LDS
                    Wt, Wn = _projection_cache_padded_multi(towers, Fi, PF.tower_projection_pitch(Fi))
                    x_src = ops.project(h, Fi, Wn) if ops.project_applies(h, Fi, Wn.shape[0]) else torch.mm(h, Wt)
                    return PF.tower_layer_degree_fused_multi(self, graph, h, snorm_n, x_src)
                return PF.tower_layer_degree_fused(self, graph, h, snorm_n, PF.linear_act(h, Wpad, bpad))
            if self.divide_input:
                W = torch.stack([t.pretrans.fully_connected[0].linear.weight for t in towers])
                b = torch.stack([t.pretrans.fully_connected[0].linear.bias for t in towers])
                hv = h.reshape(h.shape[0], T, Fi)
                x_src = torch.einsum("vti,tfi->vtf", hv, W[:, :, :Fi]).reshape(h.shape[0], T * Fi)
                x_dst = (torch.einsum("vti,tfi->vtf", hv, W[:, :, Fi:2 * Fi]) + b).reshape(h.shape[0], T * Fi)
            else:
                Wcat, bcat = _projection_cache(towers, Fi)
                x_cat = PF.linear_act(h, Wcat, bcat)
                x_src, x_dst = x_cat[:, :T * Fi], x_cat[:, T * Fi:]
            return PF.tower_layer_degree_grouped(self, graph, h, snorm_n, x_src, x_dst)
        h_cat = _towers_forward(list(self.towers), graph, h, e, snorm_n, self.divide_input)
        mix = self.mixing_network
        if (h_cat.shape[1] >= 4 and isinstance(mix.activation, nn.LeakyReLU) and mix.b_norm is None and (mix.dropout is None or not self.training)
                and not (torch.is_grad_enabled() and (h_cat.requires_grad or any(p.requires_grad for p in mix.parameters())))):
            # inference: Linear + LeakyReLU (+ residual) of the mixing network as one launch of the contraction kernel
            return PF.linear_act(h_cat, mix.linear.weight, mix.linear.bias, leaky_slope=mix.activation.negative_slope,
                                 residual=h if self.residual else None)
        h_out = mix(h_cat)
        if self.residual:
            h_out = h + h_out
        return h_out

    def __repr__(self):
        return "{}(in_channels={}, out_channels={})".format(self.__class__.__name__, self.in_dim, self.out_dim)


class PNASimpleLayer(nn.Module):
    """Tower-less layer of the MolHIV net (pna_layer.py:151-216): messages are the raw source features."""

    def __init__(self, in_dim, out_dim, aggregators, scalers, avg_d, dropout, batch_norm, residual,
                 posttrans_layers=1):
        super().__init__()
        aggregators = aggregators.split() if isinstance(aggregators, str) else list(aggregators)
        scalers = scalers.split() if isinstance(scalers, str) else list(scalers)
        self.aggregators = _names(aggregators, AGGREGATORS, "aggregator")
        self.scalers = _names(scalers, SCALERS, "scaler")
        self.in_dim, self.out_dim = in_dim, out_dim
        self.dropout = dropout
        self.batch_norm = batch_norm
        self.residual = residual
        self.avg_d = avg_d

        self.batchnorm_h = nn.BatchNorm1d(out_dim)
        self.posttrans = MLP(in_size=(len(self.aggregators) * len(self.scalers)) * in_dim, hidden_size=out_dim,
                             out_size=out_dim, layers=posttrans_layers, mid_activation="relu",
                             last_activation="none")

    def aggregate(self, g, h):
        """The (V, A*S*F) tensor reduce_func leaves in ndata['h'] (pna_layer.py:189-194), materialised.
        forward() does not use it (it keeps the scalers out of HBM); this is the drop-in for code that
        wants the reference's intermediate."""
        graph = as_graph(g)
        return PF.aggregate(graph, graph.source_features(h), self.in_dim, self.aggregators,
                            row_scales=_row_scales(graph, self.scalers, self.avg_d, h.device))

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_pna_amd_small", None)             # cached weight images follow the parameters' device / dtype
        return super()._apply(fn, *args, **kwargs)

    def _small_batch_path(self, graph, h):
        """Whether this call is served by pna_tower_layer_f32 (one C call, two launches: functional._SmallSimplePlan):
        inference on a whole graph of molecule-batch size with the four standard aggregators and a 1-layer posttrans."""
        if self.training or not h.is_cuda or h.dtype != torch.float32 or type(graph) is not Graph:
            return False
        if not (tuple(self.aggregators) == ("mean", "max", "min", "std") and len(self.scalers) <= 3 and self.posttrans.is_affine
                and (not self.residual or self.in_dim == self.out_dim) and 0 < h.shape[0] <= PF.SMALL_SIMPLE_ROWS
                and PF.small_tower_fits(1, self.in_dim, self.out_dim, False, self.out_dim)):
            return False
        return not (torch.is_grad_enabled() and (h.requires_grad or any(p.requires_grad for p in self.parameters())))

    def _degree_grouped_path(self, graph, h):
        """Inference on a large whole graph: rows ordered by in-degree, one combined scaler block per degree value
        (pna_amd/degree_groups.py: a third of the posttrans multiply-adds)."""
        from .. import degree_groups as DG
        if self.training or not h.is_cuda or h.dtype != torch.float32 or not self.posttrans.is_affine:
            return False
        if PF.ops.POSTTRANS_ARITH == "f32" or self.in_dim != h.shape[1]:
            return False
        n_src = h.shape[0] + getattr(graph, "n_halo", 0)                    # (a shard's table: local rows + halo rows)
        if not DG.applies(graph, h.shape[0], self.out_dim, len(self.scalers), self.aggregators, F=self.in_dim,
                          n_edges=graph.csr.col.numel(), x_rows=n_src):
            return False
        if self.residual and h.shape[1] != self.out_dim:
            return False
        return not (torch.is_grad_enabled() and (h.requires_grad or any(p.requires_grad for p in self.parameters())))

    def forward(self, g, h):
        graph = as_graph(g)
        if self._small_batch_path(graph, h):
            return PF.simple_layer_small(self, graph, h, _row_scales(graph, self.scalers, self.avg_d, h.device))
        h_in = h
        if self._degree_grouped_path(graph, h):
            try:
                return PF.simple_layer_degree_grouped(self, graph, h)
            except RuntimeError as e:                        # a precondition of the hand-scheduled gather this check does not
                if "hand-scheduled kernel was required" not in str(e):     # mirror: the ordinary path takes the call
                    raise
        lin = self.posttrans.fully_connected[0].linear
        y = None
        if torch.is_grad_enabled() and (h.requires_grad or any(p.requires_grad for p in self.parameters())):
            from .. import autograd as AG
            if AG.simple_layer_plan_applies(self, graph, h):
                # training on a large whole graph: gather + contraction in the degree plan's row order (autograd.SimpleLayerPlanFn)
                y = AG.SimpleLayerPlanFn.apply(h, lin.weight, lin.bias, self, graph)
        if y is not None:
            return self._train_tail(y, h_in)
        # (V, A*F), identity scaler only; on a sharded graph the halo exchange overlaps the rows that do not need it
        agg = PF.aggregate(graph, graph.source_features(h, defer=True), self.in_dim, self.aggregators)
        scales = _row_scales(graph, self.scalers, self.avg_d, h.device)
        K = len(self.aggregators) * self.in_dim
        if not self.training and self.posttrans.is_affine and not (torch.is_grad_enabled() and (
                h.requires_grad or any(p.requires_grad for p in self.parameters()))):
            # inference: BatchNorm (running stats), ReLU and the residual fold into the contraction's epilogue
            if self.residual and h_in.shape[1] != self.out_dim:
                raise RuntimeError(f"The size of tensor a ({h_in.shape[1]}) must match the size of tensor b "
                                   f"({self.out_dim}) at non-singleton dimension 1")   # same failure as the reference (:213)
            return PF.posttrans(agg, K, lin.weight, lin.bias, scales, bn=self.batchnorm_h if self.batch_norm else None,
                                relu=True, residual=h_in if self.residual else None)
        y = PF.posttrans(agg, K, lin.weight, lin.bias, scales, degree_graph=graph if type(graph) is Graph else None)
        return self._train_tail(y, h_in)

    def _train_tail(self, y, h_in):
        """models/dgl/pna_layer.py:207-215 behind the posttrans Linear: BatchNorm, ReLU, residual, dropout."""
        y = self.posttrans.tail(y)
        if self.batch_norm:
            from ..autograd import bn_relu_residual, bn_tail_applies
            res = h_in if self.residual else None
            if bn_tail_applies(self.batchnorm_h, y, res):
                # training: batch-statistics BatchNorm + ReLU + residual in two streaming passes, two more in the backward
                # (pna_bn_tail_*_f32) instead of the library's ~10 over the (V, out_dim) tensors
                return F.dropout(bn_relu_residual(self.batchnorm_h, y, res), self.dropout, training=self.training)
            y = self.batchnorm_h(y)
        y = F.relu(y)
        if self.residual:
            y = h_in + y            # like the reference, no in_dim == out_dim guard here (:212-213)
        return F.dropout(y, self.dropout, training=self.training)

    def __repr__(self):
        return "{}(in_channels={}, out_channels={})".format(self.__class__.__name__, self.in_dim, self.out_dim)
