"""Graph-level functional API used by the layers: `aggregate` (gather + segment-reduce kernel) and
`posttrans` (MFMA contraction), both differentiable through custom autograd Functions.
"""
import os

import torch

from . import ops


def _unit_stride(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError(f"pna_amd computes in fp32 like the reference; got {t.dtype}")
    return t if t.stride(-1) == 1 else t.contiguous()


def aggregate(graph, x, F, aggregators, *, n_tower=1, dst_term=None, edge_term=None, row_scales=(None,),
              edge_resident=False, edge_weight=None, col_override=None, edge_type=None):
    """(V, n_tower * S * A * F) aggregate of the messages flowing into every node of `graph`.

    message(u->v) = x[u] (+ dst_term[v]) (+ edge_term[k]);  with edge_resident=True, x already holds
    one message per edge in CSR order.  Tower t reads columns [t*F, (t+1)*F).  Column layout of the
    result: tower-major, then scaler-major, then aggregator-major (the reference's cat order).
    edge_weight: fp32 [E] in CSR order (dense-variant adjacency weights).  col_override: int32 [E] row of
    `x` to gather for each CSR edge instead of the edge's source node (x then holds per-edge messages in
    some other edge order).  edge_type: int32 [E] in CSR order -- edge_term then has one row per edge TYPE (inference only).
    """
    x, dst_term, edge_term = _unit_stride(x), _unit_stride(dst_term), _unit_stride(edge_term)
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, dst_term, edge_term)):
        graph.finish_exchange()
        from .autograd import AggregateFn
        return AggregateFn.apply(graph, x, dst_term, edge_term, F, tuple(aggregators), n_tower, tuple(row_scales),
                                 edge_resident, edge_weight, col_override)
    csr = graph.csr
    col = None if edge_resident else (csr.col if col_override is None else col_override)
    if getattr(graph, "_pending", None) is not None:
        # a sharded graph whose halo exchange is still in flight (HaloGraph.source_features(defer=True)): the rows that only
        # read local sources go first, on the hand-scheduled kernel (the only one that honours a partial work list)
        done = None
        if (col is csr.col and edge_term is None and edge_weight is None and tuple(aggregators) == ("mean", "max", "min", "std")
                and all(r is None for r in row_scales)):
            _, items_in, items_bd = graph.split_work_lists()
            A = len(aggregators)
            out = torch.empty(csr.rowptr.numel() - 1, n_tower * A * F, dtype=torch.float32, device=x.device)
            try:
                if items_in.shape[0]:
                    ops.segreduce(csr.rowptr, col, x, F, aggregators, row_scales, n_tower=n_tower, tower_stride_in=F,
                                  dst_term=dst_term, out=out, items=items_in, tune=dict(generic=2))
                graph.finish_exchange()
                if items_bd.shape[0]:
                    ops.segreduce(csr.rowptr, col, x, F, aggregators, row_scales, n_tower=n_tower, tower_stride_in=F,
                                  dst_term=dst_term, out=out, heavy=graph.heavy_schedule(), workspace=graph.workspace,
                                  items=items_bd, tune=dict(generic=2))
                done = out
            except RuntimeError as ex:                       # the call does not qualify for that kernel: one ordinary launch
                if "hand-scheduled kernel was required" not in str(ex):
                    raise
        graph.finish_exchange()
        if done is not None:
            return done
    return ops.segreduce(csr.rowptr, col, x, F, aggregators, row_scales,
                         n_tower=n_tower, tower_stride_in=F, dst_term=dst_term, edge_term=edge_term,
                         edge_weight=edge_weight, heavy=graph.heavy_schedule(), workspace=graph.workspace,
                         items=graph.work_items(), edge_type=edge_type)


def _fold_batchnorm(bn):
    """Eval-mode BatchNorm1d as y = x * col_scale + col_shift; cached ON the module per parameter/buffer version so
    that an inference loop does not relaunch the five little fold kernels every forward."""
    ts = [t for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var) if t is not None]
    key = tuple((id(t), t._version, t.data_ptr(), str(t.device)) for t in ts) + (bn.eps,)
    hit = bn.__dict__.get("_pna_amd_fold")
    if hit is None or hit[0] != key:
        with torch.no_grad():
            cs = (bn.weight if bn.weight is not None else 1.0) * torch.rsqrt(bn.running_var + bn.eps)
            ct = (bn.bias if bn.bias is not None else 0.0) - bn.running_mean * cs
        hit = (key, cs.contiguous(), ct.contiguous(), ts)      # `ts` keeps the keyed tensors alive (id() stays unique)
        bn.__dict__["_pna_amd_fold"] = hit
    return hit[1], hit[2]


def posttrans(agg, K, weight, bias, row_scales, h_self=None, *, row_post=None, bn=None, relu=False, residual=None,
              out=None, degree_graph=None):
    """y = W [h_self | s_0*agg | s_1*agg | ...] + b with `weight` in the reference's nn.Linear layout
    (N, Kh + S*K) -- models/dgl/pna_layer.py:65-68 / :206 -- computed without materialising the
    scaled copies of `agg` (row_scales[s] is a per-row vector or None for the identity scaler).

    Optional fused tail (inference): row_post = snorm_n (V,1)/(V,), bn = an nn.BatchNorm1d in eval mode (folded
    to a per-column affine map), relu, residual -- applied in the reference's order
    (pna_layer.py:71-75 / :209-213)."""
    agg, h_self = _unit_stride(agg), _unit_stride(h_self)
    Kh = 0 if h_self is None else h_self.shape[1]
    if weight.shape[1] != Kh + len(row_scales) * K:
        raise ValueError(f"posttrans weight has {weight.shape[1]} input columns, expected {Kh + len(row_scales) * K}")
    if K < 4 or 0 < Kh < 4:
        # the MFMA kernel reads its operands in 16-byte pieces: widen tiny operands (multitask layers: 2 features
        # per tower) to 4 columns with zeros, and give the weight matching zero columns
        S, N = len(row_scales), weight.shape[0]
        Kp, Khp = max(K, 4), (max(Kh, 4) if Kh else 0)
        wb = [weight[:, :Kh]] + [weight[:, Kh + s * K:Kh + (s + 1) * K] for s in range(S)]
        wb = [torch.nn.functional.pad(w, (0, (Khp if i == 0 else Kp) - w.shape[1])) for i, w in enumerate(wb)]
        weight = torch.cat(wb if Kh else wb[1:], dim=1)
        agg = torch.nn.functional.pad(agg[:, :K], (0, Kp - K))
        if Kh:
            h_self = torch.nn.functional.pad(h_self, (0, Khp - Kh))
        K, Kh = Kp, Khp
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (agg, weight, bias, h_self)):
        from .autograd import PosttransFn
        if row_post is not None or bn is not None or relu or residual is not None:
            raise RuntimeError("the fused posttrans tail is inference-only")
        # degree_graph: the (unsharded) Graph whose degree scalers `row_scales` are -- lets the weight gradient walk the rows in the
        # graph's degree-plan order (autograd.PosttransFn.backward)
        return PosttransFn.apply(agg, K, weight, bias, tuple(row_scales), h_self, degree_graph)
    col_scale = col_shift = None
    if bn is not None:
        if bn.training:
            raise RuntimeError("only an eval-mode BatchNorm (running statistics) can be folded into the epilogue")
        col_scale, col_shift = _fold_batchnorm(bn)
    if row_post is not None:
        row_post = row_post.reshape(-1).contiguous()
    w = weight if weight.stride(-1) == 1 else weight.contiguous()
    return ops.posttrans(agg, K, w, row_scales, bias, h_self, out=out, row_post=row_post, col_scale=col_scale,
                         col_shift=col_shift, relu=relu, residual=_unit_stride(residual))


def _stack_cached(owner, tag, tensors):
    """torch.stack(tensors) cached on `owner` per (version, address, device) of every tensor: the per-tower biases and folded
    BatchNorm constants of a layer, which would otherwise cost a concatenation launch per forward."""
    key = tuple((t._version, t.data_ptr(), str(t.device)) for t in tensors)
    hit = owner.__dict__.get(tag)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            hit = (key, torch.stack([t.detach() for t in tensors]).contiguous())
        owner.__dict__[tag] = hit
    return hit[1]


def posttrans_towers(agg, K, towers_lin, row_scales, hs, h_shared, out, row_post=None, bns=None, relu=False):
    """Inference-only: the first posttrans Linear of every tower (+ graph-norm / folded eval BatchNorm) in ONE call.
    towers_lin: the nn.Linear of each tower; hs: the (V, Kh) shared input or the (V, T*Kh) divided one; bns: per-tower
    BatchNorm1d modules (eval) or None."""
    weights = [lin.weight for lin in towers_lin]
    owner = towers_lin[0]
    biases = _stack_cached(owner, "_pna_amd_bias_stack", [lin.bias for lin in towers_lin]) if towers_lin[0].bias is not None else None
    cs = ct = None
    if bns is not None:
        folds = [_fold_batchnorm(bn) for bn in bns]
        cs = _stack_cached(owner, "_pna_amd_cs_stack", [f[0] for f in folds])
        ct = _stack_cached(owner, "_pna_amd_ct_stack", [f[1] for f in folds])
    if row_post is not None:
        row_post = row_post.reshape(-1).contiguous()
    return ops.posttrans_towers(_unit_stride(agg), K, weights, row_scales, biases, _unit_stride(hs), h_shared, out, row_post=row_post,
                                col_scale=cs, col_shift=ct, relu=relu)


def linear_act(x, weight, bias, leaky_slope=None, relu=False, residual=None, out=None):
    """act(x @ weight^T + bias) (+ residual) on the contraction kernels (inference): the tower layers' mixing network
    (FCLayer(out, out, 'LeakyReLU') + residual, models/dgl/pna_layer.py:128,:141-144) as one launch."""
    w = weight if weight.stride(-1) == 1 else weight.contiguous()
    return ops.posttrans(_unit_stride(x), x.shape[1], w, [None], bias, out=out, relu=relu, leaky_slope=leaky_slope,
                         residual=_unit_stride(residual))


# ---- molecule-sized batches: the whole tower layer as one C call (pna_tower_layer_f32) --------------------------------------
SMALL_TOWER_ROWS = int(os.environ.get("PNA_AMD_SMALL_TOWER_ROWS", "32768"))   # batches up to this many nodes take that path


def small_tower_fits(T, Fi, Fo, divided, No=None):
    """Mirror of pna_tower_layer_f32's LDS check for a single tower per pass (the smallest tile it can run with)."""
    quads = lambda k: (k + 15) // 16   # noqa: E731
    No = T * Fo if No is None else No
    floats = 16 * (quads(4 * Fi) * 16 + 4) + (T if divided else 1) * 16 * (quads(Fi) * 16 + 4) + 16 * (quads(T * Fo) * 16 + 4) + \
        16 * 256 + 3 * quads(T * Fo) * 16 + quads(No) * 16 + 64
    return floats * 4 <= 160 * 1024


class _SmallTowerPlan:
    """Everything pna_tower_layer_f32 reads besides the graph and h -- the packed projection / posttrans / mixing images, the
    concatenated biases and folded BatchNorm constants -- and a pre-filled argument block, built once per weight state.  The
    host side of this path is on the critical path (the two launches take ~50 us for a 128-molecule batch), so a call only
    compares the parameters' version counters and fills in the per-call pointers."""

    edge_dim = 0                                               # (plans without edge features: _SmallSimplePlan)
    _etab = None

    def __init__(self, towers, mix, divide_input):
        import ctypes
        from . import _lib
        t0 = towers[0]
        self.edge_dim = t0.edge_dim if t0.edge_features else 0
        self._etab = None                                      # (type-row tensor of the graph's cache, its projection W_e . ef_t)
        pre = [t.pretrans.fully_connected[0].linear for t in towers]
        post = [t.posttrans.fully_connected[0].linear for t in towers]
        ts = [p for l in pre + post for p in (l.weight, l.bias) if p is not None]
        if t0.batch_norm:
            for t in towers:
                bn = t.batchnorm_h
                if bn.training:
                    raise RuntimeError("only an eval-mode BatchNorm (running statistics) can be folded into the epilogue")
                ts += [x for x in (bn.weight, bn.bias, bn.running_mean, bn.running_var) if x is not None]
        if mix is not None:
            ts += [p for p in (mix.linear.weight, mix.linear.bias) if p is not None]
        self.ts = ts
        self.versions = self._state()
        T, Fi, Fo, S = len(towers), t0.in_dim, t0.out_dim, len(t0.scalers)
        self.T, self.Fi, self.Fo, self.S, self.divide_input = T, Fi, Fo, S, divide_input
        with torch.no_grad():
            Wa = [l.weight[:, :Fi] for l in pre]
            Wb = [l.weight[:, Fi:2 * Fi] for l in pre]
            if divide_input:
                Wcat = torch.cat([torch.block_diag(*Wa), torch.block_diag(*Wb)], dim=0)
            else:
                Wcat = torch.cat(Wa + Wb, dim=0)
            b = torch.cat([l.bias if l.bias is not None else torch.zeros(Fi, device=Wcat.device) for l in pre])
            # W_e of every tower stacked (T*Fi, edge_dim): the <= 4 edge-type rows are projected through it once per (graph batch, e)
            self.We = torch.cat([l.weight[:, 2 * Fi:2 * Fi + self.edge_dim] for l in pre], dim=0).contiguous() if self.edge_dim else None
            keep = dict(proj_img=ops.pack_small(Wcat.contiguous()), proj_bias=torch.cat([torch.zeros_like(b), b]).contiguous(),
                        post_img=ops.pack_tower_post([l.weight for l in post], Fi, Fo, S),
                        post_bias=torch.cat([l.bias for l in post]).contiguous() if post[0].bias is not None else None)
            if t0.batch_norm:
                folds = [_fold_batchnorm(t.batchnorm_h) for t in towers]
                keep["col_scale"] = torch.cat([f[0] for f in folds]).contiguous()
                keep["col_shift"] = torch.cat([f[1] for f in folds]).contiguous()
            if mix is not None:
                keep["mix_img"] = ops.pack_small(mix.linear.weight)
                keep["mix_bias"] = mix.linear.bias.detach().contiguous() if mix.linear.bias is not None else None
        self.keep = keep                                       # the device buffers the argument block points into
        self.device = Wcat.device
        self.width = mix.linear.weight.shape[0] if mix is not None else T * Fo
        self.xw = 2 * T * Fi
        self.graph_norm = bool(t0.graph_norm)
        a = _lib.PnaTowerLayerArgs()
        a.n_tower, a.Fi, a.Fo, a.divide_input, a.n_scaler = T, Fi, Fo, 1 if divide_input else 0, S
        for k, v in keep.items():
            if v is not None:
                setattr(a, k, _lib.dev_ptr(v, torch.float32, k))
        if mix is not None:
            a.No = self.width
            if isinstance(mix.activation, torch.nn.LeakyReLU):
                a.mix_act, a.mix_slope = 2, float(mix.activation.negative_slope)
            elif mix.activation is not None:
                a.mix_act = 1
        self.args, self.ref = a, ctypes.byref(a)
        self.fn = _lib.lib().pna_tower_layer_f32
        self.check, self.stream_ptr = _lib.check, _lib.stream_ptr

    def _state(self):
        # (version, storage address, device) per tensor: `param.data = other` (EMA / SWA swaps, vector_to_parameters, offloading)
        # keeps the version counter but moves the address -- the same key the other weight caches use
        return [(x._version, x.data_ptr(), str(x.device)) for x in self.ts]

    def stale(self):
        return self._state() != self.versions

    def edge_table(self, etab):
        """(n_types, T*Fi) = type rows @ W_e^T, kept while the graph's type table (graph.edge_type_table: cached per feature tensor)
        is the same object -- the weights' own staleness replaces the whole plan."""
        hit = self._etab
        if hit is None or hit[0] is not etab[1]:
            with torch.no_grad():
                hit = self._etab = (etab[1], (etab[1].to(torch.float32) @ self.We.t()).contiguous())
        return hit[1]

    def run(self, graph, h, snorm_n, row_scales, residual, etab=None):
        if h.stride(-1) != 1:
            h = h.contiguous()
        dev = h.device
        if dev != self.device:
            raise RuntimeError(f"layer parameters live on {self.device}, features on {dev}")
        V = h.shape[0]
        csr = graph.csr
        out = torch.empty((V, self.width), dtype=torch.float32, device=dev)
        xc = torch.empty((V, self.xw), dtype=torch.float32, device=dev)
        import ctypes
        a = type(self.args)()                                # per-call copy of the pre-filled block: concurrent calls of one layer from
        ctypes.memmove(ctypes.byref(a), self.ref, ctypes.sizeof(a))      # two host threads / streams do not race on it
        a.rowptr, a.col, a.V = csr.rowptr.data_ptr(), csr.col.data_ptr(), V
        a.h, a.ldh = h.data_ptr(), h.stride(0)
        a.x_cat, a.ldx = xc.data_ptr(), self.xw
        for i in range(self.S):
            rs = row_scales[i]
            a.row_scale[i] = None if rs is None else rs.data_ptr()
        if self.graph_norm and snorm_n is not None:
            if snorm_n.dtype != torch.float32 or snorm_n.numel() != V or snorm_n.device != dev or not snorm_n.is_contiguous():
                snorm_n = snorm_n.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
                if snorm_n.numel() != V:
                    raise ValueError("snorm_n must hold one factor per node")
            a.row_post = snorm_n.data_ptr()
        else:
            a.row_post = None
        if residual:
            a.residual, a.ld_res = a.h, a.ldh
        else:
            a.residual = None
        a.y, a.ldy = out.data_ptr(), self.width
        if self.edge_dim:
            if etab is None:
                raise RuntimeError("tower_layer_small: an edge-feature layer needs the graph's edge-type table")
            tab = self.edge_table(etab)
            a.edge_type, a.edge_table, a.ld_edge_table, a.n_edge_types = etab[0].data_ptr(), tab.data_ptr(), tab.stride(0), tab.shape[0]
        rc = self.fn(ctypes.byref(a), self.stream_ptr(dev))
        if rc != 0:
            self.check(rc, "pna_tower_layer_f32")
        del snorm_n, xc                                      # (stream-ordered allocator: safe to release after the launch)
        return out


def degree_grouped_aggregate(layer, graph, h, plan, out=None, x=None):
    """The (plan.rows, 4F) aggregate in the plan's row order (degree groups padded to whole tiles, then the rest): the gather's
    work list carries the output row of every whole-row record, heavy_out the output rows of the hub rows.  On a sharded
    graph the halo exchange is started first and the rows that read only local sources are aggregated while it is in flight
    (two launches, like functional.aggregate); `x`: the source table with the halo already in place (no exchange here)."""
    F = layer.in_dim
    K = len(layer.aggregators) * F
    if x is None:
        x = graph.source_features(h, defer=True)
    x = _unit_stride(x)
    csr = graph.csr
    if out is None:
        from . import degree_groups as DG
        out = torch.empty(plan.rows, DG.agg_pitch(K), dtype=torch.float32, device=h.device)[:, :K]     # line-aligned rows
    if getattr(graph, "_pending", None) is not None:
        items_in, items_bd = plan.split_items(graph)
        try:
            if items_in.shape[0]:
                ops.segreduce(csr.rowptr, csr.col, x, F, layer.aggregators, (None,), tower_stride_in=F, out=out, items=items_in,
                              tune=dict(generic=2))
        finally:
            graph.finish_exchange()
        if items_bd.shape[0]:
            ops.segreduce(csr.rowptr, csr.col, x, F, layer.aggregators, (None,), tower_stride_in=F, out=out,
                          heavy=graph.heavy_schedule(), workspace=graph.workspace, items=items_bd, heavy_out=plan.heavy_out,
                          tune=dict(generic=2))
        return out
    ops.segreduce(csr.rowptr, csr.col, x, F, layer.aggregators, (None,), tower_stride_in=F, out=out,
                  heavy=graph.heavy_schedule(), workspace=graph.workspace, items=plan.items, heavy_out=plan.heavy_out, tune=dict(generic=2))
    return out


def _simple_layer_state(layer):
    """(version, address) of every tensor a cached FusedDegreeCall of `layer` snapshots: the posttrans Linear and the BatchNorm."""
    lin = layer.posttrans.fully_connected[0].linear
    ts = [lin.weight, lin.bias]
    if layer.batch_norm:
        bn = layer.batchnorm_h
        ts += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
    return tuple((t._version, t.data_ptr()) for t in ts if t is not None)


def _layer_tail_operands(layer, h):
    """(column scale, column shift, residual) of the simple layer's epilogue: eval BatchNorm folded, the input as residual."""
    cs = ct = None
    if layer.batch_norm:
        cs, ct = _fold_batchnorm(layer.batchnorm_h)
    return cs, ct, (_unit_stride(h) if layer.residual else None)


def _rest_posttrans(layer, graph, agg_rest, plan, scales, y, cs, ct, res):
    """The rows no degree group holds (rare degrees, hub rows): the ordinary three-block contraction over their compact list
    `agg_rest` ((plan.NRp, 4F): the four STANDARD statistics, virtual order; (plan.NRp, 5F) with room for a fifth block when the layer
    aggregates `sum`), rows scattered to node order.  A layer with another aggregator list (round 6) contracts against its weight
    re-expressed over those statistics (degree_groups.virtual_layer_weight); the sum block = in-degree x mean is formed here."""
    from . import degree_groups as DG
    F, N = layer.in_dim, layer.out_dim
    K = 4 * F
    lin = layer.posttrans.fully_connected[0].linear
    from .dgl.pna_layer import _avg_log_value
    rest_scales = plan.rest_scales(tuple(layer.scalers) + (_avg_log_value(layer.avg_d),), scales)
    weight = lin.weight
    if tuple(layer.aggregators) != DG.STANDARD_AGGREGATORS:
        weight, K = DG.virtual_layer_weight(lin.weight, F, layer.aggregators, len(scales))
        if K == 5 * F:
            deg = plan.__dict__.get("_rest_deg")
            if deg is None:
                deg = torch.zeros(max(plan.NRp, 1), 1, dtype=torch.float32, device=y.device)
                deg[:plan.NR, 0] = plan._deg[plan.rest_rows].to(torch.float32)
                deg = plan.__dict__["_rest_deg"] = deg[:plan.NRp]
            torch.mul(agg_rest[:, :F], deg, out=agg_rest[:, 4 * F:5 * F])
    if N <= 80 and len(rest_scales) == 3:
        ops.posttrans(agg_rest, K, weight, rest_scales, lin.bias, out=y, col_scale=cs, col_shift=ct, relu=True, residual=res,
                      row_perm=plan.perm_rest, n_out=N)
    else:
        # 128-column block (three blocks x three weight buffers do not fit the LDS) or another scaler count (no grouped
        # instantiation): the few rest rows take the ordinary kernel over their compact list and are scattered by index (three
        # small torch kernels)
        rr = plan.rest_rows
        y_r = ops.posttrans(agg_rest[:plan.NR], K, weight, [None if r is None else r[:plan.NR] for r in rest_scales],
                            lin.bias, col_scale=cs, col_shift=ct, relu=True, residual=None if res is None else res.index_select(0, rr),
                            arith="bf16x3")
        y.index_copy_(0, rr, y_r)


def degree_grouped_posttrans(layer, graph, h, agg, plan, out=None):
    """The two contractions over `agg` (degree_grouped_aggregate): one combined block per degree tile, three blocks for the rest;
    both scatter their rows to node order with BatchNorm / ReLU / residual in the epilogue."""
    from . import degree_groups as DG
    from .dgl.pna_layer import _row_scales
    F, N = layer.in_dim, layer.out_dim
    K = len(layer.aggregators) * F
    lin = layer.posttrans.fully_connected[0].linear
    scales = _row_scales(graph, layer.scalers, layer.avg_d, h.device)
    y = torch.empty(h.shape[0], N, dtype=torch.float32, device=h.device) if out is None else out
    cs, ct, res = _layer_tail_operands(layer, h)
    if plan.G:
        img, stride = DG.combined_images(lin.weight, K, scales, plan)
        ops.posttrans(agg[:plan.NV], K, lin.weight, [None], lin.bias, out=y, col_scale=cs, col_shift=ct, relu=True, residual=res,
                      row_perm=plan.perm, tile_image=plan.tile_image, w_img=img, image_stride=stride, n_out=N)
    if plan.NR:                 # (on a side stream beside the grouped launch: measured, no gain -- one after the other)
        _rest_posttrans(layer, graph, agg[plan.NV:], plan, scales, y, cs, ct, res)
    return y


def _tower_collapsed_weights(owner, towers, mix, divide_input):
    """The tower layers' posttrans Linear, graph norm, eval BatchNorm and the mixing Linear have NO non-linearity between them
    (models/dgl/pna_layer.py:65-75, :141): for all towers t with z_t = [h_t | S a_t],
        W_m concat_t BN_t(snorm (W_t z_t + b_t)) + b_m  =  snorm (sum_t C_t z_t + d) + c,
        C_t = W_m[:, t] P_t W_t,   d = sum_t W_m[:, t] P_t b_t,   c = sum_t W_m[:, t] q_t + b_m        (BN_t(u) = P_t u + q_t)
    -- one contraction of [a_1 .. a_T | h] against the collapsed weight instead of T posttrans contractions, a (V, T Fo)
    intermediate and a mixing GEMM.  Returned as a virtual nn.Linear weight (out, S * K) in scaler blocks of
    K = T * 4 Fi + in_dim columns [C_1,s | .. | C_T,s | h panel (block 0 only)], so that the degree-grouped machinery of the simple
    layer (combined images W_D, three-block rest) applies unchanged; formed in float64, rounded once.  Cached on `owner`."""
    t0 = towers[0]
    T, Fi, Fo, S = len(towers), t0.in_dim, t0.out_dim, len(t0.scalers)
    lins = [t.posttrans.fully_connected[0].linear for t in towers]
    ts = [p for l in lins for p in (l.weight, l.bias)] + [mix.linear.weight, mix.linear.bias]
    if t0.batch_norm:
        ts += [x for t in towers for x in (t.batchnorm_h.weight, t.batchnorm_h.bias, t.batchnorm_h.running_mean, t.batchnorm_h.running_var)]
    key = tuple((x._version, x.data_ptr(), str(x.device)) for x in ts if x is not None) + (divide_input,)
    hit = owner.__dict__.get("_pna_amd_collapsed")
    if hit is not None and hit[0] == key:
        return hit[1:]
    with torch.no_grad():
        Wm = mix.linear.weight.double()
        out = Wm.shape[0]
        in_dim = T * Fi if divide_input else Fi
        K = T * 4 * Fi + in_dim
        Wv = torch.zeros(out, S * K, dtype=torch.float64, device=Wm.device)
        d = torch.zeros(out, dtype=torch.float64, device=Wm.device)
        c = mix.linear.bias.double().clone() if mix.linear.bias is not None else torch.zeros(out, dtype=torch.float64, device=Wm.device)
        for t, (tower, lin) in enumerate(zip(towers, lins)):
            if tower.batch_norm:
                bn = tower.batchnorm_h
                p = (bn.weight.double() if bn.weight is not None else 1.0) / torch.sqrt(bn.running_var.double() + bn.eps)
                q = (bn.bias.double() if bn.bias is not None else 0.0) - bn.running_mean.double() * p
            else:
                p, q = torch.ones(Fo, dtype=torch.float64, device=Wm.device), torch.zeros(Fo, dtype=torch.float64, device=Wm.device)
            WmP = Wm[:, t * Fo:(t + 1) * Fo] * p[None, :]
            C = WmP @ lin.weight.double()                                        # (out, Fi + S * 4 Fi): [h | scaler blocks]
            if lin.bias is not None:
                d += WmP @ lin.bias.double()
            c += Wm[:, t * Fo:(t + 1) * Fo] @ q
            for s_ in range(S):
                Wv[:, s_ * K + t * 4 * Fi:s_ * K + (t + 1) * 4 * Fi] = C[:, Fi + s_ * 4 * Fi:Fi + (s_ + 1) * 4 * Fi]
            hcol = T * 4 * Fi + (t * Fi if divide_input else 0)
            Wv[:, hcol:hcol + Fi] += C[:, :Fi]                                   # block 0 (the identity scaler's) carries the h panel
        res = (Wv.float().contiguous(), d.float().contiguous(), c.float().contiguous(), torch.ones(out, dtype=torch.float32, device=Wm.device), K)
    owner.__dict__["_pna_amd_collapsed"] = (key,) + res
    return res


def tower_layer_degree_grouped_applies(layer, graph, h):
    """Whether PNALayer.forward (eval) takes the degree-grouped path below: a large whole graph, 1-layer pretrans / posttrans, no
    edge features, the four standard aggregators, 3 scalers led by `identity`, a LeakyReLU mixing network without batch norm."""
    from . import degree_groups as DG
    from .graph import Graph
    towers = list(layer.towers)
    t0, mix = towers[0], layer.mixing_network
    if layer.training or not h.is_cuda or h.dtype != torch.float32 or type(graph) is not Graph or layer.edge_features:
        return False
    if not (DG.ENABLED and DG.TOWERS and h.shape[0] >= DG.MIN_ROWS and DG.MIN_OUT <= layer.out_dim <= 128 and t0.in_dim >= 4 and ops.POSTTRANS_ARITH != "f32"):
        return False
    if not (tuple(t0.aggregators) == ("mean", "max", "min", "std") and len(t0.scalers) == 3 and t0.scalers[0] == "identity"
            and isinstance(mix.activation, torch.nn.LeakyReLU) and mix.b_norm is None and mix.linear.bias is not None
            and all(t.pretrans.is_affine and t.posttrans.is_affine and t.scalers == t0.scalers and t.aggregators == t0.aggregators
                    and t.graph_norm == t0.graph_norm and t.batch_norm == t0.batch_norm and not t.training for t in towers)):
        return False
    if torch.is_grad_enabled() and (h.requires_grad or any(p.requires_grad for p in layer.parameters())):
        return False
    n_edges = graph.csr.col.numel()
    if not (0 < n_edges < (1 << 30)) or h.shape[0] >= (1 << 24):
        return False
    plan = DG.plan_of(graph)
    K = len(towers) * 4 * t0.in_dim + layer.in_dim
    return plan.G > 0 and plan.NR <= DG.MAX_REST_FRACTION * h.shape[0] and plan.rows * DG.agg_pitch(K) * 4 < (1 << 34)


def tower_layer_degree_grouped(layer, graph, h, snorm_n, x_src, x_dst):
    """PNALayer.forward (eval; models/dgl/pna_layer.py:130-145 over :55-76) after the node-level projections: the gather writes
    every tower's aggregate in DEGREE order (pna_segreduce_args.out_row_of), the rows' own features are copied behind it, and ONE
    degree-grouped contraction per row class (combined block W_D over the degree tiles, three scaler blocks over the rest) applies
    the collapsed posttrans . graph norm . BatchNorm . mixing weight (see _tower_collapsed_weights) with LeakyReLU and the
    residual in its epilogue."""
    from . import degree_groups as DG
    from .dgl.pna_layer import _row_scales
    towers, mix = list(layer.towers), layer.mixing_network
    t0 = towers[0]
    T, Fi = len(towers), t0.in_dim
    V, dev = h.shape[0], h.device
    plan = DG.plan_of(graph)
    Wv, d, c, ones, K = _tower_collapsed_weights(layer, towers, mix, layer.divide_input)
    N = Wv.shape[0]
    Ka = T * 4 * Fi
    buf = torch.empty(plan.rows, DG.agg_pitch(K), dtype=torch.float32, device=dev)
    csr = graph.csr
    ops.segreduce(csr.rowptr, csr.col, _unit_stride(x_src), Fi, t0.aggregators, (None,), n_tower=T, tower_stride_in=Fi, dst_term=_unit_stride(x_dst),
                  out=buf, tower_stride_out=4 * Fi, heavy=graph.heavy_schedule(), workspace=graph.workspace, items=graph.work_items(),
                  out_row_of=plan.vmap32(), heavy_out=plan.heavy_out, tune=dict(generic=2))
    hc = _unit_stride(h)
    ops.pack_rows(hc, plan.perm_all(), out=buf[:, Ka:Ka + hc.shape[1]])
    scales = _row_scales(graph, t0.scalers, t0.avg_d, dev)
    y = torch.empty(V, N, dtype=torch.float32, device=dev)
    res = hc if layer.residual else None
    slope = float(mix.activation.negative_slope)
    post_g = post_r = None
    if t0.graph_norm and snorm_n is not None:
        sn = snorm_n.reshape(-1).to(torch.float32)
        idx = plan.perm_all().long()
        post = sn[idx]
        post_g, post_r = post[:plan.NV].contiguous(), post[plan.NV:].contiguous()
    if plan.G:
        img, stride = DG.combined_images(Wv, K, scales, plan)
        ops.posttrans(buf[:plan.NV, :K], K, Wv, [None], d, out=y, row_post=post_g, col_scale=ones, col_shift=c, leaky_slope=slope, residual=res,
                      row_perm=plan.perm, tile_image=plan.tile_image, w_img=img, image_stride=stride, n_out=N)
    if plan.NR:
        from .dgl.pna_layer import _avg_log_value
        rest_scales = plan.rest_scales(tuple(t0.scalers) + (_avg_log_value(t0.avg_d),), scales)
        if N <= 80:
            ops.posttrans(buf[plan.NV:, :K], K, Wv, rest_scales, d, out=y, row_post=post_r, col_scale=ones, col_shift=c, leaky_slope=slope,
                          residual=res, row_perm=plan.perm_rest, n_out=N)
        else:
            rr = plan.rest_rows
            y_r = ops.posttrans(buf[plan.NV:plan.NV + plan.NR, :K], K, Wv, [None if r is None else r[:plan.NR] for r in rest_scales], d,
                                row_post=None if post_r is None else post_r[:plan.NR], col_scale=ones, col_shift=c, leaky_slope=slope,
                                residual=None if res is None else res.index_select(0, rr), arith="bf16x3")
            y.index_copy_(0, rr, y_r)
    return y


def tower_projection_pitch(Fi):
    """Column pitch of one half of the [x_src | x_dst] table the one-kernel tower layer gathers from: rows 16-byte aligned, the
    last strip inside the half."""
    return (Fi + 7) // 8 * 8


def tower_layer_degree_fused_applies(layer, graph, h):
    """Whether the grouped tower path runs its group rows through pna_fused_degree_f32's tower mode: gathers of 49..80 message
    features, at most 80 outputs, features the kernel can read in 16-byte pieces.  That is a layer with one tower; T towers with
    divide_input=True (round 5; models/dgl/pna_layer.py:133-136: tower t sees the input slice [t Fi, (t+1) Fi), so all towers' messages
    together are in_dim = T Fi wide: ONE gather over the block-diagonal projection, and the collapsed posttrans . BatchNorm . mixing
    weight is dense anyway); and -- round 6 -- T towers with divide_input=False (:137-139: T DIFFERENT projections of the whole input,
    T x in_dim message features per edge): one launch per tower over its own 49..80 projected features, the partial sums carried from
    launch to launch (FusedMultiTowerCall)."""
    from . import degree_groups as DG
    towers = list(layer.towers)
    T, Fi = len(towers), towers[0].in_dim
    if not DG.FUSED or not 4 <= layer.out_dim <= 80:
        return False
    if T == 1 or layer.divide_input:
        Fe = T * Fi                                        # message features per edge, all towers
        if not (49 <= Fe <= 80 and layer.in_dim == Fe and DG.fused_applies(graph, h, Fe, layer.out_dim)):
            return False
        return h.shape[0] * 2 * tower_projection_pitch(Fe) * 4 < (1 << 32)
    if not (49 <= Fi <= 80 and layer.in_dim == Fi and T <= FUSED_MAX_TOWER_PASSES and DG.fused_applies(graph, h, Fi, layer.out_dim)):
        return False
    return h.shape[0] * 2 * T * tower_projection_pitch(Fi) * 4 < (1 << 32)


FUSED_MAX_TOWER_PASSES = 8     # towers of a divide_input=False layer the one-kernel path takes (one launch -- one gather -- per tower)


def _tower_flat_weights(layer, towers, mix):
    """_tower_collapsed_weights for the one-kernel tower layer: the same collapsed weight with the columns of every scaler block
    re-ordered from tower-major [t: mean | max | min | std] (Fi each) to aggregator-major over ALL towers' features
    [mean (T Fi) | max | min | std | h panel] -- the layout of ONE tower of Fe = T Fi features, which is what the gather over the
    block-diagonal projection produces (divide_input=True; one tower: the identity)."""
    Wv, d, c, ones, K = _tower_collapsed_weights(layer, towers, mix, layer.divide_input)
    T, Fi = len(towers), towers[0].in_dim
    if T == 1:
        return Wv, d, c, ones, K
    hit = layer.__dict__.get("_pna_amd_flat")
    if hit is not None and hit[0] is Wv:
        return hit[1], d, c, ones, K
    S = Wv.shape[1] // K
    Fe = T * Fi
    a_ = torch.arange(4, device=Wv.device).view(4, 1, 1)
    t_ = torch.arange(T, device=Wv.device).view(1, T, 1)
    f_ = torch.arange(Fi, device=Wv.device).view(1, 1, Fi)
    cols = torch.cat([(t_ * 4 * Fi + a_ * Fi + f_).reshape(-1), torch.arange(4 * Fe, K, device=Wv.device)])      # new column -> old column
    idx = torch.cat([s_ * K + cols for s_ in range(S)])
    Wp = Wv.index_select(1, idx).contiguous()
    layer.__dict__["_pna_amd_flat"] = (Wv, Wp)
    return Wp, d, c, ones, K


_SIDE_STREAMS = {}
# How a one-kernel layer's two halves are recorded while a hipGraph is being CAPTURED (round 6): "beside" = as in eager
# mode (fork / join events become two branches of the graph), "behind" = one after the other on the capturing stream, whole device each,
# "rest_first" = the same with the small launches in front.  tools/graph_replay_ab.py measures the three against the eager step.
CAPTURE_OVERLAP = os.environ.get("PNA_AMD_CAPTURE_OVERLAP", "beside")


def _side_stream(device):
    """The second stream beside the CALLER's current stream: where the rest-row launches of the one-kernel layers run.  One side
    stream per (device, caller stream), so callers on different streams do not serialise on one another's chains.  What stays
    shared per GRAPH is its reallocating `graph.workspace` and the plan's lazily filled caches: layers over ONE Graph object must be
    run from one stream / thread at a time; different Graph objects are independent."""
    idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    hit = _SIDE_STREAMS.get(key)
    if hit is None:
        if len(_SIDE_STREAMS) > 64:
            _SIDE_STREAMS.clear()
        hit = _SIDE_STREAMS[key] = torch.cuda.Stream(device=idx)
    return hit


def run_fused_call(call):
    """The two halves of a one-kernel layer (FusedDegreeCall / FusedTowerCall): the persistent kernel over the group rows and the
    chain of small launches over the rest rows (hub rows, rare degrees).  On a large graph the chain runs BESIDE the kernel: the
    kernel books every register of every CU it sits on, so it leaves `degree_groups.FUSED_SPARE_WGS` of its workgroups out and
    the chain goes to a second stream, forked and joined with events around the call (benchmark graph: 0.852 -> 0.807 ms; the two
    halves write disjoint rows of y).  The caller's stream sees one call: everything it issued before is visible to both halves,
    everything it issues afterwards waits for both."""
    from . import _lib
    plan = call.plan
    call.stream = _lib.stream_ptr(call.y.device)             # (the kernel goes to the stream that is current NOW, like every other launch)
    if hasattr(call, "prologue"):
        call.prologue()                                      # (what BOTH halves read: in front of the fork)
    overlap = plan.rest_overlap_applies(call.layer_F())
    if overlap and CAPTURE_OVERLAP != "beside" and torch.cuda.is_current_stream_capturing():
        overlap = False                                      # (see CAPTURE_OVERLAP)
        if CAPTURE_OVERLAP == "rest_first":
            call.set_spare(False)
            call.rest_rows()
            call.group_rows()
            return call.y
    if not overlap:
        call.set_spare(False)
        call.group_rows()
        return call.rest_rows()
    dev = call.y.device
    main = torch.cuda.current_stream(dev)
    side, fork, join = _side_stream(dev), torch.cuda.Event(), torch.cuda.Event()   # (events per call; one Graph object: one caller at a time)
    call.set_spare(True)
    fork.record(main)
    call.group_rows()
    side.wait_event(fork)
    with torch.cuda.stream(side):
        call.rest_rows()
        join.record(side)
    main.wait_event(join)
    return call.y


def out_pitch(N):
    """Row pitch (floats) of the output the one-kernel layers allocate: whole 128-byte lines (N = 75 -> 96 floats).  The kernel
    stores a row as 64-byte pieces (16 columns: four lanes x 16 bytes); at a line-aligned pitch every piece is one aligned half
    line of a line no other row shares, at the smallest 16-byte aligned pitch (76 floats) most pieces straddle two sectors and
    neighbouring rows share lines.  Measured at C3 (one box, kernel ms): pitch 76 0.745, 80 0.735, 96 0.732.  A multi-layer net
    reads the rows back as the next layer's source table: three whole lines per 300-byte row instead of 3.4."""
    from . import degree_groups as DG
    a = max(4, int(DG.OUT_PITCH_ALIGN))
    return (N + a - 1) // a * a


def own_buffer_cols(pitch, c0, c1, N):
    """pna_fused_degree_args.y_cols_writable for the column panel [c0, c1) of a row of N columns in a buffer THIS module allocated at `pitch`
    floats (out_pitch): the last panel may write zeros into the row's padding up to the next multiple of 16 columns -- whole 32-byte
    sectors and one 16-byte store for the row's last window (WRITE_PADDING = False: off, for A/B runs)."""
    if not WRITE_PADDING or c1 != N:
        return 0
    return max(c1 - c0, min(pitch - c0, (c1 - c0 + 15) // 16 * 16))


WRITE_PADDING = os.environ.get("PNA_AMD_WRITE_PADDING", "1") != "0"


def _fused_grid(device, spare, n_tiles64):
    """Workgroups pna_fused_degree_f32 launches for the 4-wavefront shapes (pna_fused_degree.hip: two per CU, less the spare ones, never
    below one per CU, never more than tiles)."""
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    wgs = 2 * cus
    if spare > 0:
        wgs = wgs - spare if wgs - spare > cus else cus
    return max(1, min(wgs, n_tiles64))


DEBUG_COUNTERS = os.environ.get("PNA_AMD_DEBUG_COUNTERS", "0") == "1"   # host-side check (one sync per launch) that the tile counter pair and the
# guard's working words are zero in front of a launch of the one-kernel layer: a launch that left them non-zero would make every later launch
# on the plan skip tiles silently


def _check_working_words(call):
    keep = getattr(call, "_order_keep", None)
    if keep is None:
        return
    torch.cuda.current_stream(call.y.device).synchronize()
    for name, t in (("tile_counter", keep[3]), ("guard workspace", keep[4])):
        if t is not None and t[:2].tolist() != [0, 0]:
            raise RuntimeError(f"pna_amd: the {name} of this plan / stream is {t[:2].tolist()} in front of a launch (expected [0, 0]): an earlier launch "
                               "was aborted or ran on another stream than it was bound on")


def _bind_tile_order(call, spare):
    """(see _bind_tile_order_one; a layer in several output-column panels binds every panel's argument block to the same tables)"""
    if DEBUG_COUNTERS and getattr(call, "_order_keep", None) is not None:
        _check_working_words(call)
    blocks = getattr(call, "panel_args", None)
    if not blocks or len(blocks) == 1:
        return _bind_tile_order_one(call, spare)
    first, first_ref = call.args, call.ref
    _bind_tile_order_one(call, spare)
    src = call.args
    for a, _ in blocks[1:]:
        a.spare_workgroups, a.tile_desc, a.row_perm, a.tile_counter = src.spare_workgroups, src.tile_desc, src.row_perm, src.tile_counter
        a.guard_ws, a.guard_ws_bytes = src.guard_ws, src.guard_ws_bytes
    call.args, call.ref = first, first_ref


def _bind_tile_order_one(call, spare):
    """Point the call's argument block at the plan's tile list for the grid `spare` gives: the load-balanced order
    (DegreePlan.fused_balance) for the production instantiations of the 64-row-tile shapes, the plan's own order otherwise (the
    verification instantiation writes agg_out in plan order; the 8-wavefront build of the wide shapes has 128-row tiles)."""
    from . import _lib
    plan, a = call.plan, call.args
    memo = call.__dict__.setdefault("_orders", {})
    mkey = (int(spare), DG_balance_key(), int(torch.cuda.current_stream(call.y.device).cuda_stream))      # (the counter pair belongs to the stream)
    hit = memo.get(mkey)
    if hit is not None:                                   # (per call and grid: the pointers only -- this runs in front of every launch)
        a.spare_workgroups, a.tile_desc, a.row_perm, a.tile_counter = int(spare), hit[0], hit[1], hit[2]
        if hit[3] is not None:
            a.row_post = hit[3]
        if hit[5] is not None:
            a.guard_ws = hit[5]
        return
    a.spare_workgroups = int(spare)
    bal = None
    if not a.agg_out and plan.NV and _lib.lib().pna_fused_degree_tile_rows(a.F, a.N) == 64:
        bal = plan.fused_balance(_fused_grid(call.y.device, int(spare), plan.NV // 64))
    if bal is None:
        desc, perm, post = plan.fused_tables()[0], plan.perm, getattr(call, "post_g", None)
    else:
        desc, perm, src = bal
        post = getattr(call, "post_g", None)
        if post is not None and post is not plan.ones_rows():
            key = ("post", src.data_ptr())
            hit = call.__dict__.setdefault("_post_b", {}).get(key)
            if hit is None:
                hit = call._post_b[key] = post.view(plan.NV // 64, 64)[src].reshape(-1).contiguous()
            post = hit
    from . import degree_groups as DG
    counter = None
    if bal is not None and DG.FUSED_BALANCE == "dynamic":
        # ONE counter pair per (plan, stream) (zero between launches: the kernel's last workgroup resets it; launches on one stream are
        # ordered, launches on different streams get different pairs): a per-call tensor would put an allocation and a fill kernel in
        # front of every forward
        ckey = (str(call.y.device), int(torch.cuda.current_stream(call.y.device).cuda_stream))
        counters = plan.__dict__.setdefault("_tile_counters", {})
        counter = counters.get(ckey)
        if counter is None:                                  # (claims | finished: the kernel leaves them zero)
            if len(counters) > 16:
                counters.clear()
            counter = counters[ckey] = torch.zeros(2, dtype=torch.int32, device=call.y.device)
    a.tile_counter = None if counter is None else _lib.dev_ptr(counter, torch.int32, "tile_counter")
    gws = None
    if a.arith == _lib.FD_ARITH_GUARDED:                     # the guard's hand-over workspace belongs to the stream too
        gws = DG.guard_workspace(plan, call.y.device)
        a.guard_ws, a.guard_ws_bytes = _lib.dev_ptr(gws, torch.int32, "guard_ws"), gws.numel() * 4
    call._order_keep = (desc, perm, post, counter, gws)
    a.tile_desc, a.row_perm = _lib.dev_ptr(desc, torch.int32, "tile_desc"), _lib.dev_ptr(perm, torch.int32, "row_perm")
    rp = None
    if post is not None and a.row_post:
        rp = a.row_post = _lib.dev_ptr(post, torch.float32, "row_post")
    memo[mkey] = (a.tile_desc, a.row_perm, a.tile_counter, rp, call._order_keep, a.guard_ws if gws is not None else None)


def DG_balance_key():
    from . import degree_groups as DG
    return (DG.FUSED_BALANCE, DG.FUSED_TILE_COST, DG.FUSED_DYNAMIC_TAIL)


class FusedTowerCall:
    """One PNALayer forward (one tower, or T towers with divide_input=True; eval) on the one-kernel path after the node-level projection, cut into its launches like
    FusedDegreeCall: `group_rows()` = pna_fused_degree_f32 in tower mode, `rest_rows()` = gather with the destination term + the
    rows' own features + three-block contraction over the compact list of the rows no degree group holds."""

    def __init__(self, layer, graph, h, snorm_n, x_cat):
        from . import _lib, degree_groups as DG
        from .dgl.pna_layer import _row_scales
        import ctypes
        towers, mix = list(layer.towers), layer.mixing_network
        t0 = towers[0]
        Fi, V, dev = len(towers) * t0.in_dim, h.shape[0], h.device          # (Fi: message features per edge, ALL towers -- see _tower_flat_weights)
        P = x_cat.shape[1] // 2
        self.layer, self.graph, self.plan, self.t0, self.Fe = layer, graph, DG.plan_of(graph), t0, Fi
        plan = self.plan
        self.Wv, self.d, self.c, self.ones, self.K = Wv, d, c, ones, K = _tower_flat_weights(layer, towers, mix)
        N = Wv.shape[0]
        self.x_src, self.x_dst, self.h = x_cat[:, :Fi], x_cat[:, P:P + Fi], h
        self.scales = scales = _row_scales(graph, t0.scalers, t0.avg_d, dev)
        self.y = y = torch.empty(V, out_pitch(N), dtype=torch.float32, device=dev)[:, :N]
        self.res = res = h if layer.residual else None
        self.slope = float(mix.activation.negative_slope)
        if t0.graph_norm and snorm_n is not None:
            post = snorm_n.reshape(-1).to(torch.float32)[plan.perm_all().long()]
            self.post_g, self.post_r = post[:plan.NV].contiguous(), post[plan.NV:].contiguous()
        else:
            self.post_g, self.post_r = plan.ones_rows(), None
        desc, ids, n_rec = plan.fused_tables()
        self.keep = [desc, ids, x_cat]
        a = _lib.PnaFusedDegreeArgs()
        self.arith = DG.bind_fused_arith(a, self.keep, Wv, Fi, scales, plan, True, dev)
        a.tile_desc, a.tile_ids, a.n_records = _lib.dev_ptr(desc, torch.int32, "tile_desc"), _lib.dev_ptr(ids, torch.int32, "tile_ids"), n_rec
        a.x, a.ldx, a.x_rows, a.F, a.N = _lib.dev_ptr(self.x_src, torch.float32, "x_src"), x_cat.stride(0), V, Fi, N
        a.x_dst, a.ld_xdst = _lib.dev_ptr(self.x_dst, torch.float32, "x_dst"), x_cat.stride(0)
        a.h_self, a.ld_h = _lib.dev_ptr(h, torch.float32, "h"), h.stride(0)
        a.row_post = _lib.dev_ptr(self.post_g, torch.float32, "row_post")
        a.row_perm, a.M, a.n_nodes = _lib.dev_ptr(plan.perm, torch.int32, "row_perm"), plan.NV, V
        a.bias = _lib.dev_ptr(d, torch.float32, "bias")
        a.col_scale, a.col_shift = _lib.dev_ptr(ones, torch.float32, "col_scale"), _lib.dev_ptr(c, torch.float32, "col_shift")
        if res is not None:
            a.residual, a.ld_res = _lib.dev_ptr(res, torch.float32, "residual"), res.stride(0)
        a.y, a.ldy, a.relu, a.act_slope = _lib.dev_ptr(y, torch.float32, "y"), y.stride(0), 2, self.slope
        a.y_cols_writable = own_buffer_cols(y.stride(0), 0, N, N)
        self.args, self.ref = a, ctypes.byref(a)
        self.fn, self.check, self.stream = _lib.lib().pna_fused_degree_f32, _lib.check, _lib.stream_ptr(dev)
        _bind_tile_order(self, 0)

    def layer_F(self):
        return self.args.F

    def set_spare(self, on):
        from . import degree_groups as DG
        _bind_tile_order(self, DG.FUSED_SPARE_WGS if on else 0)

    def group_rows(self):
        self.check(self.fn(self.ref, self.stream), "pna_fused_degree_f32")
        return self.y

    def rest_rows(self):
        plan, graph, t0 = self.plan, self.graph, self.t0
        if plan.NR:
            from . import degree_groups as DG
            from .dgl.pna_layer import _avg_log_value
            Fi, K, N = self.Fe, self.K, self.Wv.shape[0]
            items, orow, hout, hs = plan.rest_items_by_node(graph)
            agg = torch.empty(plan.NRp, DG.agg_pitch(K), dtype=torch.float32, device=self.y.device)
            csr = graph.csr
            ops.segreduce(csr.rowptr, csr.col, self.x_src, Fi, t0.aggregators, (None,), n_tower=1, tower_stride_in=Fi, dst_term=self.x_dst,
                          out=agg, tower_stride_out=4 * Fi, heavy=hs, workspace=graph.workspace, items=items, out_row_of=orow, heavy_out=hout,
                          tune=dict(generic=2, rows_per_group=DG.REST_ROWS_PER_GROUP))
            ops.pack_rows(_unit_stride(self.h), plan.perm_all()[plan.NV:], out=agg[:, 4 * Fi:5 * Fi])
            rest_scales = plan.rest_scales(tuple(t0.scalers) + (_avg_log_value(t0.avg_d),), self.scales)
            ops.posttrans(agg[:, :K], K, self.Wv, rest_scales, self.d, out=self.y, row_post=self.post_r, col_scale=self.ones, col_shift=self.c,
                          leaky_slope=self.slope, residual=self.res, row_perm=plan.perm_rest, n_out=N)
        return self.y


def _tower_pass_weights(layer, towers, mix):
    """The collapsed weight of a divide_input=False layer (_tower_collapsed_weights: scaler blocks [C_1,s | .. | C_T,s | h panel (block 0)])
    cut for FusedMultiTowerCall: per tower the weight of its statistics, (N, S * 4 Fi) in scaler blocks [mean | max | min | std] -- the
    layer-proper format --, and ONE dense weight for everything that is linear in the row's own features h_v:
        mean / max / min of (a_u + b_v) = those of a_u, + b_v;  b_v = W_b,t h_v + beta_t   (models/dgl/pna_layer.py:35-40: the pretrans Linear)
        =>  dst term = sum_s scale_s(D_v) [deg_v > 0] (M_s h_v + beta_s),   M_s = sum_t (C_t,s^mean + C_t,s^max + C_t,s^min) W_b,t,
                                                                            beta_s = sum_t (C_t,s^mean + C_t,s^max + C_t,s^min) beta_t
    and the self panel W_self h_v: (N, in_dim + S in_dim) = [W_self | M_0 | .. | M_S-1], the layout of functional.posttrans with h_self = agg = h.
    Formed in float64, rounded once.  Cached on the layer."""
    Wv, d, c, ones, K = _tower_collapsed_weights(layer, towers, mix, False)
    hit = layer.__dict__.get("_pna_amd_pass_w")
    if hit is not None and hit[0] is Wv:
        return hit[1:]
    T, Fi = len(towers), towers[0].in_dim
    S = Wv.shape[1] // K
    N = Wv.shape[0]
    with torch.no_grad():
        Ws = []
        for t in range(T):
            W = torch.empty(N, S * 4 * Fi, dtype=torch.float32, device=Wv.device)
            for s_ in range(S):
                W[:, s_ * 4 * Fi:(s_ + 1) * 4 * Fi] = Wv[:, s_ * K + t * 4 * Fi:s_ * K + (t + 1) * 4 * Fi]
            Ws.append(W.contiguous())
        Wd = torch.zeros(N, (1 + S) * Fi, dtype=torch.float64, device=Wv.device)
        Wd[:, :Fi] = Wv[:, T * 4 * Fi:T * 4 * Fi + Fi].double()
        beta = torch.zeros(S, N, dtype=torch.float64, device=Wv.device)
        for t, tw in enumerate(towers):
            lin = tw.pretrans.fully_connected[0].linear
            Wb, bt = lin.weight[:, Fi:2 * Fi].double(), (lin.bias.double() if lin.bias is not None else torch.zeros(Fi, dtype=torch.float64, device=Wv.device))
            for s_ in range(S):
                Ct = Wv[:, s_ * K + t * 4 * Fi:s_ * K + (t + 1) * 4 * Fi].double()
                Wsum = Ct[:, :Fi] + Ct[:, Fi:2 * Fi] + Ct[:, 2 * Fi:3 * Fi]
                Wd[:, (1 + s_) * Fi:(2 + s_) * Fi] += Wsum @ Wb
                beta[s_] += Wsum @ bt
        # the rest rows (two-kernel path over their compact list): scaler blocks [the towers' aggregator columns | an N-column panel], the
        # panel the identity in block 0 (the identity scaler's) and zero elsewhere -- it carries the row's dense term, copied behind the
        # statistics of a_u in the aggregate buffer
        Ka = T * 4 * Fi
        Wr = torch.zeros(N, S * (Ka + N), dtype=torch.float32, device=Wv.device)
        for s_ in range(S):
            Wr[:, s_ * (Ka + N):s_ * (Ka + N) + Ka] = Wv[:, s_ * K:s_ * K + Ka]
        Wr[:, Ka:Ka + N] = torch.eye(N, dtype=torch.float32, device=Wv.device)
    res = (Ws, Wd.float().contiguous(), beta.float().contiguous(), Wr.contiguous(), d, c, ones)
    layer.__dict__["_pna_amd_pass_w"] = (Wv,) + res
    return res


DENSE_TERM_RESIDENT = True          # FusedMultiTowerCall.dense_term through pna_project_scaled_f32 where it applies (False: contraction + rank-S update)


class FusedMultiTowerCall:
    """One PNALayer forward with T towers over the WHOLE input (divide_input=False; models/dgl/pna_layer.py:137-139; eval) on the one-kernel
    path (round 6).  Every tower has its own projection of the input -- T x Fi message features per edge -- and a wavefront
    cannot hold T x 80 running statistics: the layer is one dense launch over the rows' own features (everything linear in h_v: the
    destination terms of all towers and the self panel, _tower_pass_weights) and T launches of pna_fused_degree_f32 -- the LAYER-PROPER
    instantiation, no node panels --, launch t gathering tower t's Fi features out of the (V, T P) source projection and adding
    W_t a_t to the partial sums of the launches before it (pna_fused_degree_args.pre_add); the last one applies bias, graph norm, the
    collapsed BatchNorm / mixing constants, LeakyReLU and the residual.  The aggregates never reach HBM; what does is T x 2 x 4 N bytes per
    row of partial sums.  `rest_rows()`: the rows no degree group holds, all towers at once on the two-kernel path."""

    def __init__(self, layer, graph, h, snorm_n, x_src):
        from . import _lib, degree_groups as DG
        from .dgl.pna_layer import _row_scales
        import ctypes
        towers, mix = list(layer.towers), layer.mixing_network
        t0 = towers[0]
        T, Fi, V, dev = len(towers), t0.in_dim, h.shape[0], h.device
        P = x_src.shape[1] // T
        self.layer, self.graph, self.plan, self.t0, self.T, self.Fi, self.P = layer, graph, DG.plan_of(graph), t0, T, Fi, P
        plan = self.plan
        Ws, self.Wd, self.beta, self.Wr, self.d, self.c, self.ones = _tower_pass_weights(layer, towers, mix)
        d, c, ones = self.d, self.c, self.ones
        N = Ws[0].shape[0]
        self.x_src, self.h = x_src, h
        self.scales = scales = _row_scales(graph, t0.scalers, t0.avg_d, dev)
        self.y = y = torch.empty(V, out_pitch(N), dtype=torch.float32, device=dev)[:, :N]
        # partial sums ping-pong between two buffers: a launch never writes the rows it reads -- the guard's second launch computes a
        # handed-over tile AGAIN from the same pre_add rows (in place it would add the tile's share twice: found by tools/fuzz_fused.py)
        self._part_full = torch.empty(V, out_pitch(N), dtype=torch.float32, device=dev)
        self.part = part = self._part_full[:, :N]
        self.part2 = torch.empty(V, out_pitch(N), dtype=torch.float32, device=dev)[:, :N] if T > 1 else None
        bp = layer.__dict__.get("_pna_amd_beta_pad")          # beta padded to the buffer's pitch: the rank-S update runs in place on the whole buffer
        if bp is None or bp[0] is not self.beta or bp[1].shape[1] != out_pitch(N):
            padded = torch.zeros(self.beta.shape[0], out_pitch(N), dtype=torch.float32, device=dev)
            padded[:, :N] = self.beta
            bp = layer.__dict__["_pna_amd_beta_pad"] = (self.beta, padded)
        self.beta_pad = bp[1]
        self.res = res = h if layer.residual else None
        self.slope = float(mix.activation.negative_slope)
        if t0.graph_norm and snorm_n is not None:
            post = snorm_n.reshape(-1).to(torch.float32)[plan.perm_all().long()]
            self.post_g, self.post_r = post[:plan.NV].contiguous(), post[plan.NV:].contiguous()
        else:
            self.post_g, self.post_r = plan.ones_rows(), None
        # the scalers of the dense term: zero for rows without in-edges (DGL leaves their aggregate -- the destination term with it -- at zero)
        hit = plan.__dict__.setdefault("_masked_scales", {})
        mkey = tuple(None if r is None else r.data_ptr() for r in scales)
        if mkey not in hit:
            if len(hit) > 8:
                hit.clear()
            live = (plan._deg > 0).to(torch.float32)
            ms = [live if r is None else (r * live).contiguous() for r in scales]
            hit[mkey] = (ms, torch.stack(ms, dim=1).contiguous())
        self.mscales, self.mscale_mat = hit[mkey]
        desc, ids, n_rec = plan.fused_tables()
        self.keep = [desc, ids, x_src, Ws]
        blocks = []
        for t in range(T):
            last = t == T - 1
            a = _lib.PnaFusedDegreeArgs()
            self.arith = DG.bind_fused_arith(a, self.keep, Ws[t], Fi, scales, plan, False, dev)
            a.tile_desc, a.tile_ids, a.n_records = _lib.dev_ptr(desc, torch.int32, "tile_desc"), _lib.dev_ptr(ids, torch.int32, "tile_ids"), n_rec
            xs = x_src[:, t * P:t * P + Fi]
            a.x, a.ldx, a.x_rows, a.F, a.N = _lib.dev_ptr(xs, torch.float32, "x_src"), x_src.stride(0), V, Fi, N
            a.row_perm, a.M, a.n_nodes = _lib.dev_ptr(plan.perm, torch.int32, "row_perm"), plan.NV, V
            src_buf, dst_buf = (part, self.part2) if t % 2 == 0 else (self.part2, part)
            a.pre_add, a.ld_pre_add = _lib.dev_ptr(src_buf, torch.float32, "pre_add"), src_buf.stride(0)
            if last:
                a.row_post = _lib.dev_ptr(self.post_g, torch.float32, "row_post")
                a.bias = _lib.dev_ptr(d, torch.float32, "bias")
                a.col_scale, a.col_shift = _lib.dev_ptr(ones, torch.float32, "col_scale"), _lib.dev_ptr(c, torch.float32, "col_shift")
                if res is not None:
                    a.residual, a.ld_res = _lib.dev_ptr(res, torch.float32, "residual"), res.stride(0)
                a.y, a.ldy, a.relu, a.act_slope = _lib.dev_ptr(y, torch.float32, "y"), y.stride(0), 2, self.slope
                a.y_cols_writable = own_buffer_cols(y.stride(0), 0, N, N)
            else:                                            # a partial sum: no bias, no factor, no activation
                a.y, a.ldy, a.relu = _lib.dev_ptr(dst_buf, torch.float32, "y"), dst_buf.stride(0), 0
                a.y_cols_writable = own_buffer_cols(dst_buf.stride(0), 0, N, N)
            blocks.append((a, ctypes.byref(a)))
        self.launch_order = blocks
        self.panel_args = [blocks[-1]] + blocks[:-1]         # (_bind_tile_order: the primary block is the one whose row_post follows the tile order)
        self.args, self.ref = blocks[-1]
        self.fn, self.check, self.stream = _lib.lib().pna_fused_degree_f32, _lib.check, _lib.stream_ptr(dev)
        _bind_tile_order(self, 0)

    def layer_F(self):
        return self.Fi

    def set_spare(self, on):
        from . import degree_groups as DG
        _bind_tile_order(self, DG.FUSED_SPARE_WGS if on else 0)

    def dense_term(self):
        """part = W_self h + sum_s scale_s [deg > 0] (M_s h + beta_s): one contraction launch over the rows' own features + a rank-S update."""
        h, Fi = _unit_stride(self.h), self.Fi
        if DENSE_TERM_RESIDENT and ops.project_scaled_applies(h, Fi, self.Wd.shape[0], 1 + len(self.mscales)):
            # every block of the weight resident in LDS, h read once, beta inside (0.66 -> 0.3 ms at 1 M rows: tools/multi_tower_time.py)
            return ops.project_scaled(h, Fi, self.Wd, self.mscale_mat, self.beta, True, out=self.part)
        ops.posttrans(h, Fi, self.Wd, self.mscales, None, h, out=self.part)
        torch.addmm(self._part_full, self.mscale_mat, self.beta_pad, out=self._part_full)   # (+= [deg > 0] scale_s beta_s; the padding columns stay unread)
        return self.part

    def prologue(self):
        """What both halves read: the dense term of every row (run_fused_call launches it in front of the fork)."""
        self.dense_term()

    def group_rows(self):
        for _, ref in self.launch_order:
            self.check(self.fn(ref, self.stream), "pna_fused_degree_f32")
        return self.y

    def rest_rows(self):
        plan, graph, t0, T, Fi, P = self.plan, self.graph, self.t0, self.T, self.Fi, self.P
        if plan.NR:
            from . import degree_groups as DG
            from .dgl.pna_layer import _avg_log_value
            N, Ka = self.Wr.shape[0], T * 4 * Fi
            K = Ka + N
            items, hout, hs = plan.rest_items(graph)
            agg = torch.empty(plan.NRp, DG.agg_pitch(K), dtype=torch.float32, device=self.y.device)
            csr = graph.csr
            # statistics of a_u alone, all towers (the destination term is in the dense term, like the group rows')
            ops.segreduce(csr.rowptr, csr.col, self.x_src, Fi, t0.aggregators, (None,), n_tower=T, tower_stride_in=P,
                          out=agg, tower_stride_out=4 * Fi, heavy=hs, workspace=graph.workspace, items=items, heavy_out=hout,
                          tune=dict(generic=2, rows_per_group=DG.REST_ROWS_PER_GROUP))
            ops.pack_rows(self.part, plan.perm_all()[plan.NV:], out=agg[:, Ka:K])
            rest_scales = plan.rest_scales(tuple(t0.scalers) + (_avg_log_value(t0.avg_d),), self.scales)
            ops.posttrans(agg[:, :K], K, self.Wr, rest_scales, self.d, out=self.y, row_post=self.post_r, col_scale=self.ones, col_shift=self.c,
                          leaky_slope=self.slope, residual=self.res, row_perm=plan.perm_rest, n_out=N)
        return self.y


def tower_layer_degree_fused_multi(layer, graph, h, snorm_n, x_src):
    """PNALayer.forward (eval, T towers, divide_input=False) after the node-level SOURCE projection x_src = [W_a,0 h | .. | W_a,T-1 h]
    (blocks of tower_projection_pitch columns): FusedMultiTowerCall."""
    return run_fused_call(FusedMultiTowerCall(layer, graph, h, snorm_n, x_src))


def tower_layer_degree_fused(layer, graph, h, snorm_n, x_cat):
    """PNALayer.forward (eval, ONE tower; models/dgl/pna_layer.py:130-145 over :33-76) after the node-level projection
    x_cat = [x_src | x_dst] (halves of tower_projection_pitch columns): gather over x_src, the destination term, the row's own
    features, the collapsed posttrans . graph norm . BatchNorm . mixing weight, LeakyReLU and the residual in ONE kernel for the
    rows of the degree groups; the aggregate never reaches HBM."""
    return run_fused_call(FusedTowerCall(layer, graph, h, snorm_n, x_cat))


class FusedDegreeCall:
    """One PNASimpleLayer forward on the one-kernel path, cut into its two halves so that bench.py can time them apart:
    `group_rows()` = pna_fused_degree_f32 (99.6 % of the benchmark graph's rows) -- one launch per output-column panel
    (degree_groups.fused_panels: one, unless the layer is wider than an instantiation) --, `rest_rows()` = gather + three-block
    contraction over the compact list of the rows no degree group holds.  Holds the argument blocks and every tensor they point into."""

    def __init__(self, layer, graph, h, x=None, out=None, agg_out=None, plan=None):
        from . import _lib, degree_groups as DG
        from .dgl.pna_layer import _row_scales
        import ctypes
        F, N = layer.in_dim, layer.out_dim
        self.layer, self.graph, self.plan = layer, graph, DG.plan_of(graph) if plan is None else plan      # (plan: a row block's)
        plan = self.plan
        self.x = x = graph.source_features(h) if x is None else x
        lin = layer.posttrans.fully_connected[0].linear
        self.scales = scales = _row_scales(graph, layer.scalers, layer.avg_d, h.device)
        V = h.shape[0]
        # (rows at a 16-byte aligned pitch: the next layer of a stack can read them in 16-byte strips, i.e. stay on this path)
        self.y = y = torch.empty(V, out_pitch(N), dtype=torch.float32, device=h.device)[:, :N] if out is None else out
        self.cs, self.ct, self.res = cs, ct, res = _layer_tail_operands(layer, h)
        desc, ids, n_rec = plan.fused_tables()
        self.keep = [desc, ids, lin, agg_out, h]
        aggs = tuple(layer.aggregators)
        panels = DG.fused_panels(F, N)
        if panels is None or not DG.aggregators_fusable(aggs):
            raise RuntimeError(f"pna_fused_degree: no instantiation for F={F}, N={N}, aggregators={aggs}")
        if agg_out is not None and (len(panels) != 1 or aggs != DG.STANDARD_AGGREGATORS):
            raise RuntimeError("pna_fused_degree: agg_out (verification) takes the four standard aggregators and a one-launch layer")
        self.panel_args = []
        whole = len(panels) == 1
        last_f0 = panels[-1][0]                               # (the feature panel whose launches apply the epilogue)
        self.part = None
        if any(f0 != last_f0 for f0, _, _, _ in panels):      # feature panels: partial sums between the launches
            self.part = torch.empty(V, out_pitch(N), dtype=torch.float32, device=h.device)[:, :N]
        for f0, f1, c0, c1 in panels:
            a = _lib.PnaFusedDegreeArgs()
            final, first = f0 == last_f0, f0 == 0
            self.arith = DG.bind_fused_arith(a, self.keep, lin.weight, F, scales, plan, False, h.device, verification=agg_out is not None,
                                             rows=None if whole else (c0, c1), aggregators=aggs, feats=None if (f0, f1) == (0, F) else (f0, f1))
            a.tile_desc, a.tile_ids, a.n_records = _lib.dev_ptr(desc, torch.int32, "tile_desc"), _lib.dev_ptr(ids, torch.int32, "tile_ids"), n_rec
            a.x, a.ldx, a.x_rows, a.F, a.N = _lib.dev_ptr(x[:, f0:f1], torch.float32, "x"), x.stride(0), x.shape[0], f1 - f0, c1 - c0
            a.row_perm, a.M, a.n_nodes = _lib.dev_ptr(plan.perm, torch.int32, "row_perm"), plan.NV, V
            sl = lambda t: None if t is None else t[c0:c1]      # noqa: E731  (a panel's slice of a per-column vector: still unit stride)
            if not first:
                a.pre_add, a.ld_pre_add = _lib.dev_ptr(self.part[:, c0:c1], torch.float32, "pre_add"), self.part.stride(0)
            if final:
                a.bias = _lib.dev_ptr(sl(lin.bias), torch.float32, "bias")
                a.col_scale, a.col_shift = _lib.dev_ptr(sl(cs), torch.float32, "col_scale"), _lib.dev_ptr(sl(ct), torch.float32, "col_shift")
                if res is not None:
                    a.residual, a.ld_res = _lib.dev_ptr(res[:, c0:c1], torch.float32, "residual"), res.stride(0)
                a.y, a.ldy, a.relu = _lib.dev_ptr(y[:, c0:c1], torch.float32, "y"), y.stride(0), 1
                if out is None:
                    a.y_cols_writable = own_buffer_cols(y.stride(0), c0, c1, N)
            else:                                            # a partial sum: no bias, no BatchNorm, no activation, no residual
                a.y, a.ldy, a.relu = _lib.dev_ptr(self.part[:, c0:c1], torch.float32, "y"), self.part.stride(0), 0
                a.y_cols_writable = own_buffer_cols(self.part.stride(0), c0, c1, N)
            if agg_out is not None:
                a.agg_out, a.ld_agg = _lib.dev_ptr(agg_out, torch.float32, "agg_out"), agg_out.stride(0)
            self.panel_args.append((a, ctypes.byref(a)))
        self.args, self.ref = self.panel_args[0]
        self.fn, self.check, self.stream = _lib.lib().pna_fused_degree_f32, _lib.check, _lib.stream_ptr(h.device)
        _bind_tile_order(self, 0)

    def layer_F(self):
        return self.layer.in_dim

    def set_spare(self, on):
        from . import degree_groups as DG
        _bind_tile_order(self, DG.FUSED_SPARE_WGS if on else 0)

    def group_rows(self):
        for _, ref in self.panel_args:
            self.check(self.fn(ref, self.stream), "pna_fused_degree_f32")
        return self.y

    def rest_rows(self):
        plan, layer, graph = self.plan, self.layer, self.graph
        if plan.NR:
            from . import degree_groups as DG
            F = layer.in_dim
            K = 4 * F                                        # (the four standard statistics whatever the layer's list: _rest_posttrans)
            Kb = 5 * F if "sum" in layer.aggregators else K  # (+ room for the sum block)
            items, hout, hs = plan.rest_items(graph)
            agg = torch.empty(plan.NRp, DG.agg_pitch(Kb), dtype=torch.float32, device=self.y.device)[:, :Kb]
            csr = graph.csr
            ops.segreduce(csr.rowptr, csr.col, _unit_stride(self.x), F, DG.STANDARD_AGGREGATORS, (None,), tower_stride_in=F, out=agg[:, :K],
                          heavy=hs, workspace=graph.workspace, items=items, heavy_out=hout,
                          tune=dict(generic=2, rows_per_group=DG.REST_ROWS_PER_GROUP))
            _rest_posttrans(layer, graph, agg, plan, self.scales, self.y, self.cs, self.ct, self.res)
        return self.y


def simple_layer_degree_fused(layer, graph, h, x=None, out=None, agg_out=None):
    """PNASimpleLayer.forward (eval) with the group rows in ONE kernel (pna_fused_degree_f32): gather, the four
    aggregators, the combined scaler block W_D and the posttrans contraction with its BatchNorm / ReLU / residual epilogue; the
    4F aggregate of those rows never reaches HBM.  The rows no degree group holds (rare degrees, hub rows: 0.4 % of the benchmark
    graph's rows, 5 % of its edges) take the two-kernel path over their compact list.  `x`: the source table (halo in place on
    a sharded graph); `agg_out` (verification): (plan.NV, >= 4F) receives the statistics the contraction consumed."""
    src = graph.source_features(h) if x is None else x
    return run_fused_call(FusedDegreeCall(layer, graph, h, x=src, out=out, agg_out=agg_out))


def simple_layer_degree_grouped(layer, graph, h):
    """PNASimpleLayer.forward (eval) with the rows grouped by in-degree (pna_amd/degree_groups.py): the gather writes the
    aggregate in degree order, the contraction multiplies every tile by its degree's combined weight W_D = sum_s s_s(D) W_s
    (one scaler block instead of three), the rows of rare degrees and the hub rows take the ordinary three-block contraction
    over a compacted list."""
    from . import degree_groups as DG
    plan = DG.plan_of(graph)
    from .graph import Graph
    aggs = tuple(layer.aggregators)
    if type(graph) is Graph and DG.fused_applies(graph, h, layer.in_dim, layer.out_dim, aggs):
        return simple_layer_degree_fused(layer, graph, h, x=h)
    # A shard (HaloGraph).  The two-kernel path cuts its gather into the rows that read only local sources -- aggregated while
    # the halo exchange is in flight -- and the rest; the one-kernel path needs the whole [local | halo] table first.  Which one
    # pays depends on how many rows the overlap covers: on a graph without locality (the benchmark graph: 1 % interior rows) the
    # overlap hides next to nothing and the one-kernel layer saves a third of the compute, so: exchange, then ONE kernel over the
    # extended table (features in the shard's resident table at a 16-byte aligned pitch); with many interior rows, the overlap.
    from .shard import HaloGraph
    two = DG.two_kernel_applies(layer.out_dim, len(layer.scalers), aggs)
    if (isinstance(graph, HaloGraph) and DG.FUSED and graph._resident(h) and graph._pending is None
            and (graph.interior_fraction() < DG.FUSED_HALO_MAX_INTERIOR or not two)):
        x_ext = graph._ext[:, : h.shape[1]]
        if DG.fused_applies(graph, x_ext, layer.in_dim, layer.out_dim, aggs):
            return simple_layer_degree_fused(layer, graph, h, x=graph.source_features(h))
    if not two:
        # (an operator set / width only the one-kernel layer serves, and it does not apply to this call: PNASimpleLayer.forward
        # catches this and takes the ordinary path)
        raise RuntimeError("pna_amd: the hand-scheduled kernel was required (one-kernel layer) but does not apply to this call")
    return degree_grouped_posttrans(layer, graph, h, degree_grouped_aggregate(layer, graph, h, plan), plan)


class _SmallSimplePlan(_SmallTowerPlan):
    """PNASimpleLayer (models/dgl/pna_layer.py:197-216, eval mode) on the same one-call kernel: a tower layer with ONE tower,
    the identity as pretrans (messages are the raw source features: x_cat = [I ; 0] h), no self panel in the posttrans
    (W_h = 0), and the identity as mixing network with the layer's ReLU and residual in its epilogue.  Products with 1 and sums
    with 0 are exact: the result is the three-kernel path's up to the summation order of the contraction.  One difference for
    NON-FINITE inputs: since round 4 the kernel multiplies ZEROS (not h) with the zero self block (`no_self_panel`), so a
    node whose own feature is Inf / NaN keeps the finite output the reference and the three-kernel path give it."""

    def __init__(self, layer):
        import ctypes
        from . import _lib
        lin = layer.posttrans.fully_connected[0].linear
        bn = layer.batchnorm_h if layer.batch_norm else None
        if bn is not None and bn.training:
            raise RuntimeError("only an eval-mode BatchNorm (running statistics) can be folded into the epilogue")
        self.ts = [x for x in (lin.weight, lin.bias) if x is not None] + \
                  ([x for x in (bn.weight, bn.bias, bn.running_mean, bn.running_var) if x is not None] if bn is not None else [])
        self.versions = self._state()
        Fi, Fo, S = layer.in_dim, layer.out_dim, len(layer.scalers)
        self.T, self.Fi, self.Fo, self.S, self.divide_input = 1, Fi, Fo, S, False
        dev = lin.weight.device
        with torch.no_grad():
            eye = torch.eye(Fi, device=dev)
            w_ext = torch.cat([torch.zeros(Fo, Fi, device=dev), lin.weight], dim=1).contiguous()      # [h panel = 0 | scaler blocks]
            keep = dict(proj_img=ops.pack_small(torch.cat([eye, torch.zeros_like(eye)], dim=0).contiguous()),
                        proj_bias=torch.zeros(2 * Fi, device=dev),
                        post_img=ops.pack_tower_post([w_ext], Fi, Fo, S),
                        post_bias=lin.bias.detach().contiguous() if lin.bias is not None else None,
                        mix_img=ops.pack_small(torch.eye(Fo, device=dev)), mix_bias=None)
            if bn is not None:
                keep["col_scale"], keep["col_shift"] = (x.contiguous() for x in _fold_batchnorm(bn))
        self.keep, self.device, self.width, self.xw, self.graph_norm = keep, dev, Fo, 2 * Fi, False
        a = _lib.PnaTowerLayerArgs()
        a.n_tower, a.Fi, a.Fo, a.divide_input, a.n_scaler = 1, Fi, Fo, 0, S
        for k, v in keep.items():
            if v is not None:
                setattr(a, k, _lib.dev_ptr(v, torch.float32, k))
        a.No, a.mix_act = Fo, 1                                # ReLU (pna_layer.py:211)
        a.no_self_panel = 1                                    # (zeros, not h, against the zero self block: Inf / NaN own features stay out)
        self.args, self.ref = a, ctypes.byref(a)
        self.fn = _lib.lib().pna_tower_layer_f32
        self.check, self.stream_ptr = _lib.check, _lib.stream_ptr


SMALL_SIMPLE_ROWS = int(os.environ.get("PNA_AMD_SMALL_SIMPLE_ROWS", "4096"))   # PNASimpleLayer batches up to this many nodes (measured
# crossover with the three-kernel path at hidden 80: 0.026 vs 0.048 ms at 1.6-3 k nodes, equal at 6.4 k, 0.087 vs 0.056 at 13 k:
# profiles/r02_small_simple_layer.json)


def simple_layer_small(layer, graph, h, row_scales):
    """PNASimpleLayer.forward (eval) through pna_tower_layer_f32; the plan is cached on the layer."""
    plan = layer.__dict__.get("_pna_amd_small")
    if plan is None or plan.stale():
        plan = _SmallSimplePlan(layer)
        layer.__dict__["_pna_amd_small"] = plan
    return plan.run(graph, h, None, row_scales, layer.residual)


def tower_layer_small(owner, towers, mix, graph, h, snorm_n, row_scales, divide_input, residual, etab=None):
    """models/dgl/pna_layer.py:133-148 in eval mode through pna_tower_layer_f32.  `mix`: the mixing FCLayer (Linear + LeakyReLU /
    ReLU / none, no batch-norm) or None.  The plan is cached on `owner` (dropped by PNALayer._apply on device / dtype moves)."""
    plan = owner.__dict__.get("_pna_amd_small")
    if plan is None or plan.divide_input != divide_input or plan.stale():
        plan = _SmallTowerPlan(towers, mix, divide_input)
        owner.__dict__["_pna_amd_small"] = plan
    return plan.run(graph, h, snorm_n, row_scales, residual, etab)


class SimpleLayerRows:
    """layer_rows callback of shard.BlockPipeline for a stack of PNASimpleLayer (eval): rows [r0, r1) of layer l from the complete
    [local | halo] table into the next table -- the gather over the block's work list (hub rows ride with block 0: their
    finalize writes wherever they live) and the three-block contraction over the block's slice of the aggregate.  Every row is
    computed by the same kernels in the same order as on one GPU: bit-identical to the unsharded ordinary path."""

    wants_full_out = True          # BlockPipeline hands __call__ the whole next table (the one-kernel layer scatters rows by node id)

    def __init__(self, layers, graph, n_blocks, fused=None):
        from . import degree_groups as DG
        self.layers, self.g = list(layers), graph
        l0 = self.layers[0]
        self.F = l0.in_dim
        self.fused = DG.FUSED if fused is None else bool(fused)
        self.n_blocks, self._plans, self._calls = int(n_blocks), {}, {}
        if any(l.in_dim != self.F or l.out_dim != self.F or tuple(l.aggregators) != ("mean", "max", "min", "std") or l.training for l in self.layers):
            raise ValueError("SimpleLayerRows: a stack of eval-mode PNASimpleLayer(F -> F) with the four standard aggregators")
        V = graph.num_nodes
        self.agg = torch.empty(V, DG.agg_pitch(4 * self.F), dtype=torch.float32, device=graph.device)
        hs = graph.heavy_schedule()
        deg = graph.csr.rowptr[1:] - graph.csr.rowptr[:-1]
        light = (deg <= hs.threshold) if hs.threshold > 0 else torch.ones_like(deg, dtype=torch.bool)
        rows = torch.arange(V, device=graph.device)
        self.items = []
        for b in range(n_blocks):
            r0, r1 = (V * b) // n_blocks, (V * (b + 1)) // n_blocks
            self.items.append(graph.work_items_subset(light & (rows >= r0) & (rows < r1), include_heavy=(b == 0)))
        self.hs = hs

    def block_plan(self, b, r0, r1):
        """The degree plan of row block b, or None when the blocks do not qualify for the one-kernel layer (a block without a
        degree value that fills a tile, mostly leftover rows).  Blocks 1.. hold only the rows of their range whose degree fills
        tiles INSIDE the block (one launch each); block 0 takes, besides its own rows, the hub rows and every block's leftovers
        (its chain of small rest-row launches runs once per layer, first: a row may be ready earlier than its block, never later)."""
        from . import degree_groups as DG
        if not self._plans:
            V, B = self.g.num_nodes, self.n_blocks
            bounds = [(V * i) // B for i in range(B + 1)]
            plans, left = {}, []
            for i in range(1, B):
                plans[i] = DG.DegreePlan(self.g, row_range=(bounds[i], bounds[i + 1]), with_heavy=False, drop_rest=True)
                left.append(plans[i].dropped_rest)
            plans[0] = DG.DegreePlan(self.g, row_range=(bounds[0], bounds[1]), with_heavy=True, extra_rows=torch.cat(left) if left else None)
            ok = (all(p.G > 0 and p.fused_tables() is not False for p in plans.values())
                  and plans[0].NR <= DG.MAX_REST_FRACTION * max(1, V // B))
            self._plans = plans if ok else {i: None for i in range(B)}
        return self._plans[b]

    def __call__(self, l, table, r0, r1, out, b):
        from .dgl.pna_layer import _row_scales
        from . import degree_groups as DG
        layer, g, F = self.layers[l], self.g, self.F
        csr = g.csr
        K = 4 * F
        x = table[:, :F]
        n_local = g.num_nodes
        full = out.shape[0] == table.shape[0]                 # (BlockPipeline: the whole next table; older callers: its rows [r0, r1))
        if self.fused and full and r1 > r0 and table.is_cuda and len(layer.scalers) == 3 and DG.fused_shape_ok(x, F, layer.out_dim):
            plan = self.block_plan(b, r0, r1)
            if plan is not None:
                # the block through the one-kernel layer: its own degree groups (pna_fused_degree_f32) + its rest rows; rows are
                # scattered to their node ids in the next table (1.47 -> ~0.85 ms / layer of compute at C3 shape)
                # the cached call snapshots the packed weight images and the folded BatchNorm scale / shift: the parameters' (version,
                # address) belong in the key, or an optimizer step / load_state_dict between two runs leaves the group rows on stale
                # weights while the rest rows read the live ones
                key = (l, b, table.data_ptr(), out.data_ptr()) + _simple_layer_state(layer)
                call = self._calls.get(key)
                if call is None:
                    if len(self._calls) > 4 * len(self.layers) * self.n_blocks:
                        self._calls.clear()
                    call = self._calls[key] = FusedDegreeCall(layer, g, table[:n_local, :F], x=x, out=out[:n_local, :layer.out_dim], plan=plan)
                run_fused_call(call)
                return
        if full:
            out = out[r0:r1]
        items = self.items[b]
        if items.shape[0]:
            ops.segreduce(csr.rowptr, csr.col, x, F, layer.aggregators, (None,), tower_stride_in=F, out=self.agg,
                          heavy=self.hs if b == 0 else None, workspace=g.workspace, items=items, tune=dict(generic=2))
        if r1 > r0:
            lin = layer.posttrans.fully_connected[0].linear
            scales = _row_scales(g, layer.scalers, layer.avg_d, table.device)
            cs, ct, _ = _layer_tail_operands(layer, x)
            ops.posttrans(self.agg[r0:r1, :K], K, lin.weight, [None if r is None else r[r0:r1] for r in scales], lin.bias, out=out[:, :layer.out_dim],
                          col_scale=cs, col_shift=ct, relu=True, residual=x[r0:r1] if layer.residual else None)
