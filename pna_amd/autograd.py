"""Backward passes of the fused operators.

`AggregateFn`  : forward = pna_segreduce_fwd_f32 (with argmax/argmin recorded), backward =
                 pna_segreduce_bwd_f32 (re-gather + fp32 atomic scatter; see include/pna_amd.h).
`PosttransFn`  : forward = pna_posttrans_f32; its backward is three plain GEMMs (grad_agg, grad_weight,
                 grad_h) -- library matmuls through torch (rocBLAS/hipBLASLt), like the other non-hot
                 linears of the path.
`BnTailFn`     : forward = pna_bn_tail_fwd_f32 (batch-statistics BatchNorm1d + ReLU + residual of PNASimpleLayer's
                 training forward, models/dgl/pna_layer.py:207-213), backward = pna_bn_tail_bwd_f32: two streaming
                 passes each instead of the library's ~10.
No CPU or eager fallback for the forward; gradients of the degree scalers themselves are not needed
(they depend on the graph only).
"""
import ctypes
import os

import torch

from . import _lib, ops

_STAT_AGGS = ("std", "var")


class AggregateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, x, dst_term, edge_term, F, aggregators, n_tower, row_scales, edge_resident,
                edge_weight, col_override):
        if len(set(aggregators)) != len(aggregators):
            raise NotImplementedError("pna_amd: backward needs distinct aggregators")
        csr = graph.csr
        col = None if edge_resident else (csr.col if col_override is None else col_override)
        aggs = list(aggregators)
        # the backward of std / var needs the forward mean (and std): make sure they are computed
        extra = []
        if any(a in _STAT_AGGS for a in aggs):
            if "mean" not in aggs:
                extra.append("mean")
            if "std" not in aggs:
                extra.append("std")
        elif "var_raw" in aggs and "mean" not in aggs:
            extra.append("mean")
        all_aggs = aggs + extra
        want_arg = any(a in ("max", "min") for a in aggs)
        res = ops.segreduce(csr.rowptr, col, x, F, all_aggs, [None], n_tower=n_tower, tower_stride_in=F,
                            dst_term=dst_term, edge_term=edge_term, edge_weight=edge_weight, want_arg=want_arg,
                            heavy=graph.heavy_schedule(), workspace=graph.workspace, items=graph.work_items())
        ident, amx, amn = res if want_arg else (res, None, None)       # (V, T*A'*F) identity-scaled
        V, T, A, A2, S = ident.shape[0], max(1, n_tower), len(aggs), len(all_aggs), len(row_scales)
        iv = ident.view(V, T, A2, F)
        if S == 1 and row_scales[0] is None and not extra:
            out = ident
        else:
            blocks = [iv[:, :, :A] if rs is None else iv[:, :, :A] * rs.view(V, 1, 1, 1) for rs in row_scales]
            out = torch.stack(blocks, dim=2).reshape(V, T * S * A * F)  # tower-major, scaler-major, aggregator-major
        ctx.graph, ctx.F, ctx.aggs, ctx.all_aggs, ctx.T, ctx.col = graph, F, aggs, all_aggs, T, col
        ctx.row_scales = row_scales
        ctx.needs = (x.requires_grad, dst_term is not None and dst_term.requires_grad,
                     edge_term is not None and edge_term.requires_grad)
        ctx.save_for_backward(x, dst_term, edge_term, ident, amx, amn, edge_weight)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, dst_term, edge_term, ident, amx, amn, edge_weight = ctx.saved_tensors
        graph, F, aggs, all_aggs, T = ctx.graph, ctx.F, ctx.aggs, ctx.all_aggs, ctx.T
        V, A, A2, S = ident.shape[0], len(aggs), len(all_aggs), len(ctx.row_scales)
        go = grad_out.reshape(V, T, S, A, F)
        # fold the degree scalers into the gradient of the unscaled aggregates: G_a = sum_s scale_s * dOut[s, a]
        gagg = None
        for s, rs in enumerate(ctx.row_scales):
            term = go[:, :, s] if rs is None else go[:, :, s] * rs.view(V, 1, 1, 1)
            gagg = term if gagg is None else gagg + term
        gagg = gagg.contiguous().view(V, T * A * F)
        csr = graph.csr
        dev = x.device
        need_x, need_d, need_e = ctx.needs
        if edge_weight is not None or "var_raw" in aggs:
            gx, gd, ge = _backward_edges_torch(graph, ctx.col, x, dst_term, edge_term, edge_weight, ident, amx, amn, gagg, aggs,
                                               all_aggs, T, F, need_x, need_d, need_e)
            return (None, gx, gd, ge, None, None, None, None, None, None, None)
        if ctx.col is not None and edge_term is None and ctx.col is csr.col and os.environ.get("PNA_AMD_BWD", "pull") == "pull":
            gx, gd = _backward_pull(graph, x, dst_term, ident, amx, amn, gagg, aggs, all_aggs, T, F, need_x, need_d)
            return (None, gx, gd, None, None, None, None, None, None, None, None)
        b = _lib.PnaSegreduceBwdArgs()
        b.rowptr = _lib.dev_ptr(csr.rowptr, torch.int32, "rowptr")
        b.col = _lib.dev_ptr(ctx.col, torch.int32, "col")
        b.V, b.F = V, F
        b.x, b.ldx = _lib.dev_ptr(x, torch.float32, "x"), x.stride(0)
        if dst_term is not None:
            b.dst_term, b.ld_dst = _lib.dev_ptr(dst_term, torch.float32, "dst_term"), dst_term.stride(0)
        if edge_term is not None:
            b.edge_term, b.ld_edge = _lib.dev_ptr(edge_term, torch.float32, "edge_term"), edge_term.stride(0)
        b.n_tower, b.n_aggr, b.tower_stride_in = T, A, F
        for i, name in enumerate(aggs):
            b.aggr[i] = _lib.AGG_CODES[name]
        b.gagg, b.ld_g, b.tower_stride_g = _lib.dev_ptr(gagg, torch.float32, "gagg"), gagg.stride(0), A * F
        if any(a in _STAT_AGGS for a in aggs):
            base = ident.data_ptr()
            b.mean = ctypes.c_void_p(base + 4 * all_aggs.index("mean") * F)
            b.stdv = ctypes.c_void_p(base + 4 * all_aggs.index("std") * F)
            if "var" in all_aggs:
                b.var = ctypes.c_void_p(base + 4 * all_aggs.index("var") * F)
            b.ld_stat, b.tower_stride_stat = ident.stride(0), A2 * F
        if amx is not None:
            b.argmax, b.argmin, b.ld_arg = (_lib.dev_ptr(amx, torch.int32, "argmax"),
                                            _lib.dev_ptr(amn, torch.int32, "argmin"), amx.stride(0))
        gx = gd = ge = None
        if need_x:
            gx = torch.zeros(x.shape[0], T * F, dtype=torch.float32, device=dev) if ctx.col is not None else \
                torch.empty(x.shape[0], T * F, dtype=torch.float32, device=dev)
            b.grad_x, b.ld_gx = _lib.dev_ptr(gx, torch.float32, "grad_x"), gx.stride(0)
        if need_d:
            gd = torch.zeros(V, T * F, dtype=torch.float32, device=dev)
            b.grad_dst, b.ld_gd = _lib.dev_ptr(gd, torch.float32, "grad_dst"), gd.stride(0)
        if need_e:
            ge = torch.empty(edge_term.shape[0], T * F, dtype=torch.float32, device=dev)
            b.grad_edge, b.ld_ge = _lib.dev_ptr(ge, torch.float32, "grad_edge"), ge.stride(0)
        hs = graph.heavy_schedule()
        if hs.n_heavy > 0:
            b.heavy_threshold, b.seg_len, b.n_heavy, b.n_seg = hs.threshold, hs.seg_len, hs.n_heavy, hs.n_seg
            b.heavy_rows = _lib.dev_ptr(hs.heavy_rows, torch.int32, "heavy_rows")
            b.heavy_segptr = _lib.dev_ptr(hs.heavy_segptr, torch.int32, "heavy_segptr")
            b.seg_heavy = _lib.dev_ptr(hs.seg_heavy, torch.int32, "seg_heavy")
        if need_x or need_d or need_e:
            rc = _lib.lib().pna_segreduce_bwd_f32(ctypes.byref(b), _lib.stream_ptr(dev))
            _lib.check(rc, "pna_segreduce_bwd_f32")
        if gx is not None and x.shape[1] != T * F:
            full = torch.zeros_like(x)
            full[:, :T * F] = gx
            gx = full
        return (None, gx, gd, ge, None, None, None, None, None, None, None)


def _backward_edges_torch(graph, col, x, dst_term, edge_term, w, ident, amx, amn, gagg, aggs, all_aggs, T, F, need_x, need_d, need_e):
    """Backward of the aggregation with per-edge WEIGHTS (the dense variant's adjacency used as a weight,
    models/pytorch/pna/aggregators.py:25,:69: D = sum_k w_k, s = sum_k w_k m_k, q = sum_k w_k m_k^2; max / min over the edges
    with w_k > 0) and of the unclamped PyG variance (`var_raw`, pytorch_geometric/aggregators.py:25-28), as torch tensor ops
    over the edge list:

        dL/dm_k = w_k [ G_sum + G_mean / D + (G_var [var > 0] + G_var_raw + G_std [var > 0] / (2 std)) (2 / D) (m_k - mean) ]
                  + [k = argmax] G_max + [k = argmin] G_min

    The forward is the HIP kernel; these configurations (small dense graphs, the exotic registry entries) are off the hot path,
    so their backward does not have a kernel of its own -- like the GEMMs of the posttrans backward it runs on the GPU
    through library ops."""
    csr = graph.csr
    V, TF = ident.shape[0], T * F
    A, A2 = len(aggs), len(all_aggs)
    E = int(csr.rowptr[-1].item())
    dev = x.device
    row = csr.row.long()
    cx = col.long() if col is not None else torch.arange(E, device=dev)
    m = x[cx][:, :TF]
    if dst_term is not None:
        m = m + dst_term[row][:, :TF]
    if edge_term is not None:
        m = m + edge_term[:, :TF]
    ww = torch.ones(E, device=dev) if w is None else w.reshape(-1).to(torch.float32)
    D = torch.zeros(V, device=dev).index_add_(0, row, ww).clamp_min(1e-30).unsqueeze(1)
    g4 = gagg.view(V, T, A, F)
    iv = ident.view(V, T, A2, F)

    def G(name):
        return g4[:, :, aggs.index(name)].reshape(V, TF) if name in aggs else None

    base = torch.zeros(V, TF, device=dev)
    if G("sum") is not None:
        base = base + G("sum")
    if G("mean") is not None:
        base = base + G("mean") / D
    r2 = None
    if any(a in aggs for a in ("std", "var", "var_raw")):
        mean = iv[:, :, all_aggs.index("mean")].reshape(V, TF) if "mean" in all_aggs else \
            torch.zeros(V, TF, device=dev).index_add_(0, row, m * ww.unsqueeze(1)) / D
        r2 = torch.zeros(V, TF, device=dev)
        if "std" in aggs or "var" in aggs:
            std = iv[:, :, all_aggs.index("std")].reshape(V, TF)
            pos = (std * std - 1e-5) > 0                      # var > 0  <=>  std > sqrt(eps): the relu mask of the forward
            if G("var") is not None:
                r2 = r2 + G("var") * pos
            if G("std") is not None:
                r2 = r2 + G("std") / (2 * std) * pos
        if G("var_raw") is not None:
            r2 = r2 + G("var_raw")
        r2 = r2 * (2.0 / D)
    dm = base[row]
    if r2 is not None:
        dm = dm + r2[row] * (m - mean[row])
    dm = dm * ww.unsqueeze(1)
    for name, arg in (("max", amx), ("min", amn)):
        g_ = G(name)
        if g_ is None:
            continue
        a_ = arg[:, :TF].long()
        ok = a_ >= 0
        cols = torch.arange(TF, device=dev).unsqueeze(0).expand(V, TF)
        dm.index_put_((a_[ok], cols[ok]), g_[ok], accumulate=True)
    gx = gd = ge = None
    if need_x:
        gx = torch.zeros_like(x)
        gx[:, :TF].index_add_(0, cx, dm)
    if need_d:
        gd = torch.zeros_like(dst_term)
        gd[:, :TF].index_add_(0, row, dm)
    if need_e:
        ge = torch.zeros_like(edge_term)
        ge[:, :TF] = dm
    return gx, gd, ge


def _column_sums(t):
    """t.sum(0) for a tall (M, N) tensor.  torch's reduction over the long outer dimension of a (1e6, 75) tensor takes
    15 ms on this stack (rocBLAS gemv: 10 ms, ones @ t: 1.5 ms); folding 64 rows into one (M/64, 64*N) row first gives the
    reduce kernel a wide inner dimension: 0.07 ms."""
    M, N = t.shape
    t = t.contiguous()
    main = (M // 64) * 64
    out = t[:main].view(-1, 64 * N).sum(0).view(64, N).sum(0) if main else torch.zeros(N, dtype=t.dtype, device=t.device)
    return out + t[main:].sum(0) if main < M else out


def _tall_tn(p, q, chunks=256):
    """p^T q for tall (M, a), (M, b) operands: the reduction runs over M.  As ONE library GEMM the 1e6-long reduction of the C3
    weight gradient ((225 x 1e6)(1e6 x 300)) runs at 44 TF/s -- the output has only 24 tiles; cut into 256 row slabs as a batched
    GEMM + a sum over the slabs it takes 1.27 instead of 3.06 ms (tools/exp_dw_gemm.py [removed in round 5: git history]), and the slab-wise summation is the more
    accurate order."""
    M = p.shape[0]
    if not p.is_cuda or M < 64 * chunks:
        return p.t() @ q
    Mc = M // chunks * chunks
    pc = p if p.is_contiguous() else p.contiguous()
    qc = q if q.is_contiguous() else q.contiguous()
    out = torch.bmm(pc[:Mc].view(chunks, Mc // chunks, -1).transpose(1, 2), qc[:Mc].view(chunks, Mc // chunks, -1)).sum(0)
    if Mc < M:
        out = out + pc[Mc:].t() @ qc[Mc:]
    return out


def _argscatter_sorted(gx, col, amx, amn, gagg, aggs, T, F):
    """grad_x[col[argmax[v, c]], c] += G_max[v, c] (and the min term) without atomics: deterministic, for small graphs."""
    V, TF, A = amx.shape[0], T * F, len(aggs)
    g4 = gagg[:, :T * A * F].reshape(V, T, A, F)
    cols = torch.arange(TF, device=gx.device)
    keys, vals = [], []
    for name, arg in (("max", amx), ("min", amn)):
        if name in aggs and arg is not None:
            G = g4[:, :, aggs.index(name), :].reshape(V, TF)
            e = arg[:, :TF].long()
            ok = e >= 0
            key = col[e.clamp(min=0)].long() * TF + cols
            keys.append(key[ok])
            vals.append(G[ok])
    if not keys:
        return
    key, val = torch.cat(keys), torch.cat(vals)
    order = torch.sort(key, stable=True).indices
    key, val = key[order], val[order].contiguous()
    uniq, counts = torch.unique_consecutive(key, return_counts=True)
    sums = torch.segment_reduce(val, "sum", lengths=counts)
    flat = gx.view(-1)                                       # (contiguous (rows, TF) by construction)
    flat[uniq] = flat[uniq] + sums


def _backward_pull(graph, x, dst_term, ident, amx, amn, gagg, aggs, all_aggs, T, F, need_x, need_d, row_of=None, packed_rows=None, node_of=None):
    """Gradient of the aggregation w.r.t. the source table / destination term WITHOUT one atomic per edge and feature
    (the scatter kernel: 750 M atomics = 11.4 ms on the roofline workload).  With messages m_k = x[u_k] + dst_term[v]:

        dL/dm_k = R1[v] + R2[v] x[u_k] + [k = argmax] G_max[v] + [k = argmin] G_min[v]      (see include/pna_amd.h)

    so  grad_x[u] = sum_{out-edges (u,v)} R1[v]  +  x[u] * sum_{out-edges} R2[v]  +  (max / min terms):
    pna_segreduce_bwd_rowprep_f32 writes the (V, 2TF) table [R1|R2] (and grad_dst), the two sums are a PULL over the
    TRANSPOSED graph with the forward kernel ("sum", 2T towers), pna_segreduce_bwd_argscatter_f32 adds the max / min
    terms with V*T*F atomics each."""
    from . import functional as PF
    from .graph import Graph
    csr = graph.csr
    dev = x.device
    V, TF = gagg.shape[0], T * F                            # (row_of: ident / amx / amn hold node v's row at row_of[v] -- plan order)
    A, A2 = len(aggs), len(all_aggs)
    has_var = any(a in _STAT_AGGS for a in aggs)
    b = _lib.PnaSegreduceBwdArgs()
    if row_of is not None:
        b.stat_row_of = _lib.dev_ptr(row_of, torch.int32, "stat_row_of")
    if node_of is not None:                                 # (the inverse map: rowprep walks ident's rows in sequence)
        b.stat_node_of, b.stat_rows = _lib.dev_ptr(node_of, torch.int32, "stat_node_of"), node_of.numel()
    b.rowptr = _lib.dev_ptr(csr.rowptr, torch.int32, "rowptr")
    b.col = _lib.dev_ptr(csr.col, torch.int32, "col")
    b.V, b.F = V, F
    if dst_term is not None:
        b.dst_term, b.ld_dst = _lib.dev_ptr(dst_term, torch.float32, "dst_term"), dst_term.stride(0)
    b.n_tower, b.n_aggr, b.tower_stride_in = T, A, F
    for i, name in enumerate(aggs):
        b.aggr[i] = _lib.AGG_CODES[name]
    b.gagg, b.ld_g, b.tower_stride_g = _lib.dev_ptr(gagg, torch.float32, "gagg"), gagg.stride(0), A * F
    if has_var:
        base = ident.data_ptr()
        b.mean = ctypes.c_void_p(base + 4 * all_aggs.index("mean") * F)
        b.stdv = ctypes.c_void_p(base + 4 * all_aggs.index("std") * F)
        if "var" in all_aggs:
            b.var = ctypes.c_void_p(base + 4 * all_aggs.index("var") * F)
        b.ld_stat, b.tower_stride_stat = ident.stride(0), A2 * F
    if amx is not None:
        b.argmax, b.argmin, b.ld_arg = (_lib.dev_ptr(amx, torch.int32, "argmax"), _lib.dev_ptr(amn, torch.int32, "argmin"),
                                        amx.stride(0))
    gd = None
    if need_d:
        gd = torch.empty(V, TF, dtype=torch.float32, device=dev)
        b.grad_dst, b.ld_gd = _lib.dev_ptr(gd, torch.float32, "grad_dst"), gd.stride(0)
    if row_of is not None and not (need_x and amx is not None and has_var and "max" in aggs and "min" in aggs and 4 <= F <= 256
                                   and csr.max_degree < 65535 and x.stride(1) == 1):
        raise RuntimeError("_backward_pull: row_of needs the ranked pull (mean, max, min, std; 4 <= F <= 256; degrees < 65535)")
    ranked = (need_x and amx is not None and has_var and "max" in aggs and "min" in aggs and 4 <= F <= 256 and csr.max_degree < 65535
              and x.stride(1) == 1 and os.environ.get("PNA_AMD_BWD_ARGS", "pull") == "pull")
    if packed_rows is not None and not ranked:
        raise RuntimeError("_backward_pull: packed_rows (in place) needs the ranked pull")
    # round 6: the pull over PER-EDGE rows (pna_segreduce_bwd_pull_args.edge_rows): one tower, no destination term, the four standard
    # aggregators -- PNASimpleLayer's backward.  2 x 4F bytes per out-edge instead of 20F + ranks; bit-identical gradients.
    edge_mode = (ranked and PULL_EDGE_ROWS and T == 1 and dst_term is None and set(aggs) == {"mean", "std", "max", "min"} and not need_d
                 and csr.col.numel() * ((F + 15) // 16 * 16) * 4 < (1 << 36))
    if edge_mode:
        b.stat_node_of, b.stat_rows = None, 0               # (the per-edge pass walks the FORWARD work list: a node's statistics through row_of)
        # rows at a 64-byte aligned pitch (F = 75: 80 floats): a 300-byte row then lies on exactly three 128-byte lines wherever it starts
        # (at the packed pitch of 76 floats: 3.4 on average, and the writer's last sector of a row is shared with the next row's first)
        Fp = (F + 15) // 16 * 16
        E_ = csr.col.numel()
        flat = graph.__dict__.get("_pna_amd_edge_rows")     # (3.2 GB at C3: kept on the graph like the packed pull rows, sized for the widest layer seen)
        if flat is None or flat.numel() < (E_ + V) * Fp or flat.device != dev:
            flat = graph.__dict__["_pna_amd_edge_rows"] = torch.empty(max((E_ + V) * Fp, 1), dtype=torch.float32, device=dev)
        P = flat[:E_ * Fp].view(E_, Fp)
        table = flat[E_ * Fp:(E_ + V) * Fp].view(V, Fp)       # R2: its own (V, Fp) table behind the edge rows
        ranks = None
    elif ranked and (PULL_PACKED or packed_rows is not None):
        # one row [R1 | R2 | G_max | G_min | 16-bit ranks] per node at a 128-byte aligned pitch: an out-edge of the pull reads 12 cache
        # lines at F = 75 instead of ~14.7 for three separate pieces (pna_segreduce_bwd_pull_f32, packed rows).  packed_rows: the
        # caller's buffer, whose first 4 T F columns ARE gagg (aggs = mean, std, max, min): rowprep works in place
        pitch = (5 * TF + 31) // 32 * 32
        packed = torch.empty(V, pitch, dtype=torch.float32, device=dev) if packed_rows is None else packed_rows
        table = packed[:, :2 * TF]
        ranks = packed.view(torch.int16)[:, 8 * TF:10 * TF]
    else:
        table = torch.empty(V, TF * (2 if has_var else 1), dtype=torch.float32, device=dev)
        ranks = None
    if not ranked:                                          # (the ranked pull below runs this pass itself, with the ranks)
        rc = _lib.lib().pna_segreduce_bwd_rowprep_f32(ctypes.byref(b), _lib.dev_ptr(table, torch.float32, "table"), table.stride(0),
                                                      _lib.stream_ptr(dev))
        _lib.check(rc, "pna_segreduce_bwd_rowprep_f32")
    gx = None
    if need_x:
        gT = getattr(graph, "_pna_amd_transposed", None)
        if gT is None or gT.num_nodes != x.shape[0]:
            gT = Graph(csr.row.long(), csr.col.long(), x.shape[0])        # edge (v -> u): pulls row v of the table into u
            graph._pna_amd_transposed = gT
        if ranked:
            # round 3: the max / min terms inside the SAME pull (pna_segreduce_bwd_pull_f32): per out-edge the rank of the edge in its
            # destination's list is compared with the 16-bit ranks of argmax / argmin -- no scattered atomics (3.8 ms at C3)
            tcsr = gT.csr
            rank_t = getattr(gT, "_pna_amd_rank_t", None)
            if rank_t is None:                                # position of every transposed edge in its destination's in-edge list
                rank_t = (tcsr.eid.to(torch.int64) - csr.rowptr.to(torch.int64)[tcsr.col.long()]).to(torch.int32).contiguous()
                gT._pna_amd_rank_t = rank_t
            items = gT.work_items()
            hs = gT.heavy_schedule()
            gx = torch.empty(x.shape[0], TF, dtype=torch.float32, device=dev)
            if hs.n_heavy > 0:
                gx.index_fill_(0, hs.heavy_rows.long(), 0.0)  # hub sources: their segments add atomically
            if ranks is None and not edge_mode:
                ranks = torch.empty(V, 2 * TF, dtype=torch.int16, device=dev)
            b.x, b.ldx = _lib.dev_ptr(x, torch.float32, "x"), x.stride(0)
            b.grad_x, b.ld_gx = _lib.dev_ptr(gx, torch.float32, "grad_x"), gx.stride(0)
            q = _lib.PnaSegreduceBwdPullArgs()
            q.base = ctypes.cast(ctypes.pointer(b), ctypes.c_void_p)
            q.table, q.ld_table = _lib.dev_ptr(table, torch.float32, "table"), table.stride(0)
            q.col_t, q.rank_t = _lib.dev_ptr(tcsr.col, torch.int32, "col_t"), _lib.dev_ptr(rank_t, torch.int32, "rank_t")
            q.items_t, q.n_items_t, q.run_rowprep = _lib.dev_ptr(items, torch.int32, "items_t"), items.shape[0], 1
            if edge_mode:
                pos_t = getattr(gT, "_pna_amd_pos_t", None)       # position of every transposed edge in the forward CSR
                if pos_t is None:
                    pos_t = gT._pna_amd_pos_t = tcsr.eid.to(torch.int32).contiguous()
                fitems = graph.work_items()
                q.edge_rows, q.ld_edge = _lib.dev_ptr(P, torch.float32, "edge_rows"), P.stride(0)
                q.pos_t = _lib.dev_ptr(pos_t, torch.int32, "pos_t")
                q.items, q.n_items = _lib.dev_ptr(fitems, torch.int32, "items"), fitems.shape[0]
            else:
                q.ranks, q.ld_rank = _lib.dev_ptr(ranks, torch.int16, "ranks"), ranks.stride(0)
            rc = _lib.lib().pna_segreduce_bwd_pull_f32(ctypes.byref(q), _lib.stream_ptr(dev))
            _lib.check(rc, "pna_segreduce_bwd_pull_f32")
            if x.shape[1] != TF:
                full = torch.zeros_like(x)
                full[:, :TF] = gx
                gx = full
            return gx, gd
        S = PF.aggregate(gT, table, F, ["sum"], n_tower=table.shape[1] // F)
        gx = (S[:, :TF] + x[:, :TF] * S[:, TF:]) if has_var else S
        gx = gx.contiguous()
        if amx is not None:
            if F < 4 or os.environ.get("PNA_AMD_BWD_DETERMINISTIC", "0") == "1":
                # Narrow towers (the dense variant's multitask nets: F = 2 or 4 per tower, a few thousand nodes) do not qualify for
                # the ranked pull (4 features per lane); the atomic scatter's order varies from run to run, which Adam amplifies over
                # a training trace (tests/test_gpu_train_trace.py: 2.8e-3 in 2 of 31 runs).  Here the max / min terms are summed in a
                # FIXED order: stable sort of (source, feature) keys, one sequential sum per key.
                _argscatter_sorted(gx, csr.col, amx, amn, gagg, aggs, T, F)
            else:
                b.grad_x, b.ld_gx = _lib.dev_ptr(gx, torch.float32, "grad_x"), gx.stride(0)
                rc = _lib.lib().pna_segreduce_bwd_argscatter_f32(ctypes.byref(b), _lib.stream_ptr(dev))
                _lib.check(rc, "pna_segreduce_bwd_argscatter_f32")
        if x.shape[1] != TF:
            full = torch.zeros_like(x)
            full[:, :TF] = gx
            gx = full
    return gx, gd


def M_rows(t):
    return t.shape[0]


DAGG_GROUPED = os.environ.get("PNA_AMD_DAGG_GROUPED", "1") != "0"     # 0: d agg = gy W^T as the three-block bf16 x 3 contraction (rounds 3-5)
PULL_EDGE_ROWS = os.environ.get("PNA_AMD_PULL_EDGE_ROWS", "1") != "0"   # 0: the ranked pull of rounds 3-5 (R1 | R2 | G_max | G_min | ranks per out-edge)
PULL_PACKED = os.environ.get("PNA_AMD_PULL_PACKED", "1") != "0"   # 0: table, aggregate gradient and ranks as three separate rows (round 3)
DW_KERNEL = os.environ.get("PNA_AMD_DW_KERNEL", "1") != "0"    # 0: the library route (slab-batched GEMM) for the weight gradient
DW_GROUPED = os.environ.get("PNA_AMD_DW_GROUPED", "1") != "0"  # 0: per-row scalers (three scaled copies of gy inside the kernel)
DW_MIN_ROWS = int(os.environ.get("PNA_AMD_DW_MIN_ROWS", "4096"))  # below: a molecule batch's product is one small library GEMM


class SimpleLayerPlanFn(torch.autograd.Function):
    """Gather + posttrans contraction of PNASimpleLayer in TRAINING on a large whole graph, in the graph's DEGREE-PLAN row order
    (models/dgl/pna_layer.py:197-206; round 4).  Forward: the arg-tracking gather writes the aggregate and the arg indices in plan
    order (the work list's rows), so the contraction is the grouped one -- ONE combined block W_D per degree tile instead of three
    scaler blocks (a third of the multiply-adds: 0.72 -> ~0.3 ms at C3) -- and scatters y to node order.  Backward: the weight
    gradient reads the aggregate in sequence (pna_posttrans_dw_grouped_f32, a_plan_order), d agg is the three-block contraction in
    node order as before, rowprep finds a node's mean / std / arg indices through the plan's row map (stat_row_of), the pull is
    unchanged.  Same arithmetic per row as the node-order route up to the combined weight's rounding (the inference paths' relation)."""

    @staticmethod
    def forward(ctx, h, weight, bias, layer, graph):
        from . import degree_groups as DG, functional as PF
        from .dgl.pna_layer import _row_scales, _avg_log_value
        plan = DG.plan_of(graph)
        F, N = layer.in_dim, layer.out_dim
        K = 4 * F
        aggs = list(layer.aggregators)
        x = h if h.stride(1) == 1 else h.contiguous()
        csr = graph.csr
        dev = h.device
        agg = torch.empty(plan.rows, DG.agg_pitch(K), dtype=torch.float32, device=dev)[:, :K]
        agg, amx, amn = ops.segreduce(csr.rowptr, csr.col, x, F, aggs, (None,), tower_stride_in=F, out=agg, want_arg=True, arg_rows=plan.rows,
                                      heavy=graph.heavy_schedule(), workspace=graph.workspace, items=plan.items, heavy_out=plan.heavy_out,
                                      tune=dict(generic=2))
        scales = _row_scales(graph, layer.scalers, layer.avg_d, dev)
        w = weight if weight.stride(-1) == 1 else weight.contiguous()
        y = torch.empty(h.shape[0], N, dtype=torch.float32, device=dev)
        if plan.G:
            img, stride = DG.combined_images(w, K, scales, plan)
            ops.posttrans(agg[:plan.NV], K, w, [None], bias, out=y, row_perm=plan.perm, tile_image=plan.tile_image, w_img=img,
                          image_stride=stride, n_out=N)
        if plan.NR:
            rest_scales = plan.rest_scales(tuple(layer.scalers) + (_avg_log_value(layer.avg_d),), scales)
            ops.posttrans(agg[plan.NV:], K, w, rest_scales, bias, out=y, row_perm=plan.perm_rest, n_out=N)
        ctx.graph, ctx.plan, ctx.scales, ctx.aggs, ctx.F, ctx.N = graph, plan, scales, aggs, F, N
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, agg, amx, amn)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, agg, amx, amn = ctx.saved_tensors
        graph, plan, scales, aggs, F, N = ctx.graph, ctx.plan, ctx.scales, ctx.aggs, ctx.F, ctx.N
        K, S = 4 * F, len(scales)
        gy = gy if gy.stride(1) == 1 else gy.contiguous()
        g_w = g_b = g_h = None
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_b:
            res = ops.posttrans_dw_grouped(gy, agg, K, None, scales, plan, want_bias=want_b, a_plan_order=True)
            if res is None:
                raise RuntimeError("SimpleLayerPlanFn: shape outside pna_posttrans_dw_grouped_f32 (the layer checks it before taking this path)")
            g_w, g_b = res
        if ctx.needs_input_grad[0]:
            # d agg = sum_s scale_s (.) (gy W_s), node order, written STRAIGHT into the pull's packed table rows with the aggregator
            # blocks in the order [mean | std | max | min]: rowprep then turns the first two blocks into R1 | R2 in place -- no
            # separate (V, 4F) gradient tensor, no copy of G_max | G_min (rowprep 0.95 -> ~0.65 ms at C3)
            V = gy.shape[0]
            order = plan.__dict__.get("_dagg_order")                                                           # rows of W^T: mean, std, max, min
            if order is None or order.numel() != K or order.device != gy.device:
                order = plan.__dict__["_dagg_order"] = torch.cat([torch.arange(a * F, (a + 1) * F, device=gy.device) for a in (0, 3, 1, 2)])
            wt = torch.cat([weight[:, s * K:(s + 1) * K].t().index_select(0, order) for s in range(S)], dim=1).contiguous()     # (K, S*N)
            pitch = (5 * F + 31) // 32 * 32
            # (1.5 GB at C3, live only inside this call: kept on the plan -- allocated per step it makes the caching allocator split and
            # re-grow its largest block as soon as the training loop frees its gradients every iteration: +1.8 ms per step measured)
            # One flat buffer sized for the widest layer seen on this graph, viewed at this layer's pitch: a net whose layers differ
            # in F does not reallocate it in every layer's backward.  One backward at a time per Graph (the buffer, the
            # plan's weight-gradient workspace and graph.workspace are shared: the one-caller-per-Graph rule of functional.py).
            flat = plan.__dict__.get("_pull_rows")
            if flat is None or flat.numel() < V * pitch or flat.device != gy.device:
                flat = plan.__dict__["_pull_rows"] = torch.empty(V * pitch, dtype=torch.float32, device=gy.device)
            packed = flat[:V * pitch].view(V, pitch)
            if DAGG_GROUPED and plan.G and ops.project_applies(gy, N, K) and plan.NV % 128 == 0:
                # rows of the degree groups: ONE combined 4F x N weight per group (W_D^T = sum_s scale_s(D) W_s^T: a third of the
                # multiply-adds), resident in LDS, exact fp32 products (pna_project_grouped_f32); the three-block contraction took 1.11 ms
                # at C3.  The few rows no group holds: the three-block contraction over their compact list.
                with torch.no_grad():
                    sc = plan.group_scaler_values(scales)                                                  # (G, S)
                    wb = weight.reshape(N, S, K).index_select(2, order)                                     # (N, S, 4F) in the packed order
                    wg = torch.einsum("gs,nsk->gkn", sc, wb).contiguous()                                  # (G, 4F, N) = W_D^T per group
                perm_g, group_g = plan.tiles_by_group()
                g_agg = ops.project_grouped(gy, N, wg, perm_g, group_g, out=packed[:, :K])
                if plan.NR:
                    rest = plan.rest_rows
                    g_rest = ops.posttrans(gy.index_select(0, rest), N, wt, [None if r is None else r.index_select(0, rest) for r in scales], None,
                                           arith="bf16x3")
                    g_agg.index_copy_(0, rest, g_rest)
            else:
                g_agg = ops.posttrans(gy, N, wt, scales, None, arith="bf16x3", out=packed[:, :K])
            g_h, _ = _backward_pull(graph, x, None, agg, amx, amn, g_agg, ["mean", "std", "max", "min"], aggs, 1, F, True, False,
                                    row_of=plan.vmap32(), packed_rows=packed, node_of=plan.node_of_rows())
        return g_h, g_w, g_b, None, None


def simple_layer_plan_applies(layer, graph, h):
    """Whether SimpleLayerPlanFn serves this training forward: a large whole graph with a degree plan, the four standard aggregators,
    three scalers, one-layer posttrans, out_dim <= 80 (the grouped contraction and the weight-gradient kernel)."""
    from . import degree_groups as DG
    from .graph import Graph
    F, N = layer.in_dim, layer.out_dim
    if not (PLAN_TRAIN and h.is_cuda and h.dtype == torch.float32 and type(graph) is Graph and h.shape[1] == F and 4 <= F and N <= 80
            and len(layer.scalers) == 3 and tuple(layer.aggregators) == ("mean", "max", "min", "std") and layer.posttrans.is_affine
            and ops.POSTTRANS_ARITH != "f32" and graph.csr.max_degree < 65535 and 4 * F + 1 <= 384):
        return False
    V = h.shape[0]
    if not DG.applies(graph, V, N, 3, layer.aggregators, F=F, n_edges=graph.csr.col.numel(), x_rows=V):
        return False
    plan = DG.plan_of(graph)
    return plan.NV % 128 == 0 and plan.NRp % 192 == 0 and plan.items is not None


PLAN_TRAIN = os.environ.get("PNA_AMD_PLAN_TRAIN", "1") != "0"   # 0: the node-order training route (AggregateFn + PosttransFn)


class PosttransFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, agg, K, weight, bias, row_scales, h_self, degree_graph=None):
        w = weight if weight.stride(-1) == 1 else weight.contiguous()
        y = ops.posttrans(agg, K, w, row_scales, bias, h_self)
        ctx.K, ctx.row_scales = K, row_scales
        ctx.degree_graph = degree_graph                   # the graph whose DEGREE scalers row_scales are (or None): see backward
        ctx.save_for_backward(agg, weight, h_self)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        agg, weight, h_self = ctx.saved_tensors
        K, scales = ctx.K, ctx.row_scales
        S = len(scales)
        Kh = 0 if h_self is None else h_self.shape[1]
        a = agg[:, :K]
        # G = [scale_0 (.) gy | scale_1 (.) gy | ...]  (M, S*N): one operand for both big products, so that each is ONE
        # library GEMM (three (N x M)(M x K) products with M = 1e6 ran at 20 TF/s; the fused (S*N x M)(M x K) one tiles better)
        N = gy.shape[1]
        _G = []

        def scaled_blocks():                              # (only the library routes below need the scaled copies of gy)
            if not _G:
                if S == 1 and scales[0] is None:
                    _G.append(gy)
                else:                                     # written block by block into place (mul + cat cost a second pass: 0.65 ms at C3)
                    G3 = gy.new_empty(gy.shape[0], S, N)
                    for s, rs in enumerate(scales):
                        if rs is None:
                            G3[:, s].copy_(gy)
                        else:
                            torch.mul(gy, rs.unsqueeze(1), out=G3[:, s])
                    _G.append(G3.view(gy.shape[0], S * N))
            return _G[0]
        g_agg = g_w = g_b = g_h = None
        if ctx.needs_input_grad[0]:
            # d agg = sum_s scale_s (.) (gy W_s) is the forward contraction with the roles of K and N swapped: input gy (M, N),
            # "weight" W_s^T (K, N) per scaler, per-row scalers applied to the accumulators.  On large batches it runs on the
            # same bf16x3 kernel as the forward (fp32-level accuracy) instead of a library fp32 GEMM: 3.0 -> ~0.9 ms at C3
            if gy.is_cuda and S <= 3 and N >= 4 and M_rows(gy) >= ops.X3_MIN_ROWS and ops.POSTTRANS_ARITH in ("auto", "bf16x3"):
                wt = torch.cat([weight[:, Kh + s * K:Kh + (s + 1) * K].t() for s in range(S)], dim=1).contiguous()   # (K, S*N)
                g_agg = ops.posttrans(gy.contiguous(), N, wt, scales, None, arith="bf16x3")
            else:
                g_agg = scaled_blocks() @ torch.cat([weight[:, Kh + s * K:Kh + (s + 1) * K] for s in range(S)], dim=0)
            if agg.shape[1] != K:
                full = torch.zeros_like(agg)
                full[:, :K] = g_agg
                g_agg = full
        want_b = ctx.has_bias and ctx.needs_input_grad[3]
        if ctx.needs_input_grad[2]:
            # round 4: weight (and bias) gradient on the hand-written bf16x3 kernel (pna_posttrans_dw_f32: no scaled copies of gy, no
            # vendor GEMM on the training step; deterministic); shapes outside it keep the slab-batched library route
            res = None
            if DW_KERNEL and gy.is_cuda and M_rows(gy) >= DW_MIN_ROWS:
                dg = ctx.degree_graph
                if dg is not None and DW_GROUPED and dg.num_nodes == M_rows(gy):
                    # the rows in the graph's degree-plan order: 32 consecutive rows share their scaler values (ops.posttrans_dw_grouped).
                    # Only for a graph that HAS a plan or is large enough for the forward to build one (DG.MIN_ROWS): a per-batch graph
                    # of a few thousand nodes -- a fresh Graph every training step -- must not pay the plan's sorts and host syncs inside
                    # every backward and then throw it away; it takes pna_posttrans_dw_f32 below, which needs no plan
                    from . import degree_groups as DG
                    plan = dg.__dict__.get("_pna_amd_degree_plan")
                    if plan is None and dg.num_nodes >= DG.MIN_ROWS:
                        plan = DG.plan_of(dg)
                    if plan is not None and plan.G > 0 and plan.NR <= DG.MAX_REST_FRACTION * dg.num_nodes:
                        res = ops.posttrans_dw_grouped(gy, a, K, h_self, scales, plan, want_bias=want_b)
                if res is None:
                    res = ops.posttrans_dw(gy, a, K, h_self, scales, want_bias=want_b)
            if res is not None:
                g_w, g_b = res
            else:
                gw = _tall_tn(scaled_blocks(), a)                                  # (S*N, K): block s = (scale_s gy)^T a
                parts = ([_tall_tn(gy, h_self)] if Kh else []) + [gw[s * N:(s + 1) * N] for s in range(S)]
                g_w = torch.cat(parts, dim=1)
        if want_b and g_b is None:
            g_b = _column_sums(gy)
        if Kh and ctx.needs_input_grad[5]:
            g_h = gy @ weight[:, :Kh]
        return g_agg, None, g_w, g_b, None, g_h, None


def bn_tail_applies(bn, y, residual):
    """Whether BnTailFn serves nn.BatchNorm1d `bn` in training mode on `y` (+ residual): fp32 rows on a GPU, batch statistics
    with an exponential running average (momentum=None, the cumulative average, and M < 2 -- where torch raises -- stay torch's)."""
    return (BN_TAIL and bn.training and y.is_cuda and y.dtype == torch.float32 and y.dim() == 2 and y.stride(1) == 1
            and 2 <= y.shape[0] and 1 <= y.shape[1] <= 128 and bn.track_running_stats and bn.momentum is not None
            and bn.running_mean is not None
            and (residual is None or (residual.shape == y.shape and residual.dtype == torch.float32 and residual.is_cuda)))


BN_TAIL = os.environ.get("PNA_AMD_BN_TAIL", "1") != "0"


def bn_relu_residual(bn, y, residual=None, relu=True):
    """residual + relu(bn(y)) for nn.BatchNorm1d `bn` in TRAINING mode: updates the running statistics and
    num_batches_tracked like the module's own forward."""
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return BnTailFn.apply(y, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, float(bn.momentum), float(bn.eps), bool(relu))


class BnTailFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, residual, running_mean, running_var, momentum, eps, relu):
        M, N = y.shape
        dev = y.device
        res = None if residual is None else (residual if residual.stride(1) == 1 else residual.contiguous())
        out = torch.empty(M, N, dtype=torch.float32, device=dev)
        stats = torch.empty(2, N, dtype=torch.float32, device=dev)
        ws = torch.empty(_lib.lib().pna_bn_tail_workspace_bytes(M, N) // 4, dtype=torch.float32, device=dev)
        a = _lib.PnaBnTailArgs()
        a.y, a.ldy, a.M, a.N, a.relu = _lib.dev_ptr(y, torch.float32, "y"), y.stride(0), M, N, int(relu)
        if gamma is not None:
            a.gamma, a.beta = _lib.dev_ptr(gamma, torch.float32, "gamma"), _lib.dev_ptr(beta, torch.float32, "beta")
        a.eps, a.momentum = eps, momentum
        a.running_mean, a.running_var = _lib.dev_ptr(running_mean, torch.float32, "running_mean"), _lib.dev_ptr(running_var, torch.float32, "running_var")
        if res is not None:
            a.residual, a.ld_res = _lib.dev_ptr(res, torch.float32, "residual"), res.stride(0)
        a.out, a.ld_out = _lib.dev_ptr(out, torch.float32, "out"), out.stride(0)
        a.save_mean, a.save_invstd = _lib.dev_ptr(stats[0], torch.float32, "save_mean"), _lib.dev_ptr(stats[1], torch.float32, "save_invstd")
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        _lib.check(_lib.lib().pna_bn_tail_fwd_f32(ctypes.byref(a), _lib.stream_ptr(dev)), "pna_bn_tail_fwd_f32")
        ctx.save_for_backward(y, gamma, beta, stats)
        ctx.relu, ctx.has_res, ctx.eps = relu, residual is not None, eps
        return out

    @staticmethod
    def backward(ctx, go):
        y, gamma, beta, stats = ctx.saved_tensors
        M, N = y.shape
        dev = y.device
        go = go if go.stride(1) == 1 else go.contiguous()
        gy = torch.empty(M, N, dtype=torch.float32, device=dev)
        gwb = torch.empty(2, N, dtype=torch.float32, device=dev)
        ws = torch.empty(_lib.lib().pna_bn_tail_workspace_bytes(M, N) // 4, dtype=torch.float32, device=dev)
        a = _lib.PnaBnTailArgs()
        a.y, a.ldy, a.M, a.N, a.relu = _lib.dev_ptr(y, torch.float32, "y"), y.stride(0), M, N, int(ctx.relu)
        if gamma is not None:
            a.gamma, a.beta = _lib.dev_ptr(gamma, torch.float32, "gamma"), _lib.dev_ptr(beta, torch.float32, "beta")
        a.eps, a.momentum = ctx.eps, -1.0
        a.save_mean, a.save_invstd = _lib.dev_ptr(stats[0], torch.float32, "save_mean"), _lib.dev_ptr(stats[1], torch.float32, "save_invstd")
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        a.grad_out, a.ld_go = _lib.dev_ptr(go, torch.float32, "grad_out"), go.stride(0)
        a.grad_y, a.ld_gy = _lib.dev_ptr(gy, torch.float32, "grad_y"), gy.stride(0)
        a.grad_gamma, a.grad_beta = _lib.dev_ptr(gwb[0], torch.float32, "grad_gamma"), _lib.dev_ptr(gwb[1], torch.float32, "grad_beta")
        _lib.check(_lib.lib().pna_bn_tail_bwd_f32(ctypes.byref(a), _lib.stream_ptr(dev)), "pna_bn_tail_bwd_f32")
        has_affine = gamma is not None
        return (gy, gwb[0] if has_affine and ctx.needs_input_grad[1] else None, gwb[1] if has_affine and ctx.needs_input_grad[2] else None,
                go if ctx.has_res else None, None, None, None, None, None)
