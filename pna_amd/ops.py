"""Python entry points of the HIP kernels (thin marshalling over the C ABI, include/pna_amd.h).

`segreduce` = fused gather + multi-aggregator segment-reduce + degree scalers
              (replaces DGL update_all + reduce_func, models/dgl/pna_layer.py:45-50,:64,:189-194,:202)
`posttrans` = post-aggregation tower contraction on the fp32 matrix cores
              (replaces the posttrans nn.Linear, models/dgl/pna_layer.py:65-68,:206)
Both require GPU tensors; there is no CPU path.
"""
import ctypes
import os
from typing import Optional, Sequence

import torch

from . import _lib
from .graph import HeavySchedule

# Arithmetic of the posttrans contraction: "f32" = v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain), "bf16x3" = fp32
# operands cut exactly into three bf16 terms, six partial products on the bf16 matrix pipe (fp32-level accuracy, see
# pna_posttrans_x3.hip), "auto" = bf16x3 where it is implemented (<= 3 scalers) and the problem fills its 256-row
# persistent tiles (>= X3_MIN_ROWS rows), f32 otherwise.
POSTTRANS_ARITH = os.environ.get("PNA_AMD_POSTTRANS", "auto")
X3_MIN_ROWS = 16384

_TUNE = {}   # process-wide tuning overrides (set by tools/sweep.py [removed in round 5: git history] and bench.py), see set_tuning()


def set_tuning(**kw):
    """Override launch tuning of pna_segreduce_fwd_f32 (lanes_per_row, unroll, rows_per_group, vec, nt_store)."""
    _TUNE.clear()
    _TUNE.update({k: int(v) for k, v in kw.items() if v is not None})


def _ld(t):
    return t.stride(0) if t.dim() == 2 else t.shape[-1]


def segreduce(rowptr: torch.Tensor, col: Optional[torch.Tensor], x: torch.Tensor, F: int,
              aggregators: Sequence[str], row_scales: Sequence[Optional[torch.Tensor]] = (None,),
              *, n_tower: int = 1, tower_stride_in: Optional[int] = None, dst_term: Optional[torch.Tensor] = None,
              edge_term: Optional[torch.Tensor] = None, edge_weight: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None, block_stride: Optional[int] = None,
              tower_stride_out: Optional[int] = None, want_arg: bool = False,
              heavy: Optional[HeavySchedule] = None, workspace=None, tune: Optional[dict] = None,
              items: Optional[torch.Tensor] = None, heavy_out: Optional[torch.Tensor] = None, out_row_of: Optional[torch.Tensor] = None,
              edge_type: Optional[torch.Tensor] = None, arg_rows: Optional[int] = None):
    """out[v, t*tso + (s*A + a)*bs + f] = aggregators[a]({m_k}) [f] * row_scales[s][v]   (see pna_amd.h).
    edge_type (int32 [E], CSR order): edge_term then holds one row per edge TYPE (ABI 14).

    rowptr:int32[V+1]; col:int32[E] or None (x edge-resident); x:(rows, >= T*F) fp32.
    Returns out, or (out, argmax, argmin) when want_arg.
    """
    V = rowptr.numel() - 1
    A, S, T = len(aggregators), len(row_scales), max(1, n_tower)
    bs = F if block_stride is None else block_stride
    tsi = F if tower_stride_in is None else tower_stride_in
    tso = A * S * bs if tower_stride_out is None else tower_stride_out
    dev = x.device
    if out is None:
        out = torch.empty(V, (T - 1) * tso + A * S * bs, dtype=torch.float32, device=dev)
    a = _lib.PnaSegreduceArgs()
    a.rowptr = _lib.dev_ptr(rowptr, torch.int32, "rowptr")
    a.col = _lib.dev_ptr(col, torch.int32, "col")
    a.V, a.F = V, F
    a.x, a.ldx, a.x_rows = _lib.dev_ptr(x, torch.float32, "x"), _ld(x), x.shape[0]
    if dst_term is not None:
        a.dst_term, a.ld_dst = _lib.dev_ptr(dst_term, torch.float32, "dst_term"), _ld(dst_term)
    if edge_term is not None:
        a.edge_term, a.ld_edge = _lib.dev_ptr(edge_term, torch.float32, "edge_term"), _ld(edge_term)
        if edge_type is not None:
            a.edge_type, a.n_edge_types = _lib.dev_ptr(edge_type, torch.int32, "edge_type"), edge_term.shape[0]
    if edge_weight is not None:
        a.edge_weight = _lib.dev_ptr(edge_weight, torch.float32, "edge_weight")
    a.n_tower, a.tower_stride_in, a.tower_stride_out = T, tsi, tso
    a.n_aggr = A
    for i, name in enumerate(aggregators):
        if name in ("moment3", "moment4", "moment5"):
            # the reference's DGL moment aggregators return a 0-dim tensor (models/dgl/aggregators.py:33: mean over the WHOLE mailbox), which
            # its reduce_func cannot concatenate (pna_layer.py:48 raises there too): registry entries, not per-node aggregators of a layer
            raise RuntimeError(f"zero-dimensional tensor cannot be concatenated: '{name}' of the DGL registry reduces the whole mailbox to one "
                               f"number (models/dgl/aggregators.py:29-36) and cannot be a layer's aggregator -- the reference fails at the same place")
        a.aggr[i] = _lib.AGG_CODES[name]          # KeyError on unknown names, like the reference's dict lookup
    a.n_scaler = S
    for i, rs in enumerate(row_scales):
        if rs is not None:
            if rs.numel() != V:
                raise ValueError("row scale must have one entry per destination row")
            a.row_scale[i] = _lib.dev_ptr(rs, torch.float32, "row_scale").value
    a.out, a.ldo, a.block_stride = _lib.dev_ptr(out, torch.float32, "out"), _ld(out), bs
    argmax = argmin = None
    if want_arg:
        # (arg_rows: a work list that re-orders the output rows -- the degree plan's -- puts the arg indices in the same rows)
        argmax = torch.empty(V if arg_rows is None else arg_rows, (T - 1) * tsi + F, dtype=torch.int32, device=dev)
        argmin = torch.empty_like(argmax)
        a.argmax, a.argmin, a.ld_arg = (_lib.dev_ptr(argmax, torch.int32, "argmax"),
                                        _lib.dev_ptr(argmin, torch.int32, "argmin"), _ld(argmax))
    keep = None
    if out_row_of is not None:                     # with dst_term: output row of every node (ABI 12)
        a.out_row_of = _lib.dev_ptr(out_row_of, torch.int32, "out_row_of")
    if heavy is not None and heavy.n_heavy > 0:
        a.heavy_threshold, a.seg_len, a.n_heavy, a.n_seg = heavy.threshold, heavy.seg_len, heavy.n_heavy, heavy.n_seg
        a.heavy_rows = _lib.dev_ptr(heavy.heavy_rows, torch.int32, "heavy_rows")
        if heavy_out is not None:                  # output row of every heavy row (re-ordered aggregates: include/pna_amd.h)
            a.heavy_out_rows = _lib.dev_ptr(heavy_out, torch.int32, "heavy_out")
        a.heavy_segptr = _lib.dev_ptr(heavy.heavy_segptr, torch.int32, "heavy_segptr")
        a.seg_heavy = _lib.dev_ptr(heavy.seg_heavy, torch.int32, "seg_heavy")
        nbytes = _lib.lib().pna_segreduce_partials_bytes(heavy.n_seg, F, T)
        keep = workspace(nbytes) if workspace is not None else torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
        a.partials = _lib.dev_ptr(keep, torch.float32, "partials")
    tn = dict(_TUNE)
    if tune:
        tn.update(tune)
    if items is not None and items.numel() > 0:
        a.work_items, a.n_work_items = _lib.dev_ptr(items, torch.int32, "work_items"), items.shape[0]
        a.n_edges = 0 if col is None else col.numel()
    for k, v in tn.items():
        setattr(a.tune, k, int(v))
    rc = _lib.lib().pna_segreduce_fwd_f32(ctypes.byref(a), _lib.stream_ptr(dev))
    _lib.check(rc, "pna_segreduce_fwd_f32")
    if keep is not None and workspace is None:
        keep.record_stream(torch.cuda.current_stream(dev))
    return (out, argmax, argmin) if want_arg else out


def pack_posttrans_weight(weight: torch.Tensor, K: int, n_scaler: int, Kh: int):
    """(w_img, wh_img) tile images of a reference-layout posttrans weight (N, Kh + n_scaler*K).  Cached on the weight
    tensor object per (version, storage address, device): inference packs once, training re-packs after every optimizer
    step, and `module.to(device)` / `param.data = ...` swaps (EMA, SWA) -- which keep the Parameter object and its version
    counter -- re-pack too.  The cache lives ON the tensor object: it is never looked up by address alone."""
    key = (weight._version, weight.data_ptr(), str(weight.device), tuple(weight.shape), weight.stride(0), K, n_scaler, Kh)
    hit = getattr(weight, "_pna_amd_pack", None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    L = _lib.lib()
    N = weight.shape[0]
    nh = ctypes.c_int64(0)
    nw = L.pna_posttrans_packed_floats(K, N, n_scaler, Kh, ctypes.byref(nh))
    w_img = torch.empty(nw, dtype=torch.float32, device=weight.device)
    wh_img = torch.empty(max(nh.value, 1), dtype=torch.float32, device=weight.device) if Kh else None
    rc = L.pna_posttrans_pack_f32(_lib.dev_ptr(weight, torch.float32, "weight"), _ld(weight), N, K, n_scaler, Kh,
                                  _lib.dev_ptr(w_img, torch.float32, "w_img"), _lib.dev_ptr(wh_img, torch.float32, "wh_img"),
                                  _lib.stream_ptr(weight.device))
    _lib.check(rc, "pna_posttrans_pack_f32")
    try:
        weight._pna_amd_pack = (key, w_img, wh_img)
    except AttributeError:
        pass
    return w_img, wh_img


def pack_posttrans_weight_x3(weight: torch.Tensor, K: int, n_scaler: int, Kh: int):
    """(w_img, wh_img) bf16x3 tile images (pna_posttrans_x3_pack_f32); cached like pack_posttrans_weight."""
    key = (weight._version, weight.data_ptr(), str(weight.device), tuple(weight.shape), weight.stride(0), K, n_scaler, Kh)
    hit = getattr(weight, "_pna_amd_pack_x3", None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    L = _lib.lib()
    N = weight.shape[0]
    nh = ctypes.c_int64(0)
    nw = L.pna_posttrans_x3_packed_bytes(K, N, n_scaler, Kh, ctypes.byref(nh))
    w_img = torch.empty(nw // 4, dtype=torch.float32, device=weight.device)
    wh_img = torch.empty(max(nh.value // 4, 1), dtype=torch.float32, device=weight.device) if Kh else None
    rc = L.pna_posttrans_x3_pack_f32(_lib.dev_ptr(weight, torch.float32, "weight"), _ld(weight), N, K, n_scaler, Kh,
                                     _lib.dev_ptr(w_img, torch.float32, "w_img"), _lib.dev_ptr(wh_img, torch.float32, "wh_img"),
                                     _lib.stream_ptr(weight.device))
    _lib.check(rc, "pna_posttrans_x3_pack_f32")
    try:
        weight._pna_amd_pack_x3 = (key, w_img, wh_img)
    except AttributeError:
        pass
    return w_img, wh_img


def posttrans_dw(gy: torch.Tensor, a_mat: torch.Tensor, K: int, h: Optional[torch.Tensor], row_scales: Sequence[Optional[torch.Tensor]],
                 want_bias: bool = True):
    """(grad_w (N, Kh + S*K), grad_b (N) or None) of the posttrans contraction through pna_posttrans_dw_f32 (bf16x3, deterministic),
    or None when the shape is outside the kernel's (the caller keeps its library route): see include/pna_amd.h."""
    M, N, S = gy.shape[0], gy.shape[1], len(row_scales)
    Kh = 0 if h is None else h.shape[1]
    if not gy.is_cuda or gy.dtype != torch.float32 or a_mat.dtype != torch.float32 or M < 1 or not 1 <= S <= _lib.PNA_MAX_SCALER:
        return None
    if row_scales[0] is not None and (Kh or want_bias):     # the h panel and the bias ride on the FIRST copy of gy: it must be unscaled
        return None
    L = _lib.lib()
    nb = L.pna_posttrans_dw_workspace_bytes(M, N, S, K, Kh)
    if nb < 0:
        return None
    gy = gy if gy.stride(1) == 1 else gy.contiguous()
    a_mat = a_mat if a_mat.stride(1) == 1 else a_mat.contiguous()
    if h is not None and h.stride(1) != 1:
        h = h.contiguous()
    dev = gy.device
    gw = torch.empty(N, Kh + S * K, dtype=torch.float32, device=dev)
    gb = torch.empty(N, dtype=torch.float32, device=dev) if want_bias else None
    ws = torch.empty(nb // 4, dtype=torch.float32, device=dev)
    a = _lib.PnaPosttransDwArgs()
    a.gy, a.ldg, a.M, a.N, a.n_scaler = _lib.dev_ptr(gy, torch.float32, "gy"), gy.stride(0), M, N, S
    a.a, a.lda, a.K, a.Kh = _lib.dev_ptr(a_mat, torch.float32, "a"), a_mat.stride(0), K, Kh
    if h is not None:
        a.h, a.ldh = _lib.dev_ptr(h, torch.float32, "h"), h.stride(0)
    keep = []
    for s, rs in enumerate(row_scales):
        if rs is not None:
            rs = rs.reshape(-1)
            rs = rs if rs.is_contiguous() and rs.dtype == torch.float32 else rs.to(torch.float32).contiguous()
            keep.append(rs)
            a.row_scale[s] = _lib.dev_ptr(rs, torch.float32, "row_scale")
    a.grad_w, a.ldw = _lib.dev_ptr(gw, torch.float32, "grad_w"), gw.stride(0)
    if gb is not None:
        a.grad_b = _lib.dev_ptr(gb, torch.float32, "grad_b")
    a.workspace, a.workspace_bytes = ws.data_ptr(), nb
    _lib.check(L.pna_posttrans_dw_f32(ctypes.byref(a), _lib.stream_ptr(dev)), "pna_posttrans_dw_f32")
    return gw, gb


def posttrans_dw_grouped(gy, a_mat, K, h, row_scales, plan, want_bias=True, a_plan_order=False):
    """posttrans_dw over the rows of a degree plan, in plan order (pna_posttrans_dw_grouped_f32: one unscaled copy of gy, a third of
    the multiply-adds, the scalers applied per degree run in the reduction) + the plan's few rest rows through small library
    products.  `row_scales` must be the plan's graph's DEGREE scalers (functions of the in-degree).  None: shape outside the kernel."""
    N, S = gy.shape[1], len(row_scales)
    Kh = 0 if h is None else h.shape[1]
    L = _lib.lib()
    if (not gy.is_cuda or gy.dtype != torch.float32 or plan.G == 0 or not 1 <= S <= 3
            or L.pna_posttrans_dw_grouped_workspace_bytes(N, K, Kh, 1) < 0):
        return None
    dev = gy.device
    n_wg = torch.cuda.get_device_properties(dev).multi_processor_count
    tg, wg_range, wg_entry, entry_group, n_entries = plan.dw_tables(n_wg)
    gy = gy if gy.stride(1) == 1 else gy.contiguous()
    a_mat = a_mat if a_mat.stride(1) == 1 else a_mat.contiguous()
    if h is not None and h.stride(1) != 1:
        h = h.contiguous()
    key = tuple(None if rs is None else (rs.data_ptr(), rs._version) for rs in row_scales)
    hit = plan.__dict__.get("_dw_gscale")
    if hit is None or hit[0] != key:
        gscale = torch.ones(plan.G, S, dtype=torch.float32, device=dev)
        for s, rs in enumerate(row_scales):
            if rs is not None:
                gscale[:, s] = rs.reshape(-1)[plan.group_first_row]
        rest = [None if rs is None else rs.reshape(-1)[plan.rest_rows].unsqueeze(1).contiguous() for rs in row_scales] if plan.NR else None
        hit = plan.__dict__["_dw_gscale"] = (key, gscale.contiguous(), rest)
    gscale, rest_scales = hit[1], hit[2]
    gw = torch.empty(N, Kh + S * K, dtype=torch.float32, device=dev)
    gb = torch.empty(N, dtype=torch.float32, device=dev) if want_bias else None
    nb = L.pna_posttrans_dw_grouped_workspace_bytes(N, K, Kh, n_entries)
    ws = plan.__dict__.get("_dw_ws")                         # (live only inside this call: kept on the plan, like the pull's packed rows)
    if ws is None or ws.numel() * 4 < nb or ws.device != dev:
        ws = plan.__dict__["_dw_ws"] = torch.empty(nb // 4, dtype=torch.float32, device=dev)
    a = _lib.PnaPosttransDwGroupedArgs()
    a.gy, a.ldg, a.N, a.n_scaler = _lib.dev_ptr(gy, torch.float32, "gy"), gy.stride(0), N, S
    a.a, a.lda, a.K, a.Kh = _lib.dev_ptr(a_mat, torch.float32, "a"), a_mat.stride(0), K, Kh
    if h is not None:
        a.h, a.ldh = _lib.dev_ptr(h, torch.float32, "h"), h.stride(0)
    a.row_perm, a.tile_group = _lib.dev_ptr(plan.perm, torch.int32, "row_perm"), _lib.dev_ptr(tg, torch.int32, "tile_group")
    a.wg_range, a.wg_entry = _lib.dev_ptr(wg_range, torch.int32, "wg_range"), _lib.dev_ptr(wg_entry, torch.int32, "wg_entry")
    a.n_workgroups, a.n_entries = n_wg, n_entries
    a.entry_group, a.group_scale = _lib.dev_ptr(entry_group, torch.int32, "entry_group"), _lib.dev_ptr(gscale, torch.float32, "group_scale")
    a.grad_w, a.ldw = _lib.dev_ptr(gw, torch.float32, "grad_w"), gw.stride(0)
    if gb is not None:
        a.grad_b = _lib.dev_ptr(gb, torch.float32, "grad_b")
    a.workspace, a.workspace_bytes = ws.data_ptr(), nb
    a.a_plan_order = 1 if a_plan_order else 0              # (a: (plan.rows, >= K) in the plan's row order -- a forward that gathered in it)
    _lib.check(L.pna_posttrans_dw_grouped_f32(ctypes.byref(a), _lib.stream_ptr(dev)), "pna_posttrans_dw_grouped_f32")
    if plan.NR:                                            # hub rows and rare degrees: a few thousand rows, plain products
        rr = plan.rest_rows
        g_r = gy.index_select(0, rr)
        a_r = a_mat[plan.NV:plan.NV + plan.NR, :K] if a_plan_order else a_mat.index_select(0, rr)[:, :K]
        parts = [g_r.t() @ h.index_select(0, rr)] if Kh else []
        for rs in rest_scales:
            parts.append((g_r if rs is None else g_r * rs).t() @ a_r)
        gw += torch.cat(parts, dim=1)
        if gb is not None:
            gb += g_r.sum(0)
    return gw, gb


def posttrans(a_mat: torch.Tensor, K: int, weight: torch.Tensor, row_scales: Sequence[Optional[torch.Tensor]],
              bias: Optional[torch.Tensor] = None, h: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
              row_post: Optional[torch.Tensor] = None, col_scale: Optional[torch.Tensor] = None,
              col_shift: Optional[torch.Tensor] = None, relu: bool = False, residual: Optional[torch.Tensor] = None,
              arith: Optional[str] = None, pipeline: int = 0, leaky_slope: Optional[float] = None,
              row_perm: Optional[torch.Tensor] = None, tile_image: Optional[torch.Tensor] = None, w_img=None, image_stride: int = 0,
              n_out: Optional[int] = None):
    """y = residual + act(((bias + h@Wh^T + sum_s row_scales[s][:,None] * (a[:, :K] @ W_s^T)) * row_post[:,None]) *
    col_scale + col_shift)                                                                      (see pna_amd.h).

    a_mat:(M, >=K) fp32; weight:(N, Kh + S*K) in the reference nn.Linear layout (columns [h | scaler blocks]).
    """
    M, S = a_mat.shape[0], len(row_scales)
    N = weight.shape[0] if n_out is None else n_out
    Kh = 0 if h is None else h.shape[1]
    if row_perm is not None:
        return _posttrans_grouped(a_mat, K, weight, row_scales, bias, out, row_post, col_scale, col_shift, relu, residual, leaky_slope,
                                  row_perm, tile_image, w_img, image_stride, N)
    if weight.shape[1] != Kh + S * K:
        raise ValueError(f"weight has {weight.shape[1]} input columns, expected Kh + n_scaler*K = {Kh + S * K}")
    dev = a_mat.device
    explicit = arith is not None
    arith = arith or POSTTRANS_ARITH
    if arith not in ("f32", "bf16x3", "auto"):
        raise ValueError(f"unknown posttrans arithmetic {arith!r} (f32 | bf16x3 | auto)")
    if arith == "bf16x3" and S > 3:
        if explicit:
            raise ValueError("the bf16x3 posttrans kernel supports at most 3 scalers")
        arith = "f32"                              # the process-wide preference falls back where bf16x3 is not implemented
    x3 = arith == "bf16x3" or (arith == "auto" and S <= 3 and M >= X3_MIN_ROWS)
    w_img, wh_img = (pack_posttrans_weight_x3 if x3 else pack_posttrans_weight)(weight, K, S, Kh)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=dev)
    g = _lib.PnaPosttransArgs()
    g.a, g.lda, g.M, g.K, g.N, g.n_scaler = _lib.dev_ptr(a_mat, torch.float32, "a"), _ld(a_mat), M, K, N, S
    for i, rs in enumerate(row_scales):
        if rs is not None:
            g.row_scale[i] = _lib.dev_ptr(rs, torch.float32, "row_scale").value
    g.w_img = _lib.dev_ptr(w_img, torch.float32, "w_img")
    if h is not None:
        g.h, g.ldh, g.Kh = _lib.dev_ptr(h, torch.float32, "h"), _ld(h), Kh
        g.wh_img = _lib.dev_ptr(wh_img, torch.float32, "wh_img")
    g.bias = _lib.dev_ptr(bias, torch.float32, "bias")
    if row_post is not None:
        if row_post.numel() != M:
            raise ValueError("row_post must have one entry per row")
        g.row_post = _lib.dev_ptr(row_post, torch.float32, "row_post")
    g.col_scale = _lib.dev_ptr(col_scale, torch.float32, "col_scale")
    g.col_shift = _lib.dev_ptr(col_shift, torch.float32, "col_shift")
    g.relu = 2 if leaky_slope is not None else (1 if relu else 0)      # leaky_slope: LeakyReLU(v) = v < 0 ? slope * v : v
    g.act_slope = float(leaky_slope) if leaky_slope is not None else 0.0
    if residual is not None:
        g.residual, g.ld_res = _lib.dev_ptr(residual, torch.float32, "residual"), _ld(residual)
    g.y, g.ldy = _lib.dev_ptr(out, torch.float32, "y"), _ld(out)
    g.pipeline = int(pipeline) if x3 else 0      # bf16x3 only: 0 = default, 2 / 3 = weight buffers in LDS (include/pna_amd.h)
    fn = "pna_posttrans_x3_f32" if x3 else "pna_posttrans_f32"
    rc = getattr(_lib.lib(), fn)(ctypes.byref(g), _lib.stream_ptr(dev))
    _lib.check(rc, fn)
    return out


def _posttrans_grouped(a_mat, K, weight, row_scales, bias, out, row_post, col_scale, col_shift, relu, residual, leaky_slope,
                       row_perm, tile_image, w_img, image_stride, N):
    """pna_posttrans_x3_f32 with rows in a virtual order (include/pna_amd.h, pna_posttrans_args.row_perm): `a_mat`, `row_scales`,
    `row_post` are indexed by virtual row, `out` / `residual` by row_perm[virtual row].  `w_img`: a packed bf16x3 image buffer
    (tile_image given: image tile_image[t] for tile t, image_stride bytes apart) or None (pack `weight` as usual)."""
    M, S = a_mat.shape[0], len(row_scales)
    if out is None:
        raise ValueError("grouped posttrans writes into a caller-provided `out` (rows in real order)")
    if w_img is None:
        w_img, _ = pack_posttrans_weight_x3(weight, K, S, 0)
    g = _lib.PnaPosttransArgs()
    g.a, g.lda, g.M, g.K, g.N, g.n_scaler = _lib.dev_ptr(a_mat, torch.float32, "a"), _ld(a_mat), M, K, N, S
    for i, rs in enumerate(row_scales):
        if rs is not None:
            g.row_scale[i] = _lib.dev_ptr(rs, torch.float32, "row_scale").value
    g.w_img = _lib.dev_ptr(w_img, torch.float32, "w_img")
    g.bias = _lib.dev_ptr(bias, torch.float32, "bias")
    if row_post is not None:
        g.row_post = _lib.dev_ptr(row_post, torch.float32, "row_post")
    g.col_scale = _lib.dev_ptr(col_scale, torch.float32, "col_scale")
    g.col_shift = _lib.dev_ptr(col_shift, torch.float32, "col_shift")
    g.relu = 2 if leaky_slope is not None else (1 if relu else 0)
    g.act_slope = float(leaky_slope) if leaky_slope is not None else 0.0
    if residual is not None:
        g.residual, g.ld_res = _lib.dev_ptr(residual, torch.float32, "residual"), _ld(residual)
    g.y, g.ldy = _lib.dev_ptr(out, torch.float32, "y"), _ld(out)
    g.row_perm = _lib.dev_ptr(row_perm, torch.int32, "row_perm")
    if tile_image is not None:
        g.tile_image, g.image_stride = _lib.dev_ptr(tile_image, torch.int32, "tile_image"), int(image_stride)
    rc = _lib.lib().pna_posttrans_x3_f32(ctypes.byref(g), _lib.stream_ptr(a_mat.device))
    _lib.check(rc, "pna_posttrans_x3_f32 (grouped)")
    return out


def pack_fused_weight(weight: torch.Tensor, F: int, n_scaler: int):
    """Packed image of a PNASimpleLayer posttrans weight (N, S*4*F) for pna_fused_simple_f32: every aggregator block
    zero-padded from F to round_up(F, 4) input columns.  Cached on the weight object per version."""
    key = (weight._version, weight.data_ptr(), str(weight.device), tuple(weight.shape), F, n_scaler)
    hit = getattr(weight, "_pna_amd_fpack", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    N, B4 = weight.shape[0], (F + 3) // 4 * 4
    wp = torch.nn.functional.pad(weight.detach().reshape(N, n_scaler * 4, F), (0, B4 - F)).reshape(N, n_scaler * 4 * B4)
    wp = wp.contiguous()
    L = _lib.lib()
    nh = ctypes.c_int64(0)
    nw = L.pna_posttrans_packed_floats(4 * B4, N, n_scaler, 0, ctypes.byref(nh))
    w_img = torch.empty(nw, dtype=torch.float32, device=weight.device)
    rc = L.pna_posttrans_pack_f32(_lib.dev_ptr(wp, torch.float32, "weight"), _ld(wp), N, 4 * B4, n_scaler, 0,
                                  _lib.dev_ptr(w_img, torch.float32, "w_img"), None, _lib.stream_ptr(weight.device))
    _lib.check(rc, "pna_posttrans_pack_f32")
    try:
        weight._pna_amd_fpack = (key, w_img)
    except AttributeError:
        pass
    return w_img


def fused_simple(rowptr: torch.Tensor, col: torch.Tensor, x: torch.Tensor, F: int, weight: torch.Tensor,
                 row_scales: Sequence[Optional[torch.Tensor]], bias: Optional[torch.Tensor] = None,
                 col_scale: Optional[torch.Tensor] = None, col_shift: Optional[torch.Tensor] = None, relu: bool = False,
                 residual: Optional[torch.Tensor] = None, heavy_threshold: int = 0, out: Optional[torch.Tensor] = None):
    """PNASimpleLayer forward ("mean max min std") in one launch (pna_fused_simple_f32, see pna_amd.h)."""
    V, S, N = rowptr.numel() - 1, len(row_scales), weight.shape[0]
    if weight.shape[1] != S * 4 * F:
        raise ValueError(f"weight has {weight.shape[1]} input columns, expected n_scaler*4*F = {S * 4 * F}")
    dev = x.device
    w_img = pack_fused_weight(weight, F, S)
    if out is None:
        out = torch.empty(V, N, dtype=torch.float32, device=dev)
    g = _lib.PnaFusedSimpleArgs()
    g.rowptr, g.col = _lib.dev_ptr(rowptr, torch.int32, "rowptr"), _lib.dev_ptr(col, torch.int32, "col")
    g.x, g.ldx = _lib.dev_ptr(x, torch.float32, "x"), _ld(x)
    g.V, g.F, g.N, g.n_scaler = V, F, N, S
    for i, rs in enumerate(row_scales):
        if rs is not None:
            if rs.numel() != V:
                raise ValueError("row scale must have one entry per destination row")
            g.row_scale[i] = _lib.dev_ptr(rs, torch.float32, "row_scale").value
    g.w_img = _lib.dev_ptr(w_img, torch.float32, "w_img")
    g.bias = _lib.dev_ptr(bias, torch.float32, "bias")
    g.col_scale = _lib.dev_ptr(col_scale, torch.float32, "col_scale")
    g.col_shift = _lib.dev_ptr(col_shift, torch.float32, "col_shift")
    if residual is not None:
        g.residual, g.ld_res = _lib.dev_ptr(residual, torch.float32, "residual"), _ld(residual)
    g.y, g.ldy = _lib.dev_ptr(out, torch.float32, "y"), _ld(out)
    g.relu, g.heavy_threshold = (1 if relu else 0), heavy_threshold
    rc = _lib.lib().pna_fused_simple_f32(ctypes.byref(g), _lib.stream_ptr(dev))
    _lib.check(rc, "pna_fused_simple_f32")
    return out


def pack_rows(x: torch.Tensor, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i] = x[idx[i]] for fp32 rows of any 4-byte aligned pitch (pna_pack_rows_f32): the send-side packing of the halo
    all-to-all (pna_amd/shard.py).  idx: int32 [n]."""
    n, F = idx.numel(), x.shape[1]
    if out is None:
        out = torch.empty(n, F, dtype=torch.float32, device=x.device)
    rc = _lib.lib().pna_pack_rows_f32(_lib.dev_ptr(x, torch.float32, "x"), _ld(x), _lib.dev_ptr(idx, torch.int32, "idx"), n, F,
                                      _lib.dev_ptr(out, torch.float32, "out"), _ld(out), _lib.stream_ptr(x.device))
    _lib.check(rc, "pna_pack_rows_f32")
    return out


def project_applies(x: torch.Tensor, K: int, N: int) -> bool:
    """pna_project_f32's domain: 4 <= K <= 128, N <= 512, the weight image (16 ceil(K / 16) x (80 ceil(N / 80) + 4) floats) within 160 KB of LDS."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and 4 <= K <= 128 and 1 <= N <= 512
            and (K + 15) // 16 * 16 * ((N + 79) // 80 * 80 + 4) * 4 <= 160 * 1024)


def project(x: torch.Tensor, K: int, weight: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = x[:, :K] weight^T with weight (N <= 512, K <= 128) in nn.Linear's layout (pna_project_f32): exact fp32 products, the weight
    resident in LDS, x read once and out written once -- the node-level source projection of all the towers of a PNALayer at once."""
    N = weight.shape[0]
    if weight.shape[1] != K or weight.stride(1) != 1:
        raise ValueError("project: weight must be (N, K) with unit inner stride")
    if out is None:
        out = torch.empty(x.shape[0], N, dtype=torch.float32, device=x.device)
    rc = _lib.lib().pna_project_f32(_lib.dev_ptr(x, torch.float32, "x"), _ld(x), x.shape[0], K, _lib.dev_ptr(weight, torch.float32, "weight"),
                                    weight.stride(0), N, _lib.dev_ptr(out, torch.float32, "out"), _ld(out), _lib.stream_ptr(x.device))
    _lib.check(rc, "pna_project_f32")
    return out


def project_scaled_applies(x: torch.Tensor, K: int, N: int, blocks: int) -> bool:
    """pna_project_scaled_f32's domain: N <= 80, 1..4 blocks (four: K <= 80), the image of all blocks within 160 KB of LDS."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and 4 <= K <= 128 and 1 <= N <= 80 and 1 <= blocks <= 4
            and (blocks < 4 or K <= 80) and ((K + 15) // 16 * 16 * (80 * blocks + 4) + 240) * 4 <= 160 * 1024)


def project_scaled(x: torch.Tensor, K: int, weight: torch.Tensor, scales: Optional[torch.Tensor], beta: Optional[torch.Tensor], self_block: bool,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = [x W_0^T] + sum_s scales[:, s] (x W_s^T + beta[s]) with weight (N <= 80, blocks x K) = [W_self | W_0 | ..] side by side
    (pna_project_scaled_f32): exact fp32 products, every block resident in LDS, x read once."""
    N = weight.shape[0]
    S = 0 if scales is None else scales.shape[1]
    if weight.shape[1] != (S + int(self_block)) * K or weight.stride(1) != 1 or (scales is not None and scales.stride(1) != 1):
        raise ValueError("project_scaled: weight must be (N, blocks x K), scales (M, S), both with unit inner stride")
    if out is None:
        out = torch.empty(x.shape[0], N, dtype=torch.float32, device=x.device)
    rc = _lib.lib().pna_project_scaled_f32(
        _lib.dev_ptr(x, torch.float32, "x"), _ld(x), x.shape[0], K, _lib.dev_ptr(weight, torch.float32, "weight"), weight.stride(0), N, S, int(self_block),
        None if scales is None else _lib.dev_ptr(scales, torch.float32, "scales"), 0 if scales is None else scales.stride(0),
        None if beta is None else _lib.dev_ptr(beta, torch.float32, "beta"), 0 if beta is None else beta.stride(0),
        _lib.dev_ptr(out, torch.float32, "out"), _ld(out), _lib.stream_ptr(x.device))
    _lib.check(rc, "pna_project_scaled_f32")
    return out


def project_grouped(x: torch.Tensor, K: int, w_groups: torch.Tensor, row_perm: torch.Tensor, tile_group: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[node] = x[node, :K] w_groups[g]^T for node = row_perm[v], g = tile_group[v // 128] (pna_project_grouped_f32): w_groups (G, N, K)
    contiguous, one matrix per degree group of a degree plan; rows of `out` that no virtual row names are left alone.  List the tiles sorted
    by group (DegreePlan.tiles_by_group): a workgroup refills its weight image only when the group changes."""
    G, N = w_groups.shape[0], w_groups.shape[1]
    if w_groups.shape[2] != K or not w_groups.is_contiguous():
        raise ValueError("project_grouped: w_groups must be a contiguous (G, N, K) tensor")
    rc = _lib.lib().pna_project_grouped_f32(
        _lib.dev_ptr(x, torch.float32, "x"), _ld(x), x.shape[0], K, _lib.dev_ptr(w_groups, torch.float32, "w_groups"), N * K, G, N,
        _lib.dev_ptr(row_perm, torch.int32, "row_perm"), row_perm.numel(), _lib.dev_ptr(tile_group, torch.int32, "tile_group"),
        _lib.dev_ptr(out, torch.float32, "out"), _ld(out), _lib.stream_ptr(x.device))
    _lib.check(rc, "pna_project_grouped_f32")
    return out


def posttrans_towers(agg: torch.Tensor, K: int, weights: Sequence[torch.Tensor], row_scales: Sequence[Optional[torch.Tensor]],
                     biases: Optional[torch.Tensor], h: Optional[torch.Tensor], h_shared: bool, out: torch.Tensor,
                     row_post: Optional[torch.Tensor] = None, col_scale: Optional[torch.Tensor] = None,
                     col_shift: Optional[torch.Tensor] = None, relu: bool = False, arith: Optional[str] = None):
    """The posttrans contraction of ALL towers of a layer in one call (one launch on the exact-f32 kernel, which is what
    small batches use): tower t maps agg[:, t*K:(t+1)*K] (and h[:, t*Kh:(t+1)*Kh], or the whole h when h_shared) through
    weights[t] (N, Kh + S*K) into out[:, t*N:(t+1)*N].  biases / col_scale / col_shift: (T, N) contiguous or None.
    models/dgl/pna_layer.py:133-139 loops over the towers in Python."""
    T, S, N = len(weights), len(row_scales), weights[0].shape[0]
    M = agg.shape[0]
    Kh = 0 if h is None else (h.shape[1] if h_shared else h.shape[1] // T)
    arith = arith or POSTTRANS_ARITH
    if arith == "bf16x3" and S > 3:
        arith = "f32"
    x3 = arith == "bf16x3" or (arith == "auto" and S <= 3 and M >= X3_MIN_ROWS)
    # one buffer holding the T packed images back to back, cached on the first weight per (version, address) of all of them
    key = tuple((w._version, w.data_ptr(), str(w.device)) for w in weights) + (K, S, Kh, x3)
    hit = getattr(weights[0], "_pna_amd_tower_pack", None)
    if hit is None or hit[0] != key:
        imgs = [(pack_posttrans_weight_x3 if x3 else pack_posttrans_weight)(w if w.stride(-1) == 1 else w.contiguous(), K, S, Kh)
                for w in weights]
        w_all = torch.cat([i[0].reshape(-1) for i in imgs])
        wh_all = torch.cat([i[1].reshape(-1) for i in imgs]) if Kh else None
        hit = (key, w_all, wh_all, imgs[0][0].numel(), imgs[0][1].numel() if Kh else 0)
        try:
            weights[0]._pna_amd_tower_pack = hit
        except AttributeError:
            pass
    _, w_all, wh_all, w_stride, wh_stride = hit
    g = _lib.PnaPosttransArgs()
    g.a, g.lda, g.M, g.K, g.N, g.n_scaler = _lib.dev_ptr(agg, torch.float32, "a"), _ld(agg), M, K, N, S
    for i, rs in enumerate(row_scales):
        if rs is not None:
            g.row_scale[i] = _lib.dev_ptr(rs, torch.float32, "row_scale").value
    g.w_img = _lib.dev_ptr(w_all, torch.float32, "w_img")
    if h is not None:
        g.h, g.ldh, g.Kh = _lib.dev_ptr(h, torch.float32, "h"), _ld(h), Kh
        g.wh_img = _lib.dev_ptr(wh_all, torch.float32, "wh_img")
    g.bias = _lib.dev_ptr(biases, torch.float32, "bias")
    if row_post is not None:
        g.row_post = _lib.dev_ptr(row_post, torch.float32, "row_post")
    g.col_scale = _lib.dev_ptr(col_scale, torch.float32, "col_scale")
    g.col_shift = _lib.dev_ptr(col_shift, torch.float32, "col_shift")
    g.relu = 1 if relu else 0
    g.y, g.ldy = _lib.dev_ptr(out, torch.float32, "y"), _ld(out)
    g.n_tower = T
    g.tower_stride_a, g.tower_stride_h, g.tower_stride_y = K, (0 if h_shared else Kh), N
    g.tower_stride_w, g.tower_stride_wh = (w_stride * 4, wh_stride * 4) if x3 else (w_stride, wh_stride)   # x3 images: bytes
    fn = "pna_posttrans_x3_f32" if x3 else "pna_posttrans_f32"
    rc = getattr(_lib.lib(), fn)(ctypes.byref(g), _lib.stream_ptr(agg.device))
    _lib.check(rc, fn)
    return out


# ---- the molecule-batch tower layer (pna_tower_fused.hip): one C call, two launches -----------------------------------------
def pack_small(weight: torch.Tensor) -> torch.Tensor:
    """MFMA-fragment image of an nn.Linear weight (N, K) for small_linear / tower_layer (pna_small_pack_f32).  Not cached here:
    callers cache the bundle of images a layer needs (functional.tower_layer_bundle)."""
    w = weight.detach()
    w = w if w.stride(-1) == 1 else w.contiguous()
    N, K = w.shape
    L = _lib.lib()
    img = torch.empty(L.pna_small_packed_floats(N, K), dtype=torch.float32, device=w.device)
    rc = L.pna_small_pack_f32(_lib.dev_ptr(w, torch.float32, "weight"), _ld(w), N, K, _lib.dev_ptr(img, torch.float32, "img"),
                              _lib.stream_ptr(w.device))
    _lib.check(rc, "pna_small_pack_f32")
    return img


def pack_tower_post(weights: Sequence[torch.Tensor], Fi: int, Fo: int, n_scaler: int) -> torch.Tensor:
    """The T posttrans weights (Fo, Fi + n_scaler*4*Fi) of a layer's towers as one buffer of pna_tower_post_pack_f32 images."""
    L = _lib.lib()
    per = L.pna_tower_post_packed_floats(Fi, Fo, n_scaler)
    dev = weights[0].device
    img = torch.empty(per * len(weights), dtype=torch.float32, device=dev)
    for t, w in enumerate(weights):
        w = w.detach()
        w = w if w.stride(-1) == 1 else w.contiguous()
        if tuple(w.shape) != (Fo, Fi * (1 + 4 * n_scaler)):
            raise ValueError(f"posttrans weight of tower {t} is {tuple(w.shape)}, expected {(Fo, Fi * (1 + 4 * n_scaler))}")
        rc = L.pna_tower_post_pack_f32(_lib.dev_ptr(w, torch.float32, "weight"), _ld(w), Fi, Fo, n_scaler,
                                       ctypes.c_void_p(img.data_ptr() + 4 * per * t), _lib.stream_ptr(dev))
        _lib.check(rc, "pna_tower_post_pack_f32")
    return img


def small_linear(x: torch.Tensor, img: torch.Tensor, N: int, bias: Optional[torch.Tensor] = None, *, act: int = 0, slope: float = 0.0,
                 residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = residual + act(x W^T + bias) with W given as its pack_small image (pna_small_linear_f32): one workgroup per 16 rows."""
    M, K = x.shape
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    g = _lib.PnaSmallLinearArgs()
    g.x, g.ldx, g.M, g.K, g.N, g.act = _lib.dev_ptr(x, torch.float32, "x"), _ld(x), M, K, N, act
    g.img, g.bias, g.act_slope = _lib.dev_ptr(img, torch.float32, "img"), _lib.dev_ptr(bias, torch.float32, "bias"), slope
    if residual is not None:
        g.residual, g.ld_res = _lib.dev_ptr(residual, torch.float32, "residual"), _ld(residual)
    g.y, g.ldy = _lib.dev_ptr(out, torch.float32, "y"), _ld(out)
    rc = _lib.lib().pna_small_linear_f32(ctypes.byref(g), _lib.stream_ptr(x.device))
    _lib.check(rc, "pna_small_linear_f32")
    return out


def tower_layer(rowptr: torch.Tensor, col: torch.Tensor, h: torch.Tensor, *, n_tower: int, Fi: int, Fo: int, divide_input: bool,
                proj_img: torch.Tensor, proj_bias: Optional[torch.Tensor], row_scales: Sequence[Optional[torch.Tensor]],
                post_img: torch.Tensor, post_bias: Optional[torch.Tensor], row_post: Optional[torch.Tensor] = None,
                col_scale: Optional[torch.Tensor] = None, col_shift: Optional[torch.Tensor] = None,
                mix_img: Optional[torch.Tensor] = None, mix_bias: Optional[torch.Tensor] = None, n_out: int = 0, mix_act: int = 0,
                mix_slope: float = 0.0, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                x_cat: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pna_tower_layer_f32 (include/pna_amd.h): projection launch + one launch for gather, tower contractions, graph-norm / eval
    BatchNorm and the mixing network of models/dgl/pna_layer.py:133-148, per 16 destination rows."""
    V = rowptr.numel() - 1
    dev = h.device
    width = n_out if mix_img is not None else n_tower * Fo
    if out is None:
        out = torch.empty(V, width, dtype=torch.float32, device=dev)
    if x_cat is None:
        x_cat = torch.empty(V, 2 * n_tower * Fi, dtype=torch.float32, device=dev)
    g = _lib.PnaTowerLayerArgs()
    g.rowptr, g.col, g.V = _lib.dev_ptr(rowptr, torch.int32, "rowptr"), _lib.dev_ptr(col, torch.int32, "col"), V
    g.n_tower, g.Fi, g.Fo, g.divide_input, g.n_scaler = n_tower, Fi, Fo, 1 if divide_input else 0, len(row_scales)
    g.h, g.ldh = _lib.dev_ptr(h, torch.float32, "h"), _ld(h)
    g.x_cat, g.ldx = _lib.dev_ptr(x_cat, torch.float32, "x_cat"), _ld(x_cat)
    g.proj_img, g.proj_bias = _lib.dev_ptr(proj_img, torch.float32, "proj_img"), _lib.dev_ptr(proj_bias, torch.float32, "proj_bias")
    for i, rs in enumerate(row_scales):
        if rs is not None:
            g.row_scale[i] = _lib.dev_ptr(rs, torch.float32, "row_scale").value
    g.post_img, g.post_bias = _lib.dev_ptr(post_img, torch.float32, "post_img"), _lib.dev_ptr(post_bias, torch.float32, "post_bias")
    g.row_post = _lib.dev_ptr(row_post, torch.float32, "row_post")
    g.col_scale, g.col_shift = _lib.dev_ptr(col_scale, torch.float32, "col_scale"), _lib.dev_ptr(col_shift, torch.float32, "col_shift")
    if mix_img is not None:
        g.mix_img, g.mix_bias = _lib.dev_ptr(mix_img, torch.float32, "mix_img"), _lib.dev_ptr(mix_bias, torch.float32, "mix_bias")
        g.No, g.mix_act, g.mix_slope = n_out, mix_act, mix_slope
    if residual is not None:
        g.residual, g.ld_res = _lib.dev_ptr(residual, torch.float32, "residual"), _ld(residual)
    g.y, g.ldy = _lib.dev_ptr(out, torch.float32, "y"), _ld(out)
    rc = _lib.lib().pna_tower_layer_f32(ctypes.byref(g), _lib.stream_ptr(dev))
    _lib.check(rc, "pna_tower_layer_f32")
    return out
