#!/usr/bin/env python
"""Drives the EXPERIMENTAL one-kernel layer (tools/ubench/degree_fused.hip: gather + degree-grouped contraction, DESIGN.md 4.7
point 7) against the shipped two-kernel degree-grouped layer and the ordinary layer; optional timing on the C3 graph.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -Iinclude -Ipna_amd/csrc \
          tools/ubench/degree_fused.hip -o tools/ubench/libdegree_fused.so
    [DF_WGS=1|2] [DF_DEBUG_AGG=1] python tools/df_check.py [time]

DF_WGS: workgroups per CU (default 2: exact, 1.257 ms on C3 against 1.245 for the shipped path; 1: 1.84 ms).  DF_LIB: another
build of the experiment -- -DDF_PACKED_FOLD (the fold as plain C++, which hipcc packs into v_pk_* ops): 1.231 ms and whole 16-row
tiles wrong at the 1e-5 level, differently every run, unless DF_WGS=1; -DDF_WAVES=8: one 128-row workgroup per CU.  DF_DEBUG_AGG=1: the kernel dumps the statistics its contraction sees; compared bit for bit
with the production aggregate.  Nothing here is part of the product: the group rows go through the experimental kernel, the
rest rows through the shipped gather + grouped contraction over their own work list.
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, _lib, degree_groups as DG, functional as PF, ops  # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer, _row_scales  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

dev = torch.device("cuda:0")
WGS = int(os.environ.get("DF_WGS", "0"))
X = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", os.environ.get("DF_LIB", "libdegree_fused.so")))


class Args(ctypes.Structure):
    _fields_ = [("rowptr", ctypes.c_void_p), ("col", ctypes.c_void_p), ("x", ctypes.c_void_p), ("ldx", ctypes.c_int64),
                ("F", ctypes.c_int32), ("N", ctypes.c_int32), ("row_perm", ctypes.c_void_p), ("M", ctypes.c_int64),
                ("tile_image", ctypes.c_void_p), ("tile_rows", ctypes.c_int32), ("relu", ctypes.c_int32),
                ("w_img", ctypes.c_void_p), ("image_stride", ctypes.c_int64), ("bias", ctypes.c_void_p),
                ("col_scale", ctypes.c_void_p), ("col_shift", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("ld_res", ctypes.c_int64),
                ("y", ctypes.c_void_p), ("ldy", ctypes.c_int64), ("act_slope", ctypes.c_float), ("workgroups_per_cu", ctypes.c_int32)]


X.degree_fused_f32.argtypes = [ctypes.POINTER(Args), ctypes.c_void_p]
X.degree_fused_f32.restype = ctypes.c_int
X.degree_fused_debug_agg.argtypes = [ctypes.c_void_p, ctypes.c_int64]


def fused_images(weight, F, row_scales, plan):
    """W_D per degree group (as degree_groups.combined_images) with K reordered into chunks of 32: chunk 4 fb + a = features
    [32 fb, 32 fb + 32) of aggregator a, zero beyond F -- the order in which a lane of the fused kernel holds its statistics."""
    K, N, G, nfb = 4 * F, weight.shape[0], plan.G, (F + 31) // 32
    with torch.no_grad():
        wc = None
        for s, rs in enumerate(row_scales):
            ws = weight[:, s * K:(s + 1) * K]
            term = ws.unsqueeze(0).expand(G, N, K) if rs is None else rs[plan.group_first_row].view(G, 1, 1) * ws.unsqueeze(0)
            wc = term.clone() if wc is None else wc + term
        w_all = torch.zeros(G, 80, nfb * 128, dtype=torch.float32, device=weight.device)
        for fb in range(nfb):
            w = min(32, F - fb * 32)
            for a in range(4):
                c = 4 * fb + a
                w_all[:, :N, c * 32:c * 32 + w] = wc[:, :, a * F + fb * 32:a * F + fb * 32 + w]
    Kp = nfb * 128
    L = _lib.lib()
    nh = ctypes.c_int64(0)
    nb = L.pna_posttrans_x3_packed_bytes(Kp, G * 80, 1, 0, ctypes.byref(nh))
    img = torch.empty(nb // 4, dtype=torch.float32, device=weight.device)
    rc = L.pna_posttrans_x3_pack_f32(_lib.dev_ptr(w_all.view(G * 80, Kp), torch.float32, "weight"), Kp, G * 80, Kp, 1, 0,
                                     _lib.dev_ptr(img, torch.float32, "w_img"), None, _lib.stream_ptr(weight.device))
    _lib.check(rc, "pna_posttrans_x3_pack_f32")
    return img, nb // G


def rest_items(plan):
    """(work list, heavy_out) of the rows no degree group holds, output rows counted from the start of the rest region."""
    it, n_seg = plan.items, plan._n_seg
    light = it[n_seg:][it[n_seg:, 0] >= plan.NV].clone()
    light[:, 0] -= plan.NV
    items = torch.cat([it[:n_seg], light], dim=0).contiguous() if n_seg else light.contiguous()
    return items, None if plan.heavy_out is None else (plan.heavy_out - plan.NV).contiguous()


class Fused:
    def __init__(self, layer, g, h):
        self.layer, self.g, self.h, self.plan = layer, g, h, DG.plan_of(g)
        F = layer.in_dim
        assert 32 < F <= 96 and layer.out_dim <= 80 and h.stride(0) % 4 == 0 and h.stride(0) >= (F + 7) // 8 * 8 and h.data_ptr() % 16 == 0
        self.lin = layer.posttrans.fully_connected[0].linear
        self.scales = _row_scales(g, layer.scalers, layer.avg_d, h.device)
        self.img, self.stride = fused_images(self.lin.weight, F, self.scales, self.plan)
        self.cs, self.ct = PF._fold_batchnorm(layer.batchnorm_h)
        self.items, self.hout = rest_items(self.plan)
        self.rest_scales = self.plan.rest_scales(("df_check",), self.scales)
        self.y = torch.empty(h.shape[0], layer.out_dim, device=dev)

    def __call__(self):
        layer, g, h, plan = self.layer, self.g, self.h, self.plan
        F, N, K, csr = layer.in_dim, layer.out_dim, 4 * layer.in_dim, g.csr
        res = h if layer.residual else None
        a = Args()
        a.rowptr, a.col, a.x, a.ldx, a.F, a.N = csr.rowptr.data_ptr(), csr.col.data_ptr(), h.data_ptr(), h.stride(0), F, N
        a.row_perm, a.M, a.tile_image, a.tile_rows, a.relu = plan.perm.data_ptr(), plan.NV, plan.tile_image.data_ptr(), DG.TILE, 1
        a.w_img, a.image_stride, a.bias = self.img.data_ptr(), self.stride, self.lin.bias.data_ptr()
        a.col_scale, a.col_shift = self.cs.data_ptr(), self.ct.data_ptr()
        if res is not None:
            a.residual, a.ld_res = res.data_ptr(), res.stride(0)
        a.y, a.ldy, a.workgroups_per_cu = self.y.data_ptr(), self.y.stride(0), WGS
        rc = X.degree_fused_f32(ctypes.byref(a), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        if plan.NR:
            agg = torch.empty(plan.NRp, DG.agg_pitch(K), dtype=torch.float32, device=dev)[:, :K]
            ops.segreduce(csr.rowptr, csr.col, h, F, layer.aggregators, (None,), tower_stride_in=F, out=agg, heavy=g.heavy_schedule(),
                          workspace=g.workspace, items=self.items, heavy_out=self.hout, tune=dict(generic=2))
            ops.posttrans(agg, K, self.lin.weight, self.rest_scales, self.lin.bias, out=self.y, col_scale=self.cs, col_shift=self.ct, relu=True,
                          residual=res, row_perm=plan.perm_rest, n_out=N)
        return self.y


def make(V, E, F, seed, N=None):
    src, dst = powerlaw_graph(V, E, seed=seed, device=dev)
    g = Graph(src, dst, V)
    avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
    torch.manual_seed(seed)
    N = N or F
    layer = PNASimpleLayer(F, N, "mean max min std", "identity amplification attenuation", avg, 0.0, True, N == F).to(dev).eval()
    with torch.no_grad():
        layer.batchnorm_h.running_mean.normal_()
        layer.batchnorm_h.running_var.uniform_(0.5, 2.0)
    h = torch.randn(V, (F + 7) // 8 * 8, device=dev)[:, :F]
    return g, layer, h


ok = True
for V, E, F, N in [(200_000, 2_000_000, 75, 75), (150_000, 900_000, 96, 80), (140_000, 1_400_000, 50, 50), (131_072, 600_000, 64, 64),
                   (160_000, 1_000_000, 40, 72)]:
    g, layer, h = make(V, E, F, V % 89, N)
    with torch.no_grad():
        fused = Fused(layer, g, h)
        plan = fused.plan
        if os.environ.get("DF_DEBUG_AGG"):
            dbg = torch.zeros(plan.NV, 4 * F, device=dev)
            X.degree_fused_debug_agg(dbg.data_ptr(), 4 * F)
            fused()
            torch.cuda.synchronize()
            X.degree_fused_debug_agg(None, 0)
            ref = PF.degree_grouped_aggregate(layer, g, h, plan)[:plan.NV]
            real = plan.perm >= 0
            nbad = int((dbg[real] != ref[real]).any(dim=1).sum())
            print(f"  statistics seen by the contraction vs production aggregate: {'bit-identical' if nbad == 0 else f'{nbad} rows differ'}", flush=True)
        y_f = fused().clone()
        DG.ENABLED = True
        y_g = layer(g, h)
        DG.ENABLED = False
        y_p = layer(g, h)
        DG.ENABLED = True
    s = y_p.abs().max().item()
    e1, e2 = (y_f - y_g).abs().max().item() / s, (y_f - y_p).abs().max().item() / s
    good = bool(torch.isfinite(y_f).all()) and e1 <= 2e-6 and e2 <= 2e-5
    ok &= good
    print(f"V={V} E={E} F={F} N={N}: one-kernel vs two-kernel grouped {e1:.2e}, vs ordinary {e2:.2e} (of max|y|) {'ok' if good else 'BAD'}", flush=True)
print("CHECK", "PASS" if ok else "FAIL", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "time":
    def ev(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / n)
        return best
    g, layer, h = make(1_000_000, 10_000_000, 75, 1234)
    with torch.no_grad():
        fused = Fused(layer, g, h)
        print(f"C3 layer: one-kernel experiment {ev(fused):.3f} ms, shipped two-kernel degree-grouped layer {ev(lambda: layer(g, h)):.3f} ms", flush=True)
