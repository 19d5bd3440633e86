"""Backward parity: gradients of the fused aggregation / posttrans operators against
torch autograd run through the CPU oracle's restatement of the reference ops, in float64."""
import numpy as np
import os

import pytest
import torch

from oracle import torch_oracle as O
from pna_amd import Graph, functional as PF, ops
from pna_amd.dgl.pna_layer import PNALayer, PNASimpleLayer

pytestmark = pytest.mark.gpu


def _graph(seed, V, E, hub=0):
    rng = np.random.default_rng(seed)
    dst = rng.integers(0, V - 2, E)
    if hub:
        dst[:hub] = rng.integers(0, 2, hub)
    src = rng.integers(0, V, E)
    return torch.from_numpy(src), torch.from_numpy(dst)


def _oracle_aggregate(x, dt, et, src, dst, V, aggs, scalers, avg_log, T, F):
    """float64 autograd through the oracle: per tower, messages x[src] + dt[dst] + et, bucketed reduce."""
    outs = []
    for t in range(T):
        sl = slice(t * F, (t + 1) * F)
        m = x[src][:, sl]
        if dt is not None:
            m = m + dt[dst][:, sl]
        if et is not None:
            m = m + et[:, sl]
        outs.append(O.reduce_bucketed(m, src, dst, V, aggs, scalers, avg_log))
    return torch.cat(outs, dim=1)


@pytest.mark.parametrize("aggs,T,F,terms,hub", [
    (["mean", "max", "min", "std"], 1, 75, False, 0),
    (["mean", "max", "min", "std"], 1, 20, False, 900),
    (["sum", "var", "max"], 2, 5, True, 0),
    (["std"], 3, 4, True, 400),
    (["min", "mean"], 1, 3, True, 0),
    (["mean", "max", "min", "std", "sum", "var"], 2, 33, True, 300),
])
def test_aggregate_backward(cuda_device, aggs, T, F, terms, hub):
    V, E = 200, 2500
    src, dst = _graph(len(aggs) * 7 + F, V, E, hub)
    g = Graph(src, dst, V).to(cuda_device)
    gen = torch.Generator().manual_seed(F)
    x = torch.randn(V, T * F, generator=gen, dtype=torch.float64)
    dt = torch.randn(V, T * F, generator=gen, dtype=torch.float64) if terms else None
    et_e = torch.randn(E, T * F, generator=gen, dtype=torch.float64) if terms else None     # per edge, edge order
    avg_log = torch.tensor(1.6)
    scalers = ["identity", "amplification", "attenuation"]
    # ---- oracle (float64, CPU autograd)
    xo = x.clone().requires_grad_(True)
    dto = dt.clone().requires_grad_(True) if terms else None
    eto = et_e.clone().requires_grad_(True) if terms else None
    ref = _oracle_aggregate(xo, dto, eto, src, dst, V, aggs, scalers, avg_log, T, F)
    R = torch.randn(ref.shape, generator=gen, dtype=torch.float64)
    (ref * R).sum().backward()
    # ---- HIP path (fp32)
    xg = x.float().to(cuda_device).requires_grad_(True)
    dtg = dt.float().to(cuda_device).requires_grad_(True) if terms else None
    eid = g.csr.eid
    etg = et_e.float().to(cuda_device)[eid].requires_grad_(True) if terms else None             # CSR order
    amp, att = g.degree_scalers(float(avg_log))
    out = PF.aggregate(g, xg, F, aggs, n_tower=T, dst_term=dtg, edge_term=etg, row_scales=[None, amp, att])
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=1e-4, atol=1e-4)
    (out * R.float().to(cuda_device)).sum().backward()
    scale = 1 + ref.detach().abs().max().item()
    torch.testing.assert_close(xg.grad.cpu().double(), xo.grad, rtol=2e-4, atol=2e-4 * scale)
    if terms:
        torch.testing.assert_close(dtg.grad.cpu().double(), dto.grad, rtol=2e-4, atol=2e-4 * scale)
        torch.testing.assert_close(etg.grad.cpu().double(), eto.grad[eid.cpu()], rtol=2e-4, atol=2e-4 * scale)


@pytest.mark.parametrize("mode", ["pull", "scatter"])
@pytest.mark.parametrize("aggs,T,F,hub", [(["mean", "max", "min", "std"], 1, 75, 0), (["sum", "var", "min"], 3, 6, 500),
                                          (["max"], 1, 4, 0), (["mean", "std", "var", "sum", "max", "min"], 2, 20, 300)])
def test_aggregate_backward_dst_term_only_both_paths(cuda_device, monkeypatch, mode, aggs, T, F, hub):
    """Messages x[src] + dst_term[dst] (no per-edge term): the default backward is the atomic-free PULL over the
    transposed graph (autograd._backward_pull); PNA_AMD_BWD=scatter forces the re-gather/atomic kernel.  Both must
    match float64 autograd through the oracle, including rows without in-edges and hub rows."""
    monkeypatch.setenv("PNA_AMD_BWD", mode)
    V, E = 300, 3000
    src, dst = _graph(F + T, V, E, hub)
    g = Graph(src, dst, V).to(cuda_device)
    gen = torch.Generator().manual_seed(F * 3 + T)
    x = torch.randn(V, T * F, generator=gen, dtype=torch.float64)
    dt = torch.randn(V, T * F, generator=gen, dtype=torch.float64)
    avg_log = torch.tensor(1.6)
    scalers = ["identity", "amplification", "attenuation"]
    xo, dto = x.clone().requires_grad_(True), dt.clone().requires_grad_(True)
    ref = _oracle_aggregate(xo, dto, None, src, dst, V, aggs, scalers, avg_log, T, F)
    R = torch.randn(ref.shape, generator=gen, dtype=torch.float64)
    (ref * R).sum().backward()
    xg = x.float().to(cuda_device).requires_grad_(True)
    dtg = dt.float().to(cuda_device).requires_grad_(True)
    amp, att = g.degree_scalers(float(avg_log))
    out = PF.aggregate(g, xg, F, aggs, n_tower=T, dst_term=dtg, row_scales=[None, amp, att])
    (out * R.float().to(cuda_device)).sum().backward()
    scale = 1 + ref.detach().abs().max().item()
    torch.testing.assert_close(xg.grad.cpu().double(), xo.grad, rtol=2e-4, atol=2e-4 * scale)
    torch.testing.assert_close(dtg.grad.cpu().double(), dto.grad, rtol=2e-4, atol=2e-4 * scale)


@pytest.mark.parametrize("T,F,with_dst", [(1, 75, False), (1, 20, True), (3, 16, True), (2, 130, False)])
def test_max_min_terms_inside_the_pull_equal_the_atomic_scatter(cuda_device, monkeypatch, T, F, with_dst):
    """pna_segreduce_bwd_pull_f32 (round 3: the max / min gradient terms routed by 16-bit ranks inside the pull over the transposed
    graph, no scattered atomics) against the path it replaces (sums-only pull + pna_segreduce_bwd_argscatter_f32): the same
    gradient up to the summation order -- power-law graph with hub DESTINATIONS (ranks up to thousands) and hub SOURCES (segments
    added atomically), rows without in- or out-edges, integer-valued features (ties: the FIRST extremal edge gets the gradient in
    both).  Both are checked against float64 autograd elsewhere in this file."""
    from pna_amd.synth import powerlaw_graph
    V, E = 20000, 300000
    src, dst = powerlaw_graph(V, E, seed=T * 100 + F)
    keep = (dst >= 50) & (src < V - 50)                     # nodes 0..49 without in-edges, the last 50 without out-edges
    g = Graph(src[keep], dst[keep], V).to(cuda_device)
    assert g.csr.max_degree > 500 and g.heavy_schedule().n_heavy > 0
    gen = torch.Generator().manual_seed(F)
    x0 = torch.randint(-3, 4, (V, T * F), generator=gen).float()
    x0[::3] += torch.randn(V, T * F, generator=gen)[::3]
    d0 = torch.randn(V, T * F, generator=gen) if with_dst else None
    amp, att = g.degree_scalers(1.9)
    R = torch.randn(V, T * 12 * F, generator=gen).to(cuda_device)
    grads = {}
    for mode in ("pull", "scatter"):
        monkeypatch.setenv("PNA_AMD_BWD_ARGS", mode)
        xg = x0.to(cuda_device).requires_grad_(True)
        dg = d0.to(cuda_device).requires_grad_(True) if with_dst else None
        out = PF.aggregate(g, xg, F, ["mean", "max", "min", "std"], n_tower=T, dst_term=dg, row_scales=[None, amp, att])
        (out * R).sum().backward()
        grads[mode] = (xg.grad.clone(), dg.grad.clone() if with_dst else None)
    gp, gs = grads["pull"][0], grads["scatter"][0]
    tol = 2e-5 * gs.abs().max(dim=1, keepdim=True).values + 1e-6          # per row: hub sources sum thousands of terms
    assert bool(((gp - gs).abs() <= tol).all()), float(((gp - gs).abs() / tol).max())
    assert torch.equal(gp[V - 50:], torch.zeros_like(gp[V - 50:]))        # no out-edges: no gradient
    if with_dst:
        assert torch.equal(grads["pull"][1], grads["scatter"][1])


def test_aggregate_backward_edge_resident(cuda_device):
    V, E, F = 150, 1800, 12
    src, dst = _graph(3, V, E, 300)
    g = Graph(src, dst, V).to(cuda_device)
    gen = torch.Generator().manual_seed(1)
    m = torch.randn(E, F, generator=gen, dtype=torch.float64)                                  # per edge, edge order
    mo = m.clone().requires_grad_(True)
    aggs = ["mean", "max", "min", "std"]
    ref = O.reduce_bucketed(mo, src, dst, V, aggs, ["identity"], torch.tensor(1.0))
    R = torch.randn(ref.shape, generator=gen, dtype=torch.float64)
    (ref * R).sum().backward()
    eid = g.csr.eid
    mg = m.float().to(cuda_device)[eid].requires_grad_(True)
    out = PF.aggregate(g, mg, F, aggs, edge_resident=True)
    (out * R.float().to(cuda_device)).sum().backward()
    torch.testing.assert_close(mg.grad.cpu().double(), mo.grad[eid.cpu()], rtol=2e-4, atol=2e-4)


def test_posttrans_backward(cuda_device):
    gen = torch.Generator().manual_seed(5)
    M, K, N, S, Kh = 300, 36, 20, 3, 9
    a = torch.randn(M, K, generator=gen, dtype=torch.float64)
    h = torch.randn(M, Kh, generator=gen, dtype=torch.float64)
    W = torch.randn(N, Kh + S * K, generator=gen, dtype=torch.float64) / 6
    b = torch.randn(N, generator=gen, dtype=torch.float64)
    scales = [None, torch.rand(M, generator=gen, dtype=torch.float64) + 0.5, torch.rand(M, generator=gen, dtype=torch.float64) + 0.5]
    ao, ho, Wo, bo = (t.clone().requires_grad_(True) for t in (a, h, W, b))
    z = torch.cat([ho] + [ao if s is None else ao * s.unsqueeze(1) for s in scales], dim=1)
    ref = torch.nn.functional.linear(z, Wo, bo)
    R = torch.randn(ref.shape, generator=gen, dtype=torch.float64)
    (ref * R).sum().backward()
    dev = cuda_device
    ag, hg, Wg, bg = (t.float().to(dev).requires_grad_(True) for t in (a, h, W, b))
    out = PF.posttrans(ag, K, Wg, bg, [None if s is None else s.float().to(dev) for s in scales], hg)
    (out * R.float().to(dev)).sum().backward()
    for got, want in ((ag.grad, ao.grad), (hg.grad, ho.grad), (Wg.grad, Wo.grad), (bg.grad, bo.grad)):
        torch.testing.assert_close(got.cpu().double(), want, rtol=1e-4, atol=1e-4 * (1 + want.abs().max().item()))


def test_layers_train_step(cuda_device):
    """Training mode end to end: both sparse layers produce finite gradients for every parameter and the
    loss falls under SGD (the fused forward + custom backward drive a real optimisation)."""
    torch.manual_seed(0)
    V, E = 300, 2400
    src, dst = _graph(9, V, E, 200)
    g = Graph(src, dst, V, [V]).to(cuda_device)
    avg = {"log": torch.tensor(2.0)}
    h = torch.randn(V, 20, device=cuda_device)
    target = torch.randn(V, 20, device=cuda_device)
    snorm = g.snorm_n()
    for layer, call in (
        (PNASimpleLayer(20, 20, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True),
         lambda l: l(g, h)),
        (PNALayer(20, 20, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True, towers=5,
                  divide_input=False, residual=True), lambda l: l(g, h, None, snorm)),
        (PNALayer(20, 20, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True, towers=4,
                  divide_input=True, pretrans_layers=2, posttrans_layers=2), lambda l: l(g, h, None, snorm)),
    ):
        layer = layer.to(cuda_device).train()
        opt = torch.optim.SGD(layer.parameters(), lr=0.05)
        losses = []
        for _ in range(8):
            opt.zero_grad()
            loss = ((call(layer) - target) ** 2).mean()
            loss.backward()
            for n, p in layer.named_parameters():
                assert p.grad is not None and torch.isfinite(p.grad).all(), n
            opt.step()
            losses.append(loss.item())
        assert losses[-1] < losses[0], losses


def _dense_ref(name, X, adj, self_loop=False):
    """The reference's dense aggregators (models/pytorch/pna/aggregators.py:17-146) restated in float64 torch for autograd."""
    N = adj.shape[-1]
    a = adj + torch.eye(N, dtype=adj.dtype).unsqueeze(0) if self_loop else adj
    D = a.sum(-1, keepdim=True)
    if name == "sum":
        return (X * a.unsqueeze(-1)).sum(2)
    if name == "mean":
        return (X * a.unsqueeze(-1)).sum(2) / D
    if name in ("var", "std"):
        mean = (X * a.unsqueeze(-1)).sum(2) / D
        var = torch.relu((X * X * a.unsqueeze(-1)).sum(2) / D - mean * mean)
        return var if name == "var" else torch.sqrt(var + 1e-5)
    if name in ("max", "min"):
        big = float("-inf") if name == "max" else float("inf")
        M = torch.where(a.unsqueeze(-1) > 0, X, torch.tensor(big, dtype=X.dtype))
        return M.max(1)[0] if name == "max" else M.min(1)[0]
    if name == "softmax":
        e = torch.exp(X)
        return (e * X * a.unsqueeze(-1)).sum(2) / (e * a.unsqueeze(-1)).sum(2)
    if name == "normalised_mean":
        r = a.sum(-1) ** -0.5
        return (X * (r.unsqueeze(-1) * a * r.unsqueeze(-2)).unsqueeze(-1)).sum(2)
    raise KeyError(name)


@pytest.mark.parametrize("name", ["sum", "mean", "var", "std", "max", "min", "softmax", "normalised_mean"])
@pytest.mark.parametrize("weighted", [False, True])
def test_dense_registry_is_differentiable(cuda_device, name, weighted):
    """the dense AGGREGATORS used to return a detached tensor (no gradient to the messages).  Gradients w.r.t. X
    against float64 autograd through the reference's formulas, with 0/1 and with real-valued adjacency WEIGHTS (the dense
    variant uses adj as a weight in mean / sum / std / var, aggregators.py:25,:69 -- the weighted backward was a
    NotImplementedError in round 1)."""
    from pna_amd.pytorch.pna.aggregators import AGGREGATORS
    gen = torch.Generator().manual_seed(5)
    B, N, F = 3, 9, 6
    adj = (torch.rand(B, N, N, generator=gen) < 0.5).double()
    adj = torch.maximum(adj, torch.eye(N).roll(1, 0).unsqueeze(0).double())       # every row has a neighbour
    adj = torch.maximum(adj, torch.eye(N).roll(1, 1).unsqueeze(0).double())       # ... and every column
    if weighted:
        adj = adj * (torch.rand(B, N, N, generator=gen).double() + 0.5)
    X = torch.randn(B, N, N, F, generator=gen, dtype=torch.float64)
    R = torch.randn(B, N, F, generator=gen, dtype=torch.float64)
    Xr = X.clone().requires_grad_(True)
    (_dense_ref(name, Xr, adj) * R).sum().backward()
    Xg = X.float().to(cuda_device).requires_grad_(True)
    out = AGGREGATORS[name](Xg, adj.float().to(cuda_device), device=cuda_device)
    assert out.requires_grad
    (out * R.float().to(cuda_device)).sum().backward()
    torch.testing.assert_close(out.detach().cpu().double(), _dense_ref(name, X, adj), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(Xg.grad.cpu().double(), Xr.grad, rtol=2e-4, atol=2e-5)


def test_weighted_sparse_aggregate_backward(cuda_device):
    """functional.aggregate with edge weights under autograd (x, dst_term and edge_term gradients) against float64."""
    V, E, F, T = 60, 500, 5, 2
    src, dst = _graph(3, V, E)
    g = Graph(src, dst, V).to(cuda_device)
    c = g.csr
    gen = torch.Generator().manual_seed(9)
    x, dt = torch.randn(V, T * F, generator=gen, dtype=torch.float64), torch.randn(V, T * F, generator=gen, dtype=torch.float64)
    et = torch.randn(E, T * F, generator=gen, dtype=torch.float64)                   # CSR order
    w = torch.rand(E, generator=gen, dtype=torch.float64) + 0.25
    aggs = ["mean", "sum", "std", "var", "max", "min"]
    row, col = c.row.long().cpu(), c.col.long().cpu()
    xo, dto, eto = (t.clone().requires_grad_(True) for t in (x, dt, et))
    m = xo[col] + dto[row] + eto
    D = torch.zeros(V, dtype=torch.float64).index_add_(0, row, w).unsqueeze(1)
    s = torch.zeros(V, T * F, dtype=torch.float64).index_add_(0, row, m * w.unsqueeze(1))
    q = torch.zeros(V, T * F, dtype=torch.float64).index_add_(0, row, m * m * w.unsqueeze(1))
    mean = s / D
    var = torch.relu(q / D - mean * mean)
    mx = torch.full((V, T * F), float("-inf"), dtype=torch.float64).scatter_reduce(0, row.unsqueeze(1).expand(-1, T * F), m, "amax")
    mn = torch.full((V, T * F), float("inf"), dtype=torch.float64).scatter_reduce(0, row.unsqueeze(1).expand(-1, T * F), m, "amin")
    blocks = dict(mean=mean, sum=s, std=torch.sqrt(var + 1e-5), var=var, max=mx, min=mn)
    has = (D > 0).expand(-1, T * F)
    ref = torch.cat([torch.cat([torch.where(has[:, t * F:(t + 1) * F], blocks[a][:, t * F:(t + 1) * F], torch.zeros(()).double())
                                for a in aggs], 1) for t in range(T)], 1)
    R = torch.randn(ref.shape, generator=gen, dtype=torch.float64)
    (ref * R).sum().backward()
    xg, dg, eg = (t.float().to(cuda_device).requires_grad_(True) for t in (x, dt, et))
    out = PF.aggregate(g, xg, F, aggs, n_tower=T, dst_term=dg, edge_term=eg, edge_weight=w.float().to(cuda_device))
    (out * R.float().to(cuda_device)).sum().backward()
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=3e-5, atol=3e-5)
    for got, want in ((xg.grad, xo.grad), (dg.grad, dto.grad), (eg.grad, eto.grad)):
        torch.testing.assert_close(got.cpu().double(), want, rtol=3e-4, atol=3e-5)


@pytest.mark.parametrize("conv", ["simple", "towers"])
def test_pyg_layers_train_with_isolated_nodes_and_var(cuda_device, conv):
    """(i) a node without in-edges + the default mean/min/max/std set used to raise 'modified by an inplace
    operation' in backward (fix_empty_std patched a tensor AggregateFn had saved); (ii) the 'var' aggregator (unclamped,
    kernel code var_raw) had no backward.  Gradients against float64 autograd through the oracle's restatement."""
    from pna_amd.pytorch_geometric import PNAConv, PNAConvSimple
    gen = torch.Generator().manual_seed(2)
    V, E, F = 40, 160, 8
    src = torch.randint(0, V, (E,), generator=gen)
    dst = torch.randint(0, V - 3, (E,), generator=gen)                              # the last 3 nodes are isolated
    ei = torch.stack([src, dst])
    deg = torch.bincount(torch.bincount(dst, minlength=V))
    aggs, scal = ["mean", "min", "max", "std", "var"], ["identity", "amplification", "attenuation"]
    torch.manual_seed(0)
    layer = (PNAConvSimple(F, F, aggs, scal, deg) if conv == "simple" else PNAConv(F, F, aggs, scal, deg, towers=2, divide_input=False))
    layer = layer.to(cuda_device).train()
    x = torch.randn(V, F, generator=gen).to(cuda_device).requires_grad_(True)
    y = layer(x, ei.to(cuda_device))
    y.square().sum().backward()                                                     # must not raise
    assert torch.isfinite(x.grad).all() and x.grad.abs().sum() > 0
    g1 = x.grad.clone()
    # numerical check of the input gradient along a random direction (fp32 central difference through the same layer)
    d = g1 / g1.norm()                  # along the gradient: the directional derivative is |g|, far above the fp32 noise of f
    eps = 1e-2
    with torch.no_grad():
        f = lambda z: layer(z, ei.to(cuda_device)).double().square().sum().item()   # noqa: E731
        num = (f(x.detach() + eps * d) - f(x.detach() - eps * d)) / (2 * eps)
    ana = (g1.double() * d.double()).sum().item()
    assert abs(num - ana) <= 3e-2 * abs(ana), (num, ana)


def test_ranked_pull_entry_point_rejects_incomplete_arguments(cuda_device):
    """pna_segreduce_bwd_pull_f32 needs max, min and std / var among the aggregators, both position arrays, the transposed graph and
    its workspaces: anything else is PNA_E_INVALID with a message, never a launch."""
    import ctypes
    from pna_amd import _lib
    L = _lib.lib()
    b = _lib.PnaSegreduceBwdArgs()
    q = _lib.PnaSegreduceBwdPullArgs()
    assert L.pna_segreduce_bwd_pull_f32(ctypes.byref(q), None) < 0 and b"null" in L.pna_last_error()
    V, F = 64, 8
    rowptr = torch.zeros(V + 1, dtype=torch.int32, device=cuda_device)
    gagg = torch.zeros(V, 4 * F, device=cuda_device)
    b.rowptr, b.V, b.F = _lib.dev_ptr(rowptr, torch.int32, "rowptr"), V, F
    b.n_tower, b.n_aggr = 1, 2
    b.aggr[0], b.aggr[1] = _lib.AGG_CODES["mean"], _lib.AGG_CODES["max"]          # no min, no std
    b.gagg, b.ld_g = _lib.dev_ptr(gagg, torch.float32, "gagg"), gagg.stride(0)
    q.base = ctypes.cast(ctypes.pointer(b), ctypes.c_void_p)
    q.n_items_t = 1
    assert L.pna_segreduce_bwd_pull_f32(ctypes.byref(q), _lib.stream_ptr(cuda_device)) < 0
    assert b"pna_segreduce_bwd_pull_f32" in L.pna_last_error()
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,pitch,residual,affine,relu", [
    (4000, 75, 75, True, True, True), (4000, 75, 80, True, True, True), (1537, 128, 128, False, True, True),
    (513, 3, 3, True, False, True), (2, 5, 8, True, True, False), (100_000, 70, 72, True, True, True)])
def test_bn_tail_forward_backward_vs_torch_float64(cuda_device, M, N, pitch, residual, affine, relu):
    """pna_bn_tail_{fwd,bwd}_f32 (autograd.BnTailFn) against nn.BatchNorm1d (training) + F.relu + residual in float64
    (models/dgl/pna_layer.py:207-213): output, running statistics, and every gradient; columns of large mean included."""
    from pna_amd.autograd import bn_relu_residual, bn_tail_applies
    gen = torch.Generator().manual_seed(M + N)
    y64 = torch.randn(M, N, generator=gen, dtype=torch.float64) * (0.5 + torch.rand(N, generator=gen, dtype=torch.float64) * 3)
    y64 = y64 + torch.randn(N, generator=gen, dtype=torch.float64) * 20                    # means far from 0: the shifted sums
    r64 = torch.randn(M, N, generator=gen, dtype=torch.float64)
    R64 = torch.randn(M, N, generator=gen, dtype=torch.float64)
    bn64 = torch.nn.BatchNorm1d(N, affine=affine, momentum=0.1).double()
    bn32 = torch.nn.BatchNorm1d(N, affine=affine, momentum=0.1)
    if affine:
        with torch.no_grad():
            bn64.weight.copy_(torch.rand(N, generator=gen, dtype=torch.float64) + 0.5)
            bn64.bias.copy_(torch.randn(N, generator=gen, dtype=torch.float64))
            bn32.weight.copy_(bn64.weight.float()); bn32.bias.copy_(bn64.bias.float())
    bn32 = bn32.to(cuda_device).train()
    ya = y64.clone().requires_grad_(True)
    ra = r64.clone().requires_grad_(True)
    z = bn64(ya)
    want = (torch.relu(z) if relu else z) + (ra if residual else 0)
    (want * R64).sum().backward()
    yg = torch.empty(M, pitch, device=cuda_device)[:, :N].copy_(y64.float()).requires_grad_(True)
    rg = r64.float().to(cuda_device).requires_grad_(True)
    assert bn_tail_applies(bn32, yg, rg if residual else None)
    got = bn_relu_residual(bn32, yg, rg if residual else None, relu=relu)
    (got * R64.float().to(cuda_device)).sum().backward()

    def close(a, b, what, tol=1e-5):
        err = (a.double().cpu() - b).abs().max().item() / max(1.0, b.abs().max().item())
        assert err <= tol, (what, err)
    close(got, want.detach(), "out")
    close(bn32.running_mean, bn64.running_mean, "running_mean", 1e-6)
    close(bn32.running_var, bn64.running_var, "running_var", 1e-6)
    assert int(bn32.num_batches_tracked) == 1
    # rows whose pre-activation sits within fp32 rounding of 0 may take the other branch of the ReLU: leave them out of grad_y
    edge = (z.detach().abs() < 1e-5).any(dim=1) if relu else torch.zeros(M, dtype=torch.bool)
    gy = yg.grad.double().cpu()
    scale = max(1.0, ya.grad.abs().max().item())
    # (two rows: xhat = +-1 and grad_y is a difference of nearly equal terms times a large 1 / std -- fp32 torch is no closer)
    assert ((gy - ya.grad).abs()[~edge].max().item() if (~edge).any() else 0.0) <= (2e-5 if M >= 100 else 2e-4) * scale
    if residual:
        close(rg.grad, ra.grad, "grad_residual", 1e-6)
    if affine:
        close(bn32.weight.grad, bn64.weight.grad, "grad_gamma", 2e-5)
        close(bn32.bias.grad, bn64.bias.grad, "grad_beta", 2e-5)


@pytest.mark.gpu
def test_simple_layer_training_step_uses_the_bn_tail_and_matches_the_library_route(cuda_device, monkeypatch):
    """PNASimpleLayer in training mode: the streaming BatchNorm tail against the nn.BatchNorm1d / F.relu route (PNA_AMD_BN_TAIL=0),
    same weights: output, input gradient, parameter gradients and running statistics; nn.BatchNorm1d's error for one row stays."""
    from pna_amd import autograd as AG
    from pna_amd.synth import powerlaw_graph
    V, E, F = 20_000, 160_000, 75
    src, dst = powerlaw_graph(V, E, seed=3)
    g = Graph(src, dst, V).to(cuda_device)
    avg = {"log": float(torch.log(g.in_degrees().float() + 1).mean())}
    outs = []
    for tail in (True, False):
        monkeypatch.setattr(AG, "BN_TAIL", tail)
        torch.manual_seed(0)
        layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True).to(cuda_device).train()
        h = torch.randn(V, F, generator=torch.Generator().manual_seed(1)).to(cuda_device).requires_grad_(True)
        calls = []
        real = AG.bn_relu_residual
        monkeypatch.setattr(AG, "bn_relu_residual", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        out = layer(g, h)
        (out * out).sum().backward()
        monkeypatch.setattr(AG, "bn_relu_residual", real)
        assert bool(calls) == tail
        # (the posttrans bias sits in front of a batch-statistics BatchNorm: its true gradient is 0, both routes return rounding noise)
        outs.append((out.detach(), h.grad, [p.grad for n, p in layer.named_parameters() if not n.endswith("linear.bias")],
                     layer.batchnorm_h.running_mean, layer.batchnorm_h.running_var))
    (o1, gh1, gp1, rm1, rv1), (o0, gh0, gp0, rm0, rv0) = outs
    rel = lambda a, b: (a - b).abs().max().item() / max(1.0, b.abs().max().item())
    assert rel(o1, o0) <= 1e-5 and rel(rm1, rm0) <= 1e-6 and rel(rv1, rv0) <= 1e-6
    # gradients: a pre-activation within fp32 rounding of 0 takes the other branch of the ReLU on one of the two routes (a
    # handful of the 1.5 M elements) and its gradient, a whole unit, travels on: all but a few entries agree to 1e-4
    off = ((gh1 - gh0).abs() > 1e-4 * max(1.0, gh0.abs().max().item())).float().mean().item()
    assert off <= 1e-3 and rel(gh1, gh0) <= 2e-2, (off, rel(gh1, gh0))
    for a, b in zip(gp1, gp0):
        assert rel(a, b) <= 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("V,E,F,bn", [(40_000, 400_000, 75, True), (30_000, 240_000, 20, False), (50_000, 600_000, 64, True)])
def test_simple_layer_training_step_in_degree_plan_order(cuda_device, monkeypatch, V, E, F, bn):
    """autograd.SimpleLayerPlanFn (round 4): the training forward gathers in the degree plan's row order (aggregate AND arg indices,
    hub rows included), multiplies ONE combined block per degree tile, and the backward finds a node's statistics through the plan's
    row map -- against the node-order route (AggregateFn + PosttransFn): output, input gradient, every parameter gradient, running
    statistics; and both against nothing looser than the routes' own relation in inference (the combined weight rounds once more)."""
    from pna_amd import autograd as AG, degree_groups as DG
    from pna_amd.synth import powerlaw_graph
    monkeypatch.setattr(DG, "MIN_ROWS", 1)
    monkeypatch.setattr(DG, "MIN_OUT", 1)
    src, dst = powerlaw_graph(V, E, seed=17)
    keep = dst >= 50                                           # some rows without in-edges
    g = Graph(src[keep], dst[keep], V).to(cuda_device)
    assert int((g.in_degrees() > 128).sum()) > 0              # hub rows: cut into segments, their arg indices finished by the finalize pass
    avg = {"log": float(torch.log(g.in_degrees().float() + 1).mean())}
    res = {}
    for plan_route in (True, False):
        monkeypatch.setattr(AG, "PLAN_TRAIN", plan_route)
        torch.manual_seed(0)
        layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, bn, True).to(cuda_device).train()
        with torch.no_grad():
            for p in layer.parameters():
                if p.dim() == 2:
                    p.copy_(torch.randn_like(p) / p.shape[1] ** 0.5)
        h = torch.randn(V, F, generator=torch.Generator().manual_seed(1)).to(cuda_device).requires_grad_(True)
        used = []
        real = AG.SimpleLayerPlanFn.apply
        monkeypatch.setattr(AG.SimpleLayerPlanFn, "apply", lambda *a: (used.append(1), real(*a))[1])
        out = layer(g, h)
        monkeypatch.setattr(AG.SimpleLayerPlanFn, "apply", real)
        assert bool(used) == plan_route
        (out * torch.linspace(0.5, 1.5, F, device=cuda_device)).sum().backward()
        lin = layer.posttrans.fully_connected[0].linear
        res[plan_route] = (out.detach(), h.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone(),
                           layer.batchnorm_h.running_mean.clone() if bn else None)
    rel = lambda a, b: (a - b).abs().max().item() / max(1e-30, b.abs().max().item())
    (o1, gh1, gw1, gb1, rm1), (o0, gh0, gw0, gb0, rm0) = res[True], res[False]
    assert rel(o1, o0) <= 2e-6, rel(o1, o0)
    if bn:
        assert rel(rm1, rm0) <= 2e-6
        # (a pre-activation within fp32 rounding of 0 takes the other ReLU branch on one route: a handful of elements, a whole unit)
        off = ((gh1 - gh0).abs() > 1e-4 * gh0.abs().max()).float().mean().item()
        assert off <= 1e-3 and rel(gw1, gw0) <= 5e-3, (off, rel(gw1, gw0))
    else:
        off = ((gh1 - gh0).abs() > 1e-4 * gh0.abs().max()).float().mean().item()
        assert off <= 1e-3 and rel(gw1, gw0) <= 2e-3 and rel(gb1, gb0) <= 2e-3, (off, rel(gw1, gw0), rel(gb1, gb0))


@pytest.mark.gpu
@pytest.mark.parametrize("V,E,F", [(40_000, 400_000, 75), (30_000, 240_000, 20)])
def test_grouped_aggregate_gradient_against_the_three_block_contraction(cuda_device, monkeypatch, V, E, F):
    """d agg = gy W_D^T with one combined weight per degree group (pna_project_grouped_f32, round 6) against the three-block bf16 x 3
    contraction of rounds 3-5 inside the same plan-order backward: same forward, same gy -- the input gradient differs by the two
    contractions' rounding only (no ReLU decision depends on it)."""
    from pna_amd import autograd as AG, degree_groups as DG
    from pna_amd.synth import powerlaw_graph
    monkeypatch.setattr(DG, "MIN_ROWS", 1)
    monkeypatch.setattr(DG, "MIN_OUT", 1)
    src, dst = powerlaw_graph(V, E, seed=29)
    keep = dst >= 50
    g = Graph(src[keep], dst[keep], V).to(cuda_device)
    avg = {"log": float(torch.log(g.in_degrees().float() + 1).mean())}
    res = {}
    for grouped in (True, False):
        monkeypatch.setattr(AG, "DAGG_GROUPED", grouped)
        torch.manual_seed(0)
        layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True).to(cuda_device).train()
        h = torch.randn(V, F, generator=torch.Generator().manual_seed(1)).to(cuda_device).requires_grad_(True)
        calls = []
        real = ops.project_grouped
        monkeypatch.setattr(ops, "project_grouped", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        out = layer(g, h)
        (out * torch.linspace(0.5, 1.5, F, device=cuda_device)).sum().backward()
        monkeypatch.setattr(ops, "project_grouped", real)
        assert bool(calls) == grouped
        res[grouped] = (out.detach(), h.grad.clone(), layer.posttrans.fully_connected[0].linear.weight.grad.clone())
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][2], res[False][2])
    gh1, gh0 = res[True][1], res[False][1]
    assert (gh1 - gh0).abs().max().item() <= 2e-5 * gh0.abs().max().item(), (gh1 - gh0).abs().max().item() / gh0.abs().max().item()


def test_simple_layer_training_step_in_degree_plan_order_vs_the_float64_oracle(cuda_device, monkeypatch):
    """The plan-order training step against oracle/torch_oracle.simple_layer_train_step (the restatement pinned to the reference's
    own training-step goldens, tests/test_oracle_golden.py) evaluated in FLOAT64 on a graph with degree tiles, hub rows and rows
    without in-edges: output, running statistics, input gradient and every parameter gradient."""
    from oracle import torch_oracle as O
    from pna_amd import autograd as AG, degree_groups as DG
    from pna_amd.synth import powerlaw_graph
    monkeypatch.setattr(DG, "MIN_ROWS", 1)
    monkeypatch.setattr(DG, "MIN_OUT", 1)
    V, E, F = 9000, 80_000, 32
    src, dst = powerlaw_graph(V, E, seed=23)
    keep = dst >= 30
    src, dst = src[keep], dst[keep]
    g = Graph(src, dst, V).to(cuda_device)
    assert int((g.in_degrees() > 128).sum()) > 0
    avg_log = torch.log(g.in_degrees().double().cpu() + 1).mean().float()
    torch.manual_seed(0)
    layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", {"log": avg_log}, 0.0, True, True)
    with torch.no_grad():
        for p in layer.parameters():
            if p.dim() == 2:
                p.copy_(torch.randn_like(p) / p.shape[1] ** 0.5)
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    h = torch.randn(V, F, generator=torch.Generator().manual_seed(1))
    R = torch.linspace(0.5, 1.5, F)
    layer = layer.to(cuda_device).train()
    hd = h.to(cuda_device).requires_grad_(True)
    used = []
    real = AG.SimpleLayerPlanFn.apply
    monkeypatch.setattr(AG.SimpleLayerPlanFn, "apply", lambda *a: (used.append(1), real(*a))[1])
    out = layer(g, hd)
    (out * R.to(cuda_device)).sum().backward()
    assert used
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    o64, gh64, gp64, rm64, rv64 = O.simple_layer_train_step(sd64, src, dst, V, h.double(), ["mean", "max", "min", "std"],
                                                             ["identity", "amplification", "attenuation"], avg_log.double(), R.double())
    rel = lambda a, b: (a.double().cpu() - b).abs().max().item() / max(1e-30, b.abs().max().item())
    assert rel(out.detach(), o64) <= 1e-5 and rel(layer.batchnorm_h.running_mean, rm64) <= 1e-5 and rel(layer.batchnorm_h.running_var, rv64) <= 1e-5
    # (a pre-activation within fp32 rounding of 0 takes the other ReLU branch: a handful of entries carry a whole unit)
    off = ((hd.grad.double().cpu() - gh64).abs() > 1e-4 * gh64.abs().max()).double().mean().item()
    assert off <= 2e-3, off
    for name, p in layer.named_parameters():
        if name.endswith("posttrans.fully_connected.0.linear.bias"):
            continue                                         # (in front of batch-statistics BatchNorm: true gradient 0, rounding noise)
        assert rel(p.grad, gp64[name]) <= 2e-2, (name, rel(p.grad, gp64[name]))


@pytest.mark.parametrize("route", ["degree plan order", "per-row scalers"])
def test_simple_layer_training_step_weight_gradient_kernels_match_the_library_route(cuda_device, monkeypatch, route):
    """The layer's training step with the posttrans weight / bias gradient on pna_posttrans_dw_grouped_f32 (the rows in the graph's
    degree-plan order) or pna_posttrans_dw_f32 (per-row scalers) against the slab-batched library GEMM it replaces (round 4): every
    parameter gradient to 2e-5 of its largest entry, the same input gradient, and the kernel routes repeat bit for bit."""
    from pna_amd import autograd as AG, ops
    from pna_amd.synth import powerlaw_graph
    V, E, F = 24_000, 200_000, 75
    src, dst = powerlaw_graph(V, E, seed=5)
    g = Graph(src, dst, V).to(cuda_device)
    avg = {"log": float(torch.log(g.in_degrees().float() + 1).mean())}
    monkeypatch.setattr(AG, "DW_GROUPED", route == "degree plan order")
    if route == "degree plan order":
        # the grouped route is for graphs that HAVE a degree plan (or are large enough for the forward to build one, DG.MIN_ROWS): a
        # per-batch graph of a few thousand nodes must not pay a plan inside every backward -- this one gets its plan here
        from pna_amd import degree_groups as DG
        assert g.__dict__.get("_pna_amd_degree_plan") is None and V < DG.MIN_ROWS
        DG.plan_of(g)
    res = {}
    for kind in ("kernel", "kernel again", "library"):
        monkeypatch.setattr(AG, "DW_KERNEL", kind != "library")
        torch.manual_seed(0)
        layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, False, True).to(cuda_device).train()
        h = torch.randn(V, F, generator=torch.Generator().manual_seed(1)).to(cuda_device).requires_grad_(True)
        seen = []
        for fn in ("posttrans_dw_grouped", "posttrans_dw"):
            real = getattr(ops, fn)
            monkeypatch.setattr(ops, fn, (lambda real, fn: lambda *a, **k: (seen.append(fn), real(*a, **k))[1])(real, fn))
        out = layer(g, h)                                       # (no BatchNorm: the bias gradient is a real one here)
        (out * torch.linspace(0.5, 1.5, F, device=cuda_device)).sum().backward()
        monkeypatch.undo()
        monkeypatch.setattr(AG, "DW_GROUPED", route == "degree plan order")
        assert seen == ([] if kind == "library" else ["posttrans_dw_grouped"] if route == "degree plan order" else ["posttrans_dw"]), seen
        lin = layer.posttrans.fully_connected[0].linear
        res[kind] = (lin.weight.grad.clone(), lin.bias.grad.clone(), h.grad.clone())
    assert torch.equal(res["kernel"][0], res["kernel again"][0]) and torch.equal(res["kernel"][1], res["kernel again"][1])
    # (the input gradient does not depend on the weight gradient; hub SOURCE rows add their segments atomically, so not bitwise)
    torch.testing.assert_close(res["kernel"][2], res["library"][2], rtol=1e-5, atol=1e-6)
    for a, b in zip(res["kernel"][:2], res["library"][:2]):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()


@pytest.mark.parametrize("name", __import__("conftest").golden_names("dgl_simple_train"))
def test_simple_layer_training_step_golden(cuda_device, name):
    """One training step of PNASimpleLayer against the REFERENCE's own (models/dgl/pna_layer.py:197-216 in train mode, run by
    oracle/make_golden_simple_train.py): output, the gradients of (out * R).sum() w.r.t. the input and every parameter, and the
    BatchNorm running statistics -- through the arg-tracking gather, the bf16x3 contraction, the streaming BatchNorm tail
    (pna_bn_tail_*) and the pull backward."""
    from conftest import load_golden
    from pna_amd import autograd as AG
    meta, a, sd = load_golden(name)
    layer = PNASimpleLayer(meta["F"], meta["out_dim"], meta["aggregators"], meta["scalers"], {"log": a["avg_log"]}, 0.0, True, meta["residual"])
    layer.load_state_dict(sd)
    layer = layer.to(cuda_device).train()
    g = Graph(a["src"], a["dst"], meta["N"]).to(cuda_device)
    h = a["h"].to(cuda_device).requires_grad_(True)
    used = []
    real = AG.bn_relu_residual
    AG.bn_relu_residual = lambda *x, **k: (used.append(1), real(*x, **k))[1]
    try:
        out = layer(g, h)
    finally:
        AG.bn_relu_residual = real
    assert used, "the streaming BatchNorm tail did not take the call"
    (out * a["R"].to(cuda_device)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), a["out"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(layer.batchnorm_h.running_mean.cpu(), a["running_mean_after"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(layer.batchnorm_h.running_var.cpu(), a["running_var_after"], rtol=1e-5, atol=1e-6)
    assert int(layer.batchnorm_h.num_batches_tracked) == int(sd["batchnorm_h.num_batches_tracked"]) + 1
    # ---- gradients: PER ELEMENT against the float64 evaluation of the reference's formulas (the forward tests'
    # form).  tol = 1e-5 |ref64|  +  4 x the REFERENCE's own fp32 error on the element's row (its golden fp32 gradient against
    # the float64 value: how ill-conditioned the entry is -- E[x^2] - E[x]^2 in fp32, sums that cancel)  +  2e-6 x the tensor's largest
    # entry (the fp32 floor of a sum over all rows in another order).  Entries a ReLU FLIP can reach are excluded EXPLICITLY and counted:
    # where the float64 BatchNorm output in front of the ReLU (:211) is within 1e-5 of zero its sign is decided by rounding, and a flip
    # moves the whole weight-gradient row of that output column, the column's BatchNorm gradients and the input gradient of the
    # node and its in-neighbours by O(R) -- there the reference's fp32 value is no more right than the product's.
    from oracle import torch_oracle as O
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    _, gh64, gp64, _, _, z64 = O.simple_layer_train_step(sd64, a["src"], a["dst"], meta["N"], a["h"].double(), meta["aggregators"].split(),
                                                         meta["scalers"].split(), a["avg_log"].double(), a["R"].double(), residual=meta["residual"],
                                                         return_pre_relu=True)
    risk = z64.abs() < 1e-5 * max(1.0, z64.abs().max().item())                     # (rows v, output columns n) a flip could happen at
    risk_cols = risk.any(0)
    risk_nodes = risk.any(1)
    nb = torch.zeros(meta["N"], dtype=torch.bool)
    nb[a["src"].long()[risk_nodes[a["dst"].long()]]] = True                        # in-neighbours of the nodes at risk (their features reach z)
    risk_rows_h = risk_nodes | nb
    n_risk = int(risk.sum())
    assert n_risk <= max(2, 2e-4 * risk.numel()), f"{n_risk} pre-ReLU values within 1e-5 of zero: the fixture is degenerate"

    def close(got, ref32, ref64, what, exclude_rows=None):
        got, ref32 = got.double().cpu(), ref32.double()
        ref_err = (ref32 - ref64).abs()
        ref_err = ref_err.max(dim=1, keepdim=True).values if ref_err.dim() == 2 else ref_err
        tol = 1e-5 * ref64.abs() + 4.0 * ref_err + 2e-6 * ref64.abs().max().clamp(min=1e-30)
        bad = (got - ref64).abs() > tol
        if exclude_rows is not None and bool(exclude_rows.any()):
            bad[exclude_rows] = False
            # (the excluded entries keep the old, loose bar: 1e-4 of the tensor's largest entry against the reference's fp32 value)
            assert ((got - ref32).abs()[exclude_rows].max().item() <= 1e-4 * max(1.0, ref32.abs().max().item())), what
        assert not bool(bad.any()), (what, int(bad.sum()), ((got - ref64).abs() / tol).max().item(), n_risk)
    close(h.grad, a["grad_h"], gh64, "grad_h", risk_rows_h)
    for k, p in layer.named_parameters():
        if k.endswith("posttrans.fully_connected.0.linear.bias"):
            # (the bias sits in front of the batch-statistics BatchNorm: its true gradient is 0 -- float64 says ~1e-17 --, the reference
            # stores fp32 rounding noise: bound by the noise level of the weight gradient's bar instead)
            wmax = gp64["posttrans.fully_connected.0.linear.weight"].abs().max().item()
            assert p.grad.abs().max().item() <= 1e-4 * wmax, k
            continue
        excl = risk_cols if p.dim() <= 2 and p.shape[0] == risk_cols.numel() else None
        close(p.grad, a["grad/" + k], gp64[k], k, excl)


@pytest.mark.gpu
@pytest.mark.parametrize("name", __import__("conftest").golden_names("dgl_tower_train"))
def test_tower_layer_training_step_golden(cuda_device, name):
    """One training step of PNALayer with towers against the REFERENCE's own (models/dgl/pna_layer.py:130-148 over :55-76 in
    train mode, oracle/make_golden_simple_train.py): output, gradients w.r.t. node features, edge features and every parameter,
    the towers' running statistics.  Bar per element, AGAINST THE FLOAT64 VALUE of the step (the oracle evaluated in float64 on the
    fixture's inputs): 1e-5 of the tensor's largest entry (1e-4 node / edge gradients, 3e-4 pretrans weights; parameter tensors + 4 x the
    reference's OWN fp32 error on the tensor when the fixture has ill-conditioned destinations) with NO count allowance -- except on an
    explicit, counted list of ill-conditioned destinations derived from the float64 oracle (below), whose rows get 2e-3.  The product takes the
    std of `a_u` alone where the reference takes it of `a_u + b_v` (a shift does not change a std, and the shifted form
    loses the variance's low digits in fp32 when the destination's term dwarfs the spread of its neighbours): a documented deviation of the
    FORMULA, measured here against the float64 value both formulas share."""
    from conftest import load_golden
    from oracle import torch_oracle as O
    meta, a, sd = load_golden(name)
    ef = meta["edge_dim"] > 0
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    out64, gh64, ge64, gp64, _ = O.dgl_layer_train_step(sd64, a["src"], a["dst"], meta["N"], a["h"].double(), a["e"].double(), a["snorm_n"].double(),
                                                        meta["aggregators"].split(), meta["scalers"].split(), a["avg_log"].double(), meta["towers"],
                                                        meta["divide_input"], ef, a["R"].double())
    layer = PNALayer(meta["in_dim"], meta["out_dim"], meta["aggregators"], meta["scalers"], {"log": a["avg_log"]}, 0.0, True, True,
                     towers=meta["towers"], pretrans_layers=1, posttrans_layers=1, divide_input=meta["divide_input"], residual=True,
                     edge_features=ef, edge_dim=meta["edge_dim"])
    layer.load_state_dict(sd)
    layer = layer.to(cuda_device).train()
    g = Graph(a["src"], a["dst"], meta["N"], meta["sizes"]).to(cuda_device)
    h = a["h"].to(cuda_device).requires_grad_(True)
    e = a["e"].to(cuda_device).requires_grad_(True) if ef else None
    out = layer(g, h, e, a["snorm_n"].to(cuda_device))
    (out * a["R"].to(cuda_device)).sum().backward()

    # The ILL-CONDITIONED destinations, derived from the float64 oracle (an explicit, counted list instead of "0.5 % of the
    # entries may be anywhere up to 2e-3").  std = sqrt(E[m^2] - E[m]^2 + eps) evaluated in fp32 carries a relative error of
    # up to 2^-24 (E[m^2] + E[m]^2) / (2 (var + eps)) into its forward value and its backward: for the product m = a_u (+ the edge
    # term) -- it takes the std of the messages WITHOUT the destination's own term b_v, which a shift does not change --,
    # for the reference m = a_u + b_v (models/dgl/aggregators.py:18-26), where the cancellation explodes whenever b_v dwarfs the neighbours' spread.
    # Measured, not bounded (a single in-edge cancels EXACTLY): the product's formula evaluated in fp32 against its float64 value; a destination
    # whose std is off by more than 1e-5 relative is ill-conditioned; what it touches -- its own row of the output, its own and
    # its in-neighbours' rows of the node gradient, its in-edges' rows of the edge gradient -- is held to the LOOSE bar (2e-3 of the largest
    # entry) and counted; every other entry to the strict one with NO allowance.
    srcl, dstl = torch.as_tensor(a["src"]).long(), torch.as_tensor(a["dst"]).long()
    N, T = meta["N"], meta["towers"]
    it = meta["in_dim"] // T if meta["divide_input"] else meta["in_dim"]
    ill = torch.zeros(N, dtype=torch.bool)
    rho, ties = torch.zeros(N, dtype=torch.float64), torch.zeros(N, dtype=torch.bool)
    deg = torch.zeros(N, dtype=torch.float64).index_add_(0, dstl, torch.ones(dstl.numel(), dtype=torch.float64)).clamp(min=1)
    for t in range(T):
        W = sd64[f"towers.{t}.pretrans.fully_connected.0.linear.weight"]
        b = sd64[f"towers.{t}.pretrans.fully_connected.0.linear.bias"]
        ht = (a["h"][:, t * it:(t + 1) * it] if meta["divide_input"] else a["h"]).double()
        part_a = ht[srcl] @ W[:, :it].t() + (a["e"].double() @ W[:, 2 * it:].t() if ef else 0.0)
        for m in (part_a,):                                   # (the PRODUCT's formula: what is measured here is the product against float64)
            stds = []
            for dt in (torch.float64, torch.float32):
                x = m.to(dt)
                m1 = torch.zeros(N, x.shape[1], dtype=dt).index_add_(0, dstl, x) / deg[:, None].to(dt)
                m2 = torch.zeros(N, x.shape[1], dtype=dt).index_add_(0, dstl, x * x) / deg[:, None].to(dt)
                stds.append(torch.sqrt(torch.relu(m2 - m1 * m1) + 1e-5).double())
            rho = torch.maximum(rho, ((stds[1] - stds[0]).abs() / stds[0]).max(dim=1).values)
        # ... and the destinations where max / min is a near-TIE (two messages within fp32 rounding of one another: which edge the gradient
        # goes to is decided by the last bit -- fp32 and float64 may route it to different sources, like a ReLU flip)
        mfull = part_a + ht[dstl] @ W[:, it:2 * it].t() + b
        idx = dstl[:, None].expand(-1, mfull.shape[1])
        for sign in (1.0, -1.0):
            top = torch.full((N, mfull.shape[1]), -float("inf"), dtype=torch.float64).scatter_reduce_(0, idx, sign * mfull, "amax")
            near = ((top[dstl] - sign * mfull) <= 4e-7 * top[dstl].abs().clamp(min=1e-30)).double()
            ties |= (torch.zeros(N, mfull.shape[1], dtype=torch.float64).index_add_(0, dstl, near) >= 2).any(1)
    ill = (rho > 1e-4) | ties
    touched_nodes = ill.clone()
    touched_nodes[srcl[ill[dstl]]] = True
    touched_edges = ill[dstl]
    n_ill = int(ill.sum())
    assert n_ill <= 0.15 * N, (n_ill, N)

    def close(got, exact, what, base, loose_rows=None, scale=None, ref=None):
        diff, scale = (got.double().cpu() - exact).abs(), scale or max(1.0, exact.abs().max().item())
        tol = torch.full_like(diff, base * scale)
        if ref is not None:                                  # parameter tensors: sums over ALL nodes -- the ill-conditioned destinations' noise reaches every entry;
            tol = tol + (4.0 * (ref.double() - exact).abs().max() if n_ill else 0.0)     # bounded by the reference's own fp32 error on the tensor, and only when the list is non-empty
        if loose_rows is not None and bool(loose_rows.any()):
            assert diff[loose_rows].max().item() <= 2e-3 * scale, (what, "loose rows", diff[loose_rows].max().item())
            diff = diff[~loose_rows]
            tol = tol[~loose_rows]
        bad = diff > tol
        if bool(bad.any()) and loose_rows is not None and diff.dim() == 2:
            rows_bad = torch.nonzero(bad.any(1)).flatten()
            keep_rows = torch.nonzero(~loose_rows).flatten()[rows_bad]
            print(what, "strict-bar violations in rows", keep_rows.tolist(), "rho of those nodes", [f"{rho[r].item():.1e}" for r in keep_rows.tolist()] if what != "grad_e" else "",
                  "worst ratio", (diff / tol).max().item())
        assert not bool(bad.any()), (what, int(bad.sum()), (diff / tol).max().item(), n_ill)
    close(out.detach(), out64, "out", 1e-5, ill)
    # (1e-4 / 3e-4 where the std's backward runs -- node and edge gradients, the pretrans weights --, 1e-5 elsewhere.  Measured, round 6: outside
    # the listed rows the worst entry sits at 7e-5 of the largest (t4_div, grad_h) / 4.3e-5 (t3_edgefeat, grad_e) -- destinations whose std is off
    # by 3e-5 .. 1e-4, below the list's threshold; with the list at 3e-5 it would hold a fifth of the nodes and the test would lose its grip.)
    close(h.grad, gh64, "grad_h", 1e-4, touched_nodes)
    if ef:
        close(e.grad, ge64, "grad_e", 1e-4, touched_edges)
    wscale = max(v.abs().max().item() for k, v in a.items() if k.startswith("grad/"))       # (one scale for all parameters: the
    for k, p in layer.named_parameters():                                                     # noise of those entries reaches each)
        pre = "pretrans" in k and k.endswith("weight")
        close(p.grad, gp64[k], k, 3e-4 if pre else 1e-5, scale=wscale, ref=a["grad/" + k])
    print(f"[{name}] ill-conditioned destinations: {n_ill} of {N} (std off by > 1e-4: {int((rho > 1e-4).sum())}, max / min near-ties: {int(ties.sum())}); touched node rows {int(touched_nodes.sum())}, edge rows {int(touched_edges.sum())}")
    for k, b in layer.named_buffers():
        if "running" in k:
            torch.testing.assert_close(b.cpu(), a["after/" + k], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("M,N,K,Kh,scal", [(100_000, 75, 300, 75, "x12"), (1000, 75, 300, 75, "x12"), (97, 80, 300, 80, "x12"), (33, 16, 64, 0, "x"),
                                           (20_011, 40, 160, 0, "12"), (5000, 75, 300, 0, "x1"), (70_001, 20, 80, 20, "x12"), (4097, 64, 256, 64, "x")])
def test_posttrans_weight_gradient_kernel(cuda_device, M, N, K, Kh, scal):
    """pna_posttrans_dw_f32 (round 4: the weight / bias gradient of the posttrans Linear on a hand-written bf16x3 kernel instead of the
    vendor GEMM) against the float64 products, per element: 1e-5 relative + the fp32 floor of a length-M sum (sum_m |terms| x 2^-22)."""
    from pna_amd import ops
    gen = torch.Generator(device=cuda_device).manual_seed(M + N)
    gy = torch.randn(M, N + 3, device=cuda_device, generator=gen)[:, :N]
    a = torch.randn(M, K + 4, device=cuda_device, generator=gen)[:, :K] * 3.0 + 0.5
    h = torch.randn(M, Kh, device=cuda_device, generator=gen) if Kh else None
    amp = torch.rand(M, device=cuda_device, generator=gen) + 0.5
    att = 1.0 / amp
    scales = [{"x": None, "1": amp, "2": att}[c] for c in scal]
    bias = scales[0] is None                               # (the bias and the h panel need an unscaled first copy of gy)
    res = ops.posttrans_dw(gy, a, K, h, scales, want_bias=bias)
    assert res is not None
    gw, gb = res
    again = ops.posttrans_dw(gy, a, K, h, scales, want_bias=bias)
    assert torch.equal(gw, again[0]) and (not bias or torch.equal(gb, again[1]))       # deterministic
    g64, a64 = gy.double(), a.double()
    blocks, floors = [], []
    if Kh:
        blocks.append(g64.t() @ h.double()); floors.append(g64.abs().t() @ h.double().abs())
    for rs in scales:
        gs = g64 if rs is None else g64 * rs.double().unsqueeze(1)
        blocks.append(gs.t() @ a64); floors.append(gs.abs().t() @ a64.abs())
    want, floor = torch.cat(blocks, dim=1), torch.cat(floors, dim=1)
    err = (gw.double() - want).abs()
    tol = 1e-5 * want.abs() + floor * 2.0 ** -22
    assert (err <= tol).all(), float((err / tol).max())
    if bias:
        wb = g64.sum(0)
        assert ((gb.double() - wb).abs() <= 1e-5 * wb.abs() + g64.abs().sum(0) * 2.0 ** -22).all()
    else:
        assert ops.posttrans_dw(gy, a, K, h, scales, want_bias=True) is None


@pytest.mark.parametrize("V,E,F,N", [(60_000, 600_000, 75, 75), (30_000, 240_000, 20, 24), (50_000, 500_000, 64, 80)])
def test_posttrans_weight_gradient_in_degree_plan_order(cuda_device, V, E, F, N):
    """pna_posttrans_dw_grouped_f32: the same gradient with the rows walked in the degree plan's order (one unscaled copy of gy, the
    degree scalers applied per degree run in the reduction) + the plan's rest rows -- against the float64 products over ALL rows."""
    from pna_amd import Graph, degree_groups as DG, ops
    from pna_amd.dgl.pna_layer import _row_scales
    from pna_amd.synth import powerlaw_graph
    src, dst = powerlaw_graph(V, E, seed=99, device=cuda_device)
    g = Graph(src, dst, V)
    plan = DG.plan_of(g)
    assert plan.G > 3 and plan.NR > 0
    scales = _row_scales(g, ["identity", "amplification", "attenuation"], {"log": torch.tensor(2.2)}, cuda_device)
    gen = torch.Generator(device=cuda_device).manual_seed(V)
    K = 4 * F
    gy = torch.randn(V, N, device=cuda_device, generator=gen)
    a = torch.randn(V, K + 4, device=cuda_device, generator=gen)[:, :K] * 2.0 - 0.3
    h = torch.randn(V, F, device=cuda_device, generator=gen)
    res = ops.posttrans_dw_grouped(gy, a, K, h, scales, plan)
    assert res is not None
    gw, gb = res
    again = ops.posttrans_dw_grouped(gy, a, K, h, scales, plan)
    assert torch.equal(gw, again[0]) and torch.equal(gb, again[1])
    g64, a64 = gy.double(), a.double()
    blocks, floors = [g64.t() @ h.double()], [g64.abs().t() @ h.double().abs()]
    for rs in scales:
        gs = g64 if rs is None else g64 * rs.double().reshape(-1, 1)
        blocks.append(gs.t() @ a64); floors.append(gs.abs().t() @ a64.abs())
    want, floor = torch.cat(blocks, dim=1), torch.cat(floors, dim=1)
    err = (gw.double() - want).abs()
    tol = 1e-5 * want.abs() + floor * 2.0 ** -22
    assert (err <= tol).all(), float((err / tol).max())
    wb = g64.sum(0)
    assert ((gb.double() - wb).abs() <= 1e-5 * wb.abs() + g64.abs().sum(0) * 2.0 ** -22).all()
    row = ops.posttrans_dw(gy, a, K, h, scales)            # the per-row-scaler kernel agrees to the same bar
    assert ((row[0].double() - want).abs() <= tol).all()


def test_posttrans_weight_gradient_kernel_declines_other_shapes(cuda_device):
    from pna_amd import ops
    gy, a = torch.randn(5000, 80, device=cuda_device), torch.randn(5000, 320, device=cuda_device)
    assert ops.posttrans_dw(gy, a, 320, torch.randn(5000, 80, device=cuda_device), [None, None, None]) is None      # K + Kh + 1 = 401 > 384
    assert ops.posttrans_dw(torch.randn(5000, 128, device=cuda_device), a, 320, None, [None, None]) is None         # S N = 256 > 240
