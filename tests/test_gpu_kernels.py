"""GPU parity of the HIP kernels, called through the C ABI, against the CPU oracle (oracle/).

Bar (BASELINE.json north_star): max/min and degree-derived values BIT-EXACT; mean/sum/std/var and the
MLP within 1e-5 relative (fp32; plus the absolute floor that fp32 cancellation in E[x^2]-E[x]^2 and in
GEMM dot products makes unavoidable -- both stated next to each assert)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from pna_amd import Graph, ops
from pna_amd.graph import build_heavy_schedule
from conftest import check_blocks as _check_blocks, mass_stats as _mass

pytestmark = pytest.mark.gpu

ALL_AGG = ["mean", "max", "min", "std", "sum", "var"]


def _rand_graph(rng, V, E, hub=0):
    dst = rng.integers(0, V - 3, E)                    # the last 3 nodes have no in-edges
    if hub:
        dst[:hub] = rng.integers(0, 3, hub)            # a few very heavy destinations
    src = rng.integers(0, V, E)
    return torch.from_numpy(src), torch.from_numpy(dst)


@pytest.mark.parametrize("F", [1, 2, 3, 4, 5, 7, 16, 33, 75, 80, 128, 260])
def test_segreduce_widths(cuda_device, F):
    rng = np.random.default_rng(F)
    V, E = 700, 6000
    src, dst = _rand_graph(rng, V, E)
    g = Graph(src, dst, V).to(cuda_device)
    c = g.csr
    x = torch.randn(V, F, generator=torch.Generator().manual_seed(F))
    amp, att = g.degree_scalers(1.7)
    got = ops.segreduce(c.rowptr, c.col, x.to(cuda_device), F, ALL_AGG, [None, amp, att]).cpu().numpy()
    rp, col = c.rowptr.cpu().numpy(), c.col.cpu().numpy()
    amp_o, att_o = c_oracle.degree_scalers(rp, 1.7)
    assert np.array_equal(amp.cpu().numpy(), amp_o) and np.array_equal(att.cpu().numpy(), att_o)
    ref = c_oracle.segreduce(rp, col, x.numpy(), F, ALL_AGG, [None, amp_o, att_o])
    ref64 = c_oracle.segreduce(rp, col, x.numpy(), F, ALL_AGG, [None, amp_o, att_o], acc_double=True)
    mass = _mass(rp, x.numpy()[col])
    _check_blocks(got, ref, ref64, ALL_AGG, 3, F, f"F={F}", mass, [None, amp_o, att_o])
    # rows without in-edges are zero in every block
    empty = (rp[1:] == rp[:-1])
    assert empty.any() and not got[empty].any()


@pytest.mark.parametrize("ld_extra,tune", [(0, {}), (5, {}), (0, dict(vec=1)), (1, dict(prefetch=-1)), (0, dict(unroll=2)),
                                           (3, dict(unroll=8)), (0, dict(lanes_per_row=32)), (0, dict(lanes_per_row=10)),
                                           (0, dict(rows_per_group=1)), (0, dict(rows_per_group=16, nt_store=-1)), (0, dict(generic=1)),
                                           (2, dict(generic=1, unroll=8)), (0, dict(rows_per_group=3, unroll=2))])
def test_segreduce_tunings_and_strides_agree_bitwise(cuda_device, ld_extra, tune):
    """Every launch geometry computes bit-identical results (sequential edge order per row)."""
    rng = np.random.default_rng(3)
    V, E, F = 900, 9000, 75
    src, dst = _rand_graph(rng, V, E)
    g = Graph(src, dst, V).to(cuda_device)
    c = g.csr
    xs = torch.randn(V, F + ld_extra, generator=torch.Generator().manual_seed(1)).to(cuda_device)
    x = xs[:, :F]
    base = ops.segreduce(c.rowptr, c.col, x.contiguous(), F, ALL_AGG)
    got = ops.segreduce(c.rowptr, c.col, x, F, ALL_AGG, tune=tune)
    assert torch.equal(base, got)


def test_segreduce_heavy_rows_split(cuda_device):
    rng = np.random.default_rng(11)
    V, E, F = 500, 30000, 75
    src, dst = _rand_graph(rng, V, E, hub=12000)
    g = Graph(src, dst, V).to(cuda_device)
    c = g.csr
    assert c.max_degree > 2000
    x = torch.randn(V, F, generator=torch.Generator().manual_seed(5))
    rp, col = c.rowptr.cpu().numpy(), c.col.cpu().numpy()
    ref = c_oracle.segreduce(rp, col, x.numpy(), F, ALL_AGG)
    ref64 = c_oracle.segreduce(rp, col, x.numpy(), F, ALL_AGG, acc_double=True)
    xd = x.to(cuda_device)
    mass = _mass(rp, x.numpy()[col])
    outs = []
    for thr, seg in [(64, 64), (16, 8), (100, 37), (0, 0)]:
        hs = build_heavy_schedule(c.rowptr, c.max_degree, thr, seg) if thr else None
        got = ops.segreduce(c.rowptr, c.col, xd, F, ALL_AGG, heavy=hs).cpu().numpy()
        _check_blocks(got, ref, ref64, ALL_AGG, 1, F, f"thr={thr}", mass)
        outs.append(got)
    # same schedule, different launch geometry -> bit-identical (segment order is fixed)
    hs = build_heavy_schedule(c.rowptr, c.max_degree, 64, 64)
    a = ops.segreduce(c.rowptr, c.col, xd, F, ALL_AGG, heavy=hs, tune=dict(unroll=2, rows_per_group=2))
    assert np.array_equal(a.cpu().numpy(), outs[0])


@pytest.mark.parametrize("T,F", [(1, 75), (5, 15), (5, 75), (4, 4), (3, 2), (2, 130)])
def test_segreduce_towers_terms_weights_args(cuda_device, T, F):
    """n_tower slices + dst_term + edge_term + edge_weight + argmax/argmin (the EXTRA kernel path)."""
    rng = np.random.default_rng(T * 100 + F)
    V, E = 300, 5000
    src, dst = _rand_graph(rng, V, E, hub=900)
    g = Graph(src, dst, V).to(cuda_device)
    c = g.csr
    gen = torch.Generator().manual_seed(7)
    x, dt = torch.randn(V, T * F, generator=gen), torch.randn(V, T * F, generator=gen)
    et = torch.randn(E, T * F, generator=gen)
    w = (torch.rand(E, generator=gen) * 3).round() * 0.5           # weights incl. zeros
    hs = build_heavy_schedule(c.rowptr, c.max_degree, 32, 16)
    rp, col = c.rowptr.cpu().numpy(), c.col.cpu().numpy()
    for use_w in (False, True):
        got, amx, amn = ops.segreduce(c.rowptr, c.col, x.to(cuda_device), F, ALL_AGG, n_tower=T, tower_stride_in=F,
                                      dst_term=dt.to(cuda_device), edge_term=et.to(cuda_device),
                                      edge_weight=w.to(cuda_device) if use_w else None, want_arg=True, heavy=hs)
        got, amx, amn = got.cpu().numpy(), amx.cpu().numpy(), amn.cpu().numpy()
        A = len(ALL_AGG)
        for t in range(T):
            kw = dict(dst_term=dt.numpy(), edge_term=et.numpy(), edge_weight=w.numpy() if use_w else None, col_offset=t * F)
            ref = c_oracle.segreduce(rp, col, x.numpy(), F, ALL_AGG, **kw)
            ref64 = c_oracle.segreduce(rp, col, x.numpy(), F, ALL_AGG, acc_double=True, **kw)
            sl = got[:, t * A * F:(t + 1) * A * F]
            ok = np.isfinite(ref64).all(axis=1)                    # weighted rows with sum(w)=0 are NaN in both
            if use_w:
                assert np.array_equal(np.isnan(sl), np.isnan(ref))
            msg = (x.numpy()[col][:, t * F:(t + 1) * F] + dt.numpy()[np.repeat(np.arange(V), np.diff(rp))][:, t * F:(t + 1) * F]) \
                + et.numpy()[:, t * F:(t + 1) * F]
            mass = [m[ok] for m in _mass(rp, msg, w.numpy() if use_w else None)]
            _check_blocks(sl[ok], ref[ok], ref64[ok], ALL_AGG, 1, F, f"t={t} w={use_w}", mass)
            # argmax / argmin point at a CSR edge whose message equals the max / min
            ax, an = amx[:, t * F:(t + 1) * F], amn[:, t * F:(t + 1) * F]
            has = (ax >= 0)
            rows, cols = np.nonzero(has)
            assert np.array_equal(msg[ax[rows, cols], cols], sl[rows, cols + 1 * F])      # block 1 = max
            assert np.array_equal(msg[an[rows, cols], cols], sl[rows, cols + 2 * F])      # block 2 = min
            assert ((ax[rows, cols] >= rp[rows]) & (ax[rows, cols] < rp[rows + 1])).all()


@pytest.mark.parametrize("F,with_dst", [(75, False), (75, True), (16, False), (130, True)])
def test_hand_scheduled_gather_with_arg_tracking_equals_the_generic_kernel(cuda_device, F, with_dst):
    """The training forward's gather (argmax / argmin recorded for the backward) on the hand-scheduled kernel against the
    compiler-scheduled one it replaces there: every output block and both position arrays identical -- hub rows (partials +
    finalize), isolated rows, TIES (small-integer features: the FIRST extremal edge must win) and NaN messages included."""
    rng = np.random.default_rng(F)
    V, E = 3000, 40000
    src, dst = _rand_graph(rng, V, E, hub=1500)
    keep_e = (dst < 20) | (dst >= 40)                       # rows 20..39 without in-edges (0..2 are the hubs)
    g = Graph(src[keep_e], dst[keep_e], V).to(cuda_device)
    c = g.csr
    gen = torch.Generator().manual_seed(F)
    x = torch.randint(-3, 4, (V, F), generator=gen).float()               # many exact ties
    x[::7] += torch.randn(V, F, generator=gen)[::7]
    x[5, 3], x[77, 0] = float("nan"), float("inf")
    x = x.to(cuda_device)
    dt = torch.randint(-2, 3, (V, F), generator=gen).float().to(cuda_device) if with_dst else None
    res = {}
    for name, tune, items in (("fast", dict(generic=2), g.work_items()), ("generic", dict(generic=1), None)):
        out, amx, amn = ops.segreduce(c.rowptr, c.col, x, F, ["mean", "max", "min", "std"], (None,), tower_stride_in=F, dst_term=dt, want_arg=True,
                                      heavy=g.heavy_schedule(), workspace=g.workspace, items=items, tune=tune)
        res[name] = (out.cpu(), amx.cpu(), amn.cpu())
    (o1, x1, n1), (o2, x2, n2) = res["fast"], res["generic"]
    assert torch.equal(x1, x2) and torch.equal(n1, n2)
    assert torch.equal(torch.isnan(o1), torch.isnan(o2)) and torch.equal(torch.nan_to_num(o1, nan=0.0), torch.nan_to_num(o2, nan=0.0))
    assert int((x1 < 0).sum()) >= 20 * F and int((c.rowptr[1:] - c.rowptr[:-1]).max()) > 128


@pytest.mark.parametrize("T,F", [(1, 75), (5, 15), (4, 16), (1, 130)])
def test_hand_scheduled_gather_with_edge_terms_equals_the_generic_kernel(cuda_device, T, F):
    """The tower layers WITH edge features (models/dgl/pna_layer.py:35-40): message = x[src] + dst_term[v] + edge_term, on the
    hand-scheduled kernel -- per-edge term, and the term as a table of <= 4 edge types (ABI 14) -- against the compiler-scheduled
    kernel with the per-edge term: identical bits (the message is formed in the same order), hub rows and isolated rows included;
    and the type table through the compiler-scheduled kernel too."""
    rng = np.random.default_rng(T * 1000 + F)
    V, E = 3000, 40000
    src, dst = _rand_graph(rng, V, E, hub=1500)
    keep_e = (dst < 20) | (dst >= 40)
    g = Graph(src[keep_e], dst[keep_e], V).to(cuda_device)
    c = g.csr
    Ecsr = c.col.numel()
    gen = torch.Generator().manual_seed(F)
    x, dt = torch.randn(V, T * F, generator=gen).to(cuda_device), torch.randn(V, T * F, generator=gen).to(cuda_device)
    table = torch.randn(4, T * F, generator=gen).to(cuda_device)
    types = torch.randint(0, 4, (Ecsr,), generator=gen).to(torch.int32).to(cuda_device)
    per_edge = table[types.long()].contiguous()
    aggs = ["mean", "max", "min", "std"]
    kw = dict(n_tower=T, tower_stride_in=F, dst_term=dt, heavy=g.heavy_schedule(), workspace=g.workspace)
    ref = ops.segreduce(c.rowptr, c.col, x, F, aggs, (None,), edge_term=per_edge, tune=dict(generic=1), **kw).clone()
    fast_e = ops.segreduce(c.rowptr, c.col, x, F, aggs, (None,), edge_term=per_edge, items=g.work_items(), tune=dict(generic=2), **kw).clone()
    fast_t = ops.segreduce(c.rowptr, c.col, x, F, aggs, (None,), edge_term=table, edge_type=types, items=g.work_items(), tune=dict(generic=2), **kw).clone()
    gen_t = ops.segreduce(c.rowptr, c.col, x, F, aggs, (None,), edge_term=table, edge_type=types, tune=dict(generic=1), **kw).clone()
    assert torch.equal(fast_e, ref) and torch.equal(fast_t, ref) and torch.equal(gen_t, ref)
    # ... and a random per-edge term (no table) against the oracle-checked generic kernel
    et = torch.randn(Ecsr, T * F, generator=gen).to(cuda_device)
    a = ops.segreduce(c.rowptr, c.col, x, F, aggs, (None,), edge_term=et, tune=dict(generic=1), **kw).clone()
    b = ops.segreduce(c.rowptr, c.col, x, F, aggs, (None,), edge_term=et, items=g.work_items(), tune=dict(generic=2), **kw)
    assert torch.equal(a, b)
    assert int((c.rowptr[1:] - c.rowptr[:-1]).max()) > 128


def test_segreduce_edge_resident_messages(cuda_device):
    rng = np.random.default_rng(5)
    V, E, F = 400, 3000, 20
    src, dst = _rand_graph(rng, V, E)
    g = Graph(src, dst, V).to(cuda_device)
    c = g.csr
    m = torch.randn(E, F, generator=torch.Generator().manual_seed(2))
    got = ops.segreduce(c.rowptr, None, m.to(cuda_device), F, ["max", "mean", "min"]).cpu().numpy()
    ref = c_oracle.segreduce(c.rowptr.cpu().numpy(), None, m.numpy(), F, ["max", "mean", "min"])
    ref64 = c_oracle.segreduce(c.rowptr.cpu().numpy(), None, m.numpy(), F, ["max", "mean", "min"], acc_double=True)
    _check_blocks(got, ref, ref64, ["max", "mean", "min"], 1, F, "edge-resident", _mass(c.rowptr.cpu().numpy(), m.numpy()))


def test_segreduce_nan_inf_propagation(cuda_device):
    V, F = 6, 8
    src = torch.tensor([0, 1, 2, 3, 4, 5, 0, 1])
    dst = torch.tensor([0, 0, 0, 1, 1, 2, 3, 3])
    x = torch.randn(V, F)
    x[1, 2] = float("nan"); x[4, 0] = float("inf"); x[0, 5] = float("-inf")
    g = Graph(src, dst, V).to(cuda_device)
    c = g.csr
    got = ops.segreduce(c.rowptr, c.col, x.to(cuda_device), F, ["max", "min", "mean"]).cpu()
    from oracle import torch_oracle as O
    ref = O.reduce_bucketed(x[src], src, dst, V, ["max", "min", "mean"], ["identity"], torch.tensor(1.0))
    assert torch.equal(torch.isnan(got), torch.isnan(ref))
    ok = ~torch.isnan(ref)
    torch.testing.assert_close(got[ok], ref[ok], rtol=1e-6, atol=0)


def test_segreduce_argument_validation(cuda_device):
    g = Graph(torch.tensor([0, 1]), torch.tensor([1, 0]), 2).to(cuda_device)
    c = g.csr
    x = torch.randn(2, 8, device=cuda_device)
    with pytest.raises(KeyError):
        ops.segreduce(c.rowptr, c.col, x, 8, ["median"])
    with pytest.raises(RuntimeError, match="leading dimensions"):
        ops.segreduce(c.rowptr, c.col, x, 16, ["mean"])
    with pytest.raises(RuntimeError, match="unroll"):
        ops.segreduce(c.rowptr, c.col, x, 8, ["mean"], tune=dict(unroll=7))
    with pytest.raises(RuntimeError, match="GPU"):
        ops.segreduce(c.rowptr, c.col, x.cpu(), 8, ["mean"])
    with pytest.raises(TypeError):
        ops.segreduce(c.rowptr, c.col, x.double(), 8, ["mean"])


@pytest.mark.parametrize("M,K,N,S,Kh", [(1000, 300, 75, 3, 0), (777, 300, 15, 3, 75), (64, 16, 16, 1, 0), (65, 17, 5, 2, 3),
                                        (300, 60, 14, 3, 15), (129, 20, 100, 5, 4), (50, 8, 4, 1, 2), (2000, 320, 80, 3, 0)])
def test_posttrans_mfma_vs_float64(cuda_device, M, K, N, S, Kh):
    gen = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K + 3, generator=gen)[:, :K]                    # non-trivial lda
    W = torch.randn(N, Kh + S * K, generator=gen) / (K ** 0.5)
    b = torch.randn(N, generator=gen)
    h = torch.randn(M, Kh, generator=gen) if Kh else None
    scales = [None] + [torch.rand(M, generator=gen) + 0.5 for _ in range(S - 1)]
    from pna_amd import functional as PF
    dev = cuda_device
    got = PF.posttrans(a.to(dev), K, W.to(dev), b.to(dev), [None if s is None else s.to(dev) for s in scales],
                       None if h is None else h.to(dev)).cpu()
    # float64 reference of the reference's formulation: Linear(cat[h, s0*a, s1*a, ...])
    ad = a.double()
    cat = ([h.double()] if Kh else []) + [ad if s is None else ad * s.double().unsqueeze(1) for s in scales]
    z = torch.cat(cat, dim=1)
    ref = z @ W.double().t() + b.double()
    mag = z.abs() @ W.double().abs().t() + b.double().abs()             # sum of |terms| of every dot product
    err = (got.double() - ref).abs()
    # fp32 fma-chain bound: relative to the dot product's absolute mass (the quantity rounding scales with)
    assert (err <= 1e-5 * ref.abs() + 2e-6 * mag).all(), f"max err {err.max():.3e}"
    assert (err / mag).max() < 3e-6


def test_posttrans_fused_epilogue(cuda_device):
    """graph-norm * eval-BatchNorm affine * ReLU + residual folded into the MFMA epilogue (pna_layer.py:71-75,:209-213)."""
    from pna_amd import functional as PF
    gen = torch.Generator().manual_seed(3)
    M, K, N, S = 333, 40, 24, 3
    a = torch.randn(M, K, generator=gen)
    W = torch.randn(N, S * K, generator=gen) / (K ** 0.5)
    b = torch.randn(N, generator=gen)
    scales = [None, torch.rand(M, generator=gen) + 0.5, torch.rand(M, generator=gen) + 0.5]
    snorm = torch.rand(M, 1, generator=gen) + 0.2
    res = torch.randn(M, N, generator=gen)
    bn = torch.nn.BatchNorm1d(N).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(N, generator=gen) + 0.5); bn.bias.copy_(torch.randn(N, generator=gen))
        bn.running_mean.copy_(torch.randn(N, generator=gen)); bn.running_var.copy_(torch.rand(N, generator=gen) + 0.5)
        z = torch.cat([a if s is None else a * s.unsqueeze(1) for s in scales], dim=1).double() @ W.double().t() + b.double()
        ref = res.double() + torch.relu(bn.double()(z * snorm.double()))
        dev = cuda_device
        got = PF.posttrans(a.to(dev), K, W.to(dev), b.to(dev), [None if s is None else s.to(dev) for s in scales],
                           row_post=snorm.to(dev), bn=bn.float().to(dev), relu=True, residual=res.to(dev)).cpu()
    torch.testing.assert_close(got.double(), ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("T,F,pitch", [(1, 75, 80), (5, 15, 75), (5, 75, 375), (1, 128, 128), (2, 33, 70)])
def test_hand_scheduled_kernel_with_dst_term_equals_template(cuda_device, T, F, pitch):
    """The tower layers' message x[src] + dst_term[dst] on the hand-scheduled kernel (k_segreduce_fast<U, DST>): bit-identical
    to the compiler-scheduled template for every unroll, with towers, hub rows cut into segments, a NaN and an Inf message;
    max/min bit-exact and mean/std within the stated bound against the C oracle."""
    rng = np.random.default_rng(T * 1000 + F)
    V, E = 1200, 14000
    src, dst = _rand_graph(rng, V, E, hub=3000)
    g = Graph(src, dst, V).to(cuda_device)
    c = g.csr
    gen = torch.Generator().manual_seed(9)
    xs, ds = torch.randn(V, max(pitch, T * F), generator=gen), torch.randn(V, max(pitch, T * F) + 3, generator=gen)
    xs[7, 3], ds[11, 5] = float("nan"), float("inf")
    x, d = xs.to(cuda_device)[:, :T * F], ds.to(cuda_device)[:, :T * F]
    aggs = ["mean", "max", "min", "std"]
    hs = g.heavy_schedule(64, 32)
    items = g.work_items(64, 32)
    kw = dict(n_tower=T, tower_stride_in=F, dst_term=d, heavy=hs, workspace=g.workspace)
    ref = ops.segreduce(c.rowptr, c.col, x, F, aggs, items=items, tune=dict(generic=1), **kw)
    for U in (2, 3, 4, 5, 6, 8):
        got = ops.segreduce(c.rowptr, c.col, x, F, aggs, items=items, tune=dict(unroll=U, rows_per_group=1 + U % 3), **kw)
        assert torch.equal(torch.nan_to_num(got, nan=123.0), torch.nan_to_num(ref, nan=123.0)), U
        assert torch.equal(torch.isnan(got), torch.isnan(ref))
    rp, col = c.rowptr.cpu().numpy(), c.col.cpu().numpy()
    got = ref.cpu().numpy()
    xc, dc = xs.numpy()[:, :T * F].copy(), ds.numpy()[:, :T * F].copy()
    xc[7, 3], dc[11, 5] = 0.0, 0.0                                  # the oracle comparison runs on the finite problem
    x2, d2 = torch.from_numpy(xc).to(cuda_device), torch.from_numpy(dc).to(cuda_device)
    got = ops.segreduce(c.rowptr, c.col, x2, F, aggs, items=items, n_tower=T, tower_stride_in=F, dst_term=d2, heavy=hs,
                        workspace=g.workspace).cpu().numpy()
    A = len(aggs)
    for t in range(T):
        kwo = dict(dst_term=dc, col_offset=t * F)
        r32 = c_oracle.segreduce(rp, col, xc, F, aggs, **kwo)
        r64 = c_oracle.segreduce(rp, col, xc, F, aggs, acc_double=True, **kwo)
        msgs = xc[col][:, t * F:(t + 1) * F] + dc[np.repeat(np.arange(V), np.diff(rp))][:, t * F:(t + 1) * F]
        _check_blocks(got[:, t * A * F:(t + 1) * A * F], r32, r64, aggs, 1, F, f"tower {t}", _mass(rp, msgs))


@pytest.mark.parametrize("F", [4, 16, 75, 128])
def test_light_rows_are_bit_exact_in_every_block(cuda_device, F):
    """Rows that one lane group walks alone (in-degree <= the hub threshold) are summed in CSR edge order, exactly like the C
    restatement of the reference's reduce_func, and mean = s / D, E[x^2] = q / D are correctly rounded divisions (div_rn): so
    not only max / min but mean, sum, var and std are BIT-identical to the oracle there -- including rows of equal neighbours,
    where the reference's variance is exactly 0 and std exactly sqrt(1e-5) (the case a reciprocal-multiply mean gets wrong by
    percents).  Hub rows differ only through the association of their 128-edge segments."""
    rng = np.random.default_rng(F)
    V, E = 3000, 30000
    src, dst = _rand_graph(rng, V, E, hub=2000)
    g = Graph(src, dst, V).to(cuda_device)
    c = g.csr
    x = torch.randn(V, F, generator=torch.Generator().manual_seed(F)) * 3.0
    x[::7] = x[0]                                            # many equal source rows: exact-zero variances
    aggs = ["mean", "max", "min", "std", "sum", "var"]
    got = ops.segreduce(c.rowptr, c.col, x.to(cuda_device), F, aggs, heavy=g.heavy_schedule(), workspace=g.workspace,
                        items=g.work_items()).cpu().numpy()
    got4 = ops.segreduce(c.rowptr, c.col, x.to(cuda_device), F, aggs[:4], heavy=g.heavy_schedule(), workspace=g.workspace,
                         items=g.work_items()).cpu().numpy()                    # the hand-scheduled kernel (mean|max|min|std)
    rp, col = c.rowptr.cpu().numpy(), c.col.cpu().numpy()
    ref = c_oracle.segreduce(rp, col, x.numpy(), F, aggs)
    light = np.diff(rp) <= g.heavy_schedule().threshold
    assert light.sum() > 2900 and (~light).sum() >= 1
    assert np.array_equal(got[light], ref[light])
    assert np.array_equal(got4[light], ref[light][:, :4 * F])
    deg = np.diff(rp)
    eq = np.array([d > 1 and len(set(col[rp[v]:rp[v + 1]] % 7)) == 1 and (col[rp[v]] % 7 == 0) for v, d in enumerate(deg)])
    if eq.any():
        assert np.all(got[eq][:, 3 * F:4 * F] == np.float32(np.sqrt(np.float32(1e-5))))
