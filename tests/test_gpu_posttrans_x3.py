"""pna_posttrans_x3_f32 (fp32 contraction evaluated as six bf16 partial products per multiply) against float64 and
against the exact-fp32 MFMA kernel: its error relative to the mass sum_k |a_k w_k| must stay at fp32 level (a few 1e-7;
the f32 kernel's own summation-order error is ~1e-7), on every shape class the kernels dispatch on (K tails, tiny K,
the h block, 1..3 scalers, all column-tile counts, M tails, more than 80 output columns)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # M, K, N, S, Kh, input scale
    (1000, 300, 75, 3, 0, 1.0), (257, 300, 75, 3, 0, 1.0), (64, 32, 16, 1, 0, 1.0), (100, 12, 5, 2, 0, 1.0),
    (300, 280, 70, 3, 70, 1.0), (129, 33, 40, 2, 7, 1.0), (50, 4, 80, 3, 4, 1.0), (513, 900, 150, 1, 0, 1.0),
    (1000, 300, 75, 3, 0, 1e4), (777, 64, 48, 3, 16, 1e-3), (5000, 75, 75, 3, 75, 1.0), (1, 300, 75, 3, 0, 1.0),
    (3000, 512, 128, 3, 0, 1.0), (1001, 512, 128, 3, 128, 1.0), (333, 100, 97, 2, 0, 1.0), (40000, 512, 128, 3, 0, 1.0),   # one 128-column block
]


def _ref64(a, K, W, scales, b, h):
    M, S = a.shape[0], len(scales)
    Kh = 0 if h is None else h.shape[1]
    y = b.double()[None, :].repeat(M, 1)
    mass = b.abs().double()[None, :].repeat(M, 1)
    if h is not None:
        y = y + h.double() @ W[:, :Kh].double().t()
        mass = mass + h.abs().double() @ W[:, :Kh].abs().double().t()
    for s in range(S):
        sc = torch.ones(M, device=a.device, dtype=torch.float64) if scales[s] is None else scales[s].double()
        Ws = W[:, Kh + s * K:Kh + (s + 1) * K].double()
        y = y + sc[:, None] * (a[:, :K].double() @ Ws.t())
        mass = mass + sc.abs()[:, None] * (a[:, :K].abs().double() @ Ws.abs().t())
    return y, mass


@pytest.mark.parametrize("M,K,N,S,Kh,scale", CASES)
def test_bf16x3_posttrans_is_fp32_accurate(M, K, N, S, Kh, scale):
    from pna_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(M + K)
    a = (torch.randn(M, K, generator=gen) * scale).to(dev)
    h = torch.randn(M, Kh, generator=gen).to(dev) if Kh else None
    W = (torch.randn(N, Kh + S * K, generator=gen) / (S * K) ** 0.5).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    scales = [None] + [(torch.rand(M, generator=gen) + 0.5).to(dev) for _ in range(S - 1)]
    y64, mass = _ref64(a, K, W, scales, b, h)
    err = {}
    for arith in ("f32", "bf16x3"):
        y = ops.posttrans(a, K, W, scales, b, h, arith=arith)
        err[arith] = ((y.double() - y64).abs() / mass).max().item()
    assert err["bf16x3"] <= 5e-7, err           # fp32 unit roundoff is 6e-8; the f32 kernel sits at 1-2.5e-7 here
    assert err["bf16x3"] <= 4 * err["f32"] + 1e-7, err
    # the two LDS pipelines of the kernel (2 buffers + barrier per chunk boundary / 3 buffers + mid-chunk barrier) do the
    # same arithmetic in the same order: bit-identical
    y2 = ops.posttrans(a, K, W, scales, b, h, arith="bf16x3", pipeline=2)
    assert torch.equal(y2, ops.posttrans(a, K, W, scales, b, h, arith="bf16x3", pipeline=3))


def test_bf16x3_fused_tail_matches_f32_kernel():
    """BatchNorm fold, ReLU, graph-norm, residual, non-contiguous output pitch: same epilogue semantics on both kernels."""
    from pna_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(3)
    M, K, N, S = 700, 300, 75, 3
    a = torch.randn(M, K, generator=gen).to(dev)
    W = (torch.randn(N, S * K, generator=gen) / 30).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    scales = [None, (torch.rand(M, generator=gen) + 0.5).to(dev), (torch.rand(M, generator=gen) + 0.5).to(dev)]
    cs, ct = (torch.rand(N, generator=gen) + 0.5).to(dev), torch.randn(N, generator=gen).to(dev)
    rp = (torch.rand(M, generator=gen) + 0.5).to(dev)
    res = torch.randn(M, 96, generator=gen).to(dev)[:, :N]
    outs = []
    for arith in ("f32", "bf16x3"):
        out = torch.full((M, 80), 7.0, device=dev)[:, :N]
        ops.posttrans(a, K, W, scales, b, out=out, row_post=rp, col_scale=cs, col_shift=ct, relu=True, residual=res, arith=arith)
        outs.append(out)
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-5 * outs[0].abs().max().item()
    assert torch.isnan(ops.posttrans(torch.full_like(a, float("nan")), K, W, scales, b, arith="bf16x3")).all()


@pytest.mark.parametrize("M,N,Kh,post,bn,relu,resid", [
    (1000, 75, 0, False, True, True, True),      # the simple layer's tail (C3 shape class): full tiles + a tail tile
    (960, 80, 0, True, True, False, False),      # every column tile whole, no tail tile, graph-norm + BN (tower tail)
    (400, 150, 0, False, False, True, True),     # two column blocks, the second one partial
    (777, 15, 15, True, True, False, False),     # tower layer: h block, one column tile
    (192, 75, 0, False, False, False, False),    # exactly one tile, nothing optional
    (5000, 40, 7, True, True, True, True),       # everything at once
    (1000, 128, 0, False, True, True, True),     # the 128-column block (80 < N <= 128; BASELINE configs[4]'s F): whole + tail rows
    (2005, 100, 100, True, True, True, True),    # ... partial last column tile, h block, everything
    (640, 81, 0, False, False, False, False),    # ... its narrowest case
])
def test_bf16x3_straight_line_epilogue_every_flag_combination(M, N, Kh, post, bn, relu, resid):
    """The straight-line epilogue (whole tiles) and the generic one (the matrix's last tile) against torch, elementwise,
    for every optional operand present/absent; writes go into a wider buffer whose other columns must stay untouched."""
    from pna_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(M * 131 + N)
    K, S = 64, 3
    a = torch.randn(M, K, generator=gen).to(dev)
    h = torch.randn(M, Kh, generator=gen).to(dev) if Kh else None
    W = (torch.randn(N, Kh + S * K, generator=gen) / 14).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    scales = [None, (torch.rand(M, generator=gen) + 0.5).to(dev), (torch.rand(M, generator=gen) + 0.5).to(dev)]
    cs = (torch.rand(N, generator=gen) - 0.5).to(dev) * 2 if bn else None      # both signs: ReLU sees negative scales
    ct = torch.randn(N, generator=gen).to(dev) if bn else None
    rp = (torch.rand(M, generator=gen) + 0.5).to(dev) if post else None
    res = torch.randn(M, N + 9, generator=gen).to(dev)[:, 3:3 + N] if resid else None
    buf = torch.full((M, N + 21), 7.0, device=dev)
    out = buf[:, 8:8 + N]
    ops.posttrans(a, K, W, scales, b, h, out=out, row_post=rp, col_scale=cs, col_shift=ct, relu=relu, residual=res, arith="bf16x3")
    for pl in (2, 3):
        o2 = torch.full((M, N), 3.0, device=dev)
        ops.posttrans(a, K, W, scales, b, h, out=o2, row_post=rp, col_scale=cs, col_shift=ct, relu=relu, residual=res, arith="bf16x3",
                      pipeline=pl)
        assert torch.equal(o2, out), pl
    z = b.double()[None, :] + (h.double() @ W[:, :Kh].double().t() if Kh else 0)
    for s_ in range(S):
        sc = 1.0 if scales[s_] is None else scales[s_].double()[:, None]
        z = z + sc * (a.double() @ W[:, Kh + s_ * K:Kh + (s_ + 1) * K].double().t())
    if post:
        z = z * rp.double()[:, None]
    if bn:
        z = z * cs.double() + ct.double()
    if relu:
        z = torch.relu(z)
    if resid:
        z = z + res.double()
    assert (out.double() - z).abs().max().item() <= 2e-5 * max(1.0, z.abs().max().item())
    assert (buf[:, :8] == 7.0).all() and (buf[:, 8 + N:] == 7.0).all()


@pytest.mark.parametrize("K", [32, 64, 96, 160])
@pytest.mark.parametrize("N,S,Kh", [(16, 3, 0), (16, 1, 0), (75, 3, 0), (15, 3, 15), (40, 2, 40), (75, 1, 75), (128, 3, 0), (128, 1, 128),
                                    (96, 2, 20)])
def test_bf16x3_short_k_every_pipeline_repeated(K, N, S, Kh):
    """Few K chunks per tile (1 .. 5 plus the h chunks) and few MFMA groups per chunk: the A fragment of the next chunk is
    consumed a few hundred cycles after its load was issued, so any use of an asm-loaded register ahead of its wait (a
    compiler-inserted copy, a mis-counted vmcnt) shows here -- it did once, see take() in pna_posttrans_x3.hip.  Both LDS
    pipelines, several M (one wavefront tile, several tiles, a tail), repeated, against the exact-f32 kernel."""
    from pna_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(K * 1000 + N * 10 + S)
    for M in (48, 200, 1000, 20000):
        a = torch.randn(M, K, generator=gen).to(dev)
        h = torch.randn(M, Kh, generator=gen).to(dev) if Kh else None
        W = (torch.randn(N, Kh + S * K, generator=gen) / 10).to(dev)
        b = torch.randn(N, generator=gen).to(dev)
        scales = [None] + [(torch.rand(M, generator=gen) + 0.5).to(dev) for _ in range(S - 1)]
        res = torch.randn(M, N, generator=gen).to(dev)
        ref = ops.posttrans(a, K, W, scales, b, h, relu=True, residual=res, arith="f32")
        tol = 2e-5 * max(1.0, ref.abs().max().item())
        for rep in range(4):
            for pl in (2, 3):
                y = ops.posttrans(a, K, W, scales, b, h, relu=True, residual=res, arith="bf16x3", pipeline=pl)
                assert (y - ref).abs().max().item() <= tol, (M, pl, rep)


def _adversarial_ref(a, K, W, scales):
    y64, mass = _ref64(a, K, W, scales, torch.zeros(W.shape[0], device=a.device), None)
    return y64, mass


def test_bf16x3_mixed_magnitudes_per_row():
    """Rows mixing 1e-20 ... 1e+20 magnitudes: the three-term split is exact at every exponent, so the error stays at the
    fp32 level relative to the mass sum |a_k w_k| (the small terms vanish in the fp32 accumulator of either kernel)."""
    from pna_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(11)
    M, K, N, S = 2000, 300, 75, 3
    expo = torch.rand(M, K, generator=gen) * 40 - 20
    a = (torch.randn(M, K, generator=gen) * 10.0 ** expo).to(dev)
    W = (torch.randn(N, S * K, generator=gen) / 30).to(dev)
    scales = [None, (torch.rand(M, generator=gen) + 0.5).to(dev), (torch.rand(M, generator=gen) + 0.5).to(dev)]
    y64, mass = _adversarial_ref(a, K, W, scales)
    err = {ar: ((ops.posttrans(a, K, W, scales, arith=ar).double() - y64).abs() / mass).max().item() for ar in ("f32", "bf16x3")}
    # measured: f32-MFMA kernel 3.6e-7, bf16x3 5.1e-7 (the row maxima of 600 k draws from a 40-decade distribution)
    assert err["bf16x3"] <= 1e-6 and err["bf16x3"] <= 4 * err["f32"] + 1e-7, err


def test_bf16x3_cancellation_between_scaler_blocks():
    """amp * (W_1 a) ~ -att * (W_2 a): the scaler blocks cancel to ~1e-6 of their size; both kernels must stay at fp32
    rounding of the MASS (not of the tiny result) -- the reference's own fp32 GEMM is no better."""
    from pna_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(12)
    M, K, N = 3000, 300, 75
    a = torch.randn(M, K, generator=gen).to(dev)
    W1 = torch.randn(N, K, generator=gen) / 17
    amp = (torch.rand(M, generator=gen) + 0.5)
    att = 1.0 / amp
    W = torch.cat([torch.zeros(N, K), W1, -W1 * (1 + 1e-6)], dim=1).to(dev)        # amp*W1 a - att*(1+1e-6) W1 a
    scales = [None, amp.to(dev), (att * amp * amp).to(dev)]                         # second scaler = amp again: exact cancellation up to 1e-6
    y64, mass = _adversarial_ref(a, K, W, scales)
    err = {ar: ((ops.posttrans(a, K, W, scales, arith=ar).double() - y64).abs() / mass).max().item() for ar in ("f32", "bf16x3")}
    assert err["bf16x3"] <= 5e-7 and err["bf16x3"] <= 4 * err["f32"] + 1e-7, err


def test_bf16x3_k900_equal_sign_sum():
    """All 900 products positive (no cancellation to hide behind): relative error of the SUM at the fp32 level."""
    from pna_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(13)
    M, K, N = 2048, 900, 75
    a = (torch.rand(M, K, generator=gen) + 0.1).to(dev)
    W = (torch.rand(N, K, generator=gen) + 0.1).to(dev)
    y64, mass = _adversarial_ref(a, K, W, [None])
    err = {ar: ((ops.posttrans(a, K, W, [None], arith=ar).double() - y64).abs() / y64.abs()).max().item() for ar in ("f32", "bf16x3")}
    # every rounding of the fp32 accumulator goes the same way here: measured 2.2e-6 (f32-MFMA kernel) and 2.0e-6 (bf16x3)
    assert err["bf16x3"] <= 5e-6 and err["bf16x3"] <= 1.5 * err["f32"] + 1e-7, err


def test_bf16x3_subnormal_operands():
    """fp32-subnormal inputs / weights, and small normal ones whose residual bf16 terms are subnormal.  The matrix pipe may
    flush bf16-subnormal terms: the bound is the fp32-level relative one PLUS an absolute 2^-126 per product and operand
    (what a flushed term can lose).  The exact-f32 kernel is held to the same bound."""
    from pna_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(14)
    M, K, N, S = 1024, 300, 75, 3
    tiny = 2.0 ** -126
    for amag, wmag in ((1e-40, 1.0), (1.0, 1e-40), (1e-36, 1.0), (3e-20, 3e-20)):
        a = (torch.randn(M, K, generator=gen).double() * amag).float().to(dev)
        W = (torch.randn(N, S * K, generator=gen).double() * wmag).float().to(dev)
        scales = [None, (torch.rand(M, generator=gen) + 0.5).to(dev), (torch.rand(M, generator=gen) + 0.5).to(dev)]
        y64, mass = _adversarial_ref(a, K, W, scales)
        # a flushed term of a_k loses < tiny * |w_k| (and vice versa); products below tiny are lost by fp32 itself
        flush = 3.0 * tiny * (W.abs().double().sum(1)[None, :] + S * a.abs().double().sum(1, keepdim=True)) + 4 * S * K * tiny
        for ar in ("f32", "bf16x3"):
            y = ops.posttrans(a, K, W, scales, arith=ar).double()
            bad = (y - y64).abs() > 5e-7 * mass + flush
            assert not bad.any(), (ar, amag, wmag, ((y - y64).abs() / (5e-7 * mass + flush)).max().item())
            assert torch.isfinite(y).all()


def test_bf16x3_single_infinity_keeps_the_fp32_pattern():
    """+-Inf elements in `a` (and one in the weight): the result has the fp32 kernel's exact Inf / NaN pattern, signs
    included -- also under weights that are exactly bf16-representable or zero (their residual terms are 0: the infinity is
    carried by the operand's lowest term so that it never meets them) -- and every other row is untouched."""
    from pna_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(15)
    M, K, N, S = 700, 300, 75, 3
    a = torch.randn(M, K, generator=gen)
    a[5, 17], a[300, 299], a[650, 0], a[651, 100] = float("inf"), float("-inf"), float("inf"), float("nan")
    a[420, 8], a[420, 200] = float("inf"), float("-inf")                       # Inf - Inf inside one row
    a = a.to(dev)
    W = torch.randn(N, S * K, generator=gen) / 30
    W[:, 17] = 0.5                                                            # bf16-exact weights under the +Inf of row 5
    W[3, 299], W[9, 0] = 0.0, 0.0                                             # Inf * 0 = NaN in fp32 as well
    scales = [None, (torch.rand(M, generator=gen) + 0.5).to(dev), (torch.rand(M, generator=gen) + 0.5).to(dev)]
    for Wv in (W, torch.where(torch.arange(S * K)[None, :] == 50, torch.tensor(float("inf")), W)):     # + an infinite weight column
        Wd = Wv.to(dev)
        y32 = ops.posttrans(a, K, Wd, scales, arith="f32")
        y3 = ops.posttrans(a, K, Wd, scales, arith="bf16x3")
        assert torch.equal(torch.isnan(y32), torch.isnan(y3))
        assert torch.equal(torch.isinf(y32), torch.isinf(y3))
        inf = torch.isinf(y32)
        assert inf.any() and torch.equal(torch.sign(y32[inf]), torch.sign(y3[inf]))
        fin = torch.isfinite(y32)
        if fin.any():
            assert (y32[fin] - y3[fin]).abs().max().item() <= 2e-5 * y32[fin].abs().max().item()
    y32 = ops.posttrans(a, K, W.to(dev), scales, arith="f32")
    assert torch.isfinite(y32[[0, 1, 4, 6, 299, 301, 419, 421, 649, 652]]).all()          # only the poisoned rows are non-finite
    assert not torch.isfinite(y32[[5, 300, 420, 650, 651]]).any()


def test_bf16x3_rejects_more_than_three_scalers():
    from pna_amd import ops
    dev = torch.device("cuda:0")
    a, W = torch.randn(8, 16, device=dev), torch.randn(4, 64, device=dev)
    with pytest.raises(ValueError, match="at most 3 scalers"):
        ops.posttrans(a, 16, W, [None] * 4, arith="bf16x3")


def _force_x3(monkeypatch):
    from pna_amd import ops
    monkeypatch.setattr(ops, "POSTTRANS_ARITH", "bf16x3")
    from pna_amd import functional as PF
    monkeypatch.setattr(PF, "SMALL_SIMPLE_ROWS", 0)          # (the one-call small-batch path has its own contraction)


@pytest.mark.parametrize("name", __import__("conftest").golden_names("dgl_simple"))
def test_simple_layer_golden_through_bf16x3(cuda_device, monkeypatch, name):
    """The reference's own output (golden fixture) through the layer with the bf16x3 contraction forced."""
    from conftest import load_golden
    from pna_amd import Graph
    from pna_amd.dgl.pna_layer import PNASimpleLayer
    meta, a, sd = load_golden(name)
    if len(meta["scalers"].split()) > 3:
        pytest.skip("bf16x3 kernel: at most 3 scalers")
    _force_x3(monkeypatch)
    layer = PNASimpleLayer(meta["F"], meta["out_dim"], meta["aggregators"], meta["scalers"], {"log": a["avg_log"]}, 0.0,
                           True, meta["residual"], posttrans_layers=meta["posttrans_layers"])
    layer.load_state_dict(sd)
    layer = layer.to(cuda_device).eval()
    g = Graph(a["src"], a["dst"], meta["N"]).to(cuda_device)
    with torch.no_grad():
        out = layer(g, a["h"].to(cuda_device)).cpu()
    torch.testing.assert_close(out, a["out"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", __import__("conftest").golden_names("dgl_tower"))
def test_tower_layer_golden_through_bf16x3(cuda_device, monkeypatch, name):
    from conftest import load_golden
    from pna_amd import Graph
    from pna_amd.dgl.pna_layer import PNALayer
    meta, a, sd = load_golden(name)
    if len(meta["scalers"].split()) > 3:
        pytest.skip("bf16x3 kernel: at most 3 scalers")
    _force_x3(monkeypatch)
    layer = PNALayer(meta["in_dim"], meta["out_dim"], meta["aggregators"], meta["scalers"], {"log": a["avg_log"]}, 0.0,
                     meta["graph_norm"], meta["batch_norm"], towers=meta["towers"],
                     pretrans_layers=meta["pretrans_layers"], posttrans_layers=meta["posttrans_layers"],
                     divide_input=meta["divide_input"], residual=meta["residual"], edge_features=meta["edge_dim"] > 0,
                     edge_dim=meta["edge_dim"])
    layer.load_state_dict(sd)
    layer = layer.to(cuda_device).eval()
    g = Graph(a["src"], a["dst"], meta["N"], meta["sizes"]).to(cuda_device)
    e = a["e"].to(cuda_device) if meta["edge_dim"] > 0 else None
    with torch.no_grad():
        out = layer(g, a["h"].to(cuda_device), e, a["snorm_n"].to(cuda_device)).cpu()
    torch.testing.assert_close(out, a["out"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("arith,M", [("f32", 700), ("bf16x3", 700), ("bf16x3", 20000)])
def test_leaky_relu_epilogue_and_tower_batching(arith, M):
    """ABI 6: (i) relu = 2 / act_slope = LeakyReLU (the mixing network of the tower layers, models/layers.py:157) against torch,
    including -Inf (LeakyReLU(-Inf) = -Inf, ReLU(-Inf) = 0) and NaN inputs; (ii) n_tower contractions in one call = the same
    contractions one by one, bit for bit, with per-tower h slices / shared h, biases and folded BatchNorm constants."""
    from pna_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(M)
    K, N = 48, 20
    a = torch.randn(M, K, generator=gen).to(dev)
    a[3, 5], a[9, 1] = float("-inf"), float("nan")
    W = (torch.randn(N, K, generator=gen).abs() / 7).to(dev)                      # positive weights: row 3 is -Inf everywhere
    b = torch.randn(N, generator=gen).to(dev)
    res = torch.randn(M, N, generator=gen).to(dev)
    y = ops.posttrans(a, K, W, [None], b, leaky_slope=0.01, residual=res, arith=arith)
    ref = res + torch.nn.functional.leaky_relu(a @ W.t() + b, 0.01)
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isnan(y), torch.isnan(ref)) and torch.equal(torch.isinf(y), torch.isinf(ref))
    assert (y[fin] - ref[fin]).abs().max().item() <= 2e-5 * ref[fin].abs().max().item()
    yr = ops.posttrans(a, K, W, [None], b, relu=True, arith=arith)
    assert torch.equal(yr[3], torch.zeros(N, device=dev)) and torch.isnan(yr[9]).all()          # ReLU(-Inf) = +0, NaN kept
    assert not torch.signbit(yr[3]).any()
    # towers
    T, S, Kh = 5, 3, 8
    agg = torch.randn(M, T * K, generator=gen).to(dev)
    scales = [None, (torch.rand(M, generator=gen) + 0.5).to(dev), (torch.rand(M, generator=gen) + 0.5).to(dev)]
    Ws = [(torch.randn(N, Kh + S * K, generator=gen) / 12).to(dev) for _ in range(T)]
    bs = torch.randn(T, N, generator=gen).to(dev)
    cs, ct = (torch.rand(T, N, generator=gen) + 0.5).to(dev), torch.randn(T, N, generator=gen).to(dev)
    rp = (torch.rand(M, generator=gen) + 0.5).to(dev)
    for shared in (True, False):
        h = torch.randn(M, Kh if shared else T * Kh, generator=gen).to(dev)
        out = torch.full((M, T * N + 3), 5.0, device=dev)
        ops.posttrans_towers(agg, K, Ws, scales, bs, h, shared, out[:, :T * N], row_post=rp, col_scale=cs, col_shift=ct, relu=True, arith=arith)
        for t in range(T):
            want = ops.posttrans(agg[:, t * K:(t + 1) * K], K, Ws[t], scales, bs[t], h if shared else h[:, t * Kh:(t + 1) * Kh],
                                 row_post=rp, col_scale=cs[t], col_shift=ct[t], relu=True, arith=arith)
            assert torch.equal(out[:, t * N:(t + 1) * N], want), (shared, t)
        assert (out[:, T * N:] == 5.0).all()
