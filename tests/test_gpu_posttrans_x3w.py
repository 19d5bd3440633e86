"""pna_posttrans_x3w_f32 -- the bf16x3 contraction on 32x32x16 matrix-core tiles (packed scaler columns, a VALU column, an
LDS-staged coalesced epilogue; opt-in, PNA_AMD_X3_WIDE=1) -- against float64, against the exact-fp32 MFMA kernel and against
the 16x16 bf16x3 kernel: same accuracy bar (error relative to the mass sum_k |a_k w_k| at fp32 level), same epilogue semantics,
same Inf / NaN pattern, deterministic and independent of a row's position."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# M, K, N, tail flags, output pitch, input pitch: every instantiation (N = 75: 7 tiles + VALU column; 76..80: 8 tiles;
# 65..74: 7 tiles; 64 / 128: 6 tiles per 64-column block), K tails of 4..12 columns, partial row tiles, unaligned pitches
CASES = [
    (1000, 300, 75, False, None, None), (257, 300, 75, True, None, None), (31, 300, 75, False, None, None),
    (4096, 300, 75, True, 80, None), (1000, 320, 80, True, None, None), (777, 300, 70, True, None, None),
    (513, 256, 64, False, None, None), (2000, 512, 128, True, None, None), (300, 44, 75, False, None, None),
    (300, 20, 66, True, None, None), (70000, 300, 75, True, 80, 304), (1000, 300, 75, True, 77, 301), (1, 4, 64, False, None, None),
    (5000, 308, 77, True, 80, None), (999, 296, 74, True, 76, None), (640, 192, 192, True, None, None),
]


@pytest.fixture()
def wide():
    from pna_amd import ops
    keep = ops.X3_WIDE
    ops.X3_WIDE = True
    yield ops
    ops.X3_WIDE = keep


def _inputs(M, K, N, seed, lda=None, S=3):
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(seed)
    a = torch.randn(M, lda or K, generator=gen).to(dev)[:, :K]
    W = (torch.randn(N, S * K, generator=gen) / (S * K) ** 0.5).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    scales = [None] + [(torch.rand(M, generator=gen) + 0.5).to(dev) for _ in range(S - 1)]
    tail = dict(row_post=(torch.rand(M, generator=gen) + 0.5).to(dev), col_scale=(torch.rand(N, generator=gen) + 0.5).to(dev),
                col_shift=torch.randn(N, generator=gen).to(dev), relu=True, residual=torch.randn(M, N, generator=gen).to(dev))
    return a, W, b, scales, tail


def _ref64(a, K, W, scales, b, tail):
    M = a.shape[0]
    y = b.double()[None, :].repeat(M, 1)
    mass = b.abs().double()[None, :].repeat(M, 1)
    for s, sc in enumerate(scales):
        scd = torch.ones(M, device=a.device, dtype=torch.float64) if sc is None else sc.double()
        Ws = W[:, s * K:(s + 1) * K].double()
        y = y + scd[:, None] * (a.double() @ Ws.t())
        mass = mass + scd.abs()[:, None] * (a.abs().double() @ Ws.abs().t())
    if tail:
        z = (y * tail["row_post"].double()[:, None]) * tail["col_scale"].double() + tail["col_shift"].double()
        y = tail["residual"].double() + torch.relu(z)
        mass = mass * tail["row_post"].double()[:, None] * tail["col_scale"].double() + tail["col_shift"].abs().double() + tail["residual"].abs().double()
    return y, mass


@pytest.mark.parametrize("M,K,N,with_tail,ldy,lda", CASES)
def test_wide_kernel_is_fp32_accurate(wide, M, K, N, with_tail, ldy, lda):
    ops = wide
    assert ops.x3w_supported(K, N, 3, 0)
    a, W, b, scales, tail = _inputs(M, K, N, M + K, lda)
    kw = tail if with_tail else {}
    y64, mass = _ref64(a, K, W, scales, b, kw)
    out = torch.full((M, ldy), float("nan"), device=a.device)[:, :N] if ldy else None
    y = ops.posttrans(a, K, W, scales, b, arith="bf16x3", out=out, **kw)
    err = ((y.double() - y64).abs() / mass).max().item()
    y32 = ops.posttrans(a, K, W, scales, b, arith="f32", **kw)
    err32 = ((y32.double() - y64).abs() / mass).max().item()
    assert err <= 5e-7 and err <= 4 * err32 + 1e-7, (err, err32)
    if ldy:                                                  # nothing outside the N columns of a padded output is written
        assert torch.isnan(out.as_strided((M, ldy - N), (ldy, 1), N)).all()
    assert torch.equal(y, ops.posttrans(a, K, W, scales, b, arith="bf16x3", **kw))          # deterministic


def test_wide_kernel_is_the_one_that_ran(wide):
    """The dispatcher takes the wide kernel only for the shapes it serves; K % 4 != 0, an h panel, towers and other scaler
    counts stay on the 16x16 kernel (same results to summation-order noise either way)."""
    ops = wide
    assert ops.x3w_supported(300, 75, 3, 0) and ops.x3w_supported(512, 128, 3, 0)
    assert not ops.x3w_supported(300, 75, 2, 0) and not ops.x3w_supported(300, 75, 3, 75) and not ops.x3w_supported(301, 75, 3, 0)
    assert not ops.x3w_supported(300, 96, 3, 0) and not ops.x3w_supported(300, 40, 3, 0)
    a, W, b, scales, _ = _inputs(3000, 300, 75, 3)
    yw = ops.posttrans(a, 300, W, scales, b, arith="bf16x3")
    ops.X3_WIDE = False
    yn = ops.posttrans(a, 300, W, scales, b, arith="bf16x3")
    assert not torch.equal(yw, yn)                           # different k order: not bit-identical ...
    assert (yw - yn).abs().max().item() <= 2e-5 * yn.abs().max().item()          # ... and equal to fp32 noise


@pytest.mark.parametrize("post,bn,relu,resid", [(p, b, r, s) for p in (0, 1) for b in (0, 1) for r in (0, 1, 2) for s in (0, 1)])
def test_wide_epilogue_every_flag_combination(wide, post, bn, relu, resid):
    ops = wide
    M, K, N = 1500, 300, 75
    a, W, b, scales, tail = _inputs(M, K, N, 77)
    kw = {}
    if post:
        kw["row_post"] = tail["row_post"]
    if bn:
        kw["col_scale"], kw["col_shift"] = tail["col_scale"], tail["col_shift"]
    if relu == 1:
        kw["relu"] = True
    if relu == 2:
        kw["leaky_slope"] = 0.01
    if resid:
        kw["residual"] = tail["residual"]
    y = ops.posttrans(a, K, W, scales, b, arith="bf16x3", **kw)
    y32 = ops.posttrans(a, K, W, scales, b, arith="f32", **kw)
    assert (y - y32).abs().max().item() <= 2e-5 * max(1.0, y32.abs().max().item())


def test_wide_single_infinity_keeps_the_fp32_pattern(wide):
    """+-Inf / NaN elements of `a` and an infinite weight: the fp32 kernel's exact Inf / NaN pattern, signs included -- also
    in the column evaluated on the VALU (output column 74 of scaler 2) and in the LDS-combined columns 64..74."""
    ops = wide
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(15)
    M, K, N, S = 700, 300, 75, 3
    a = torch.randn(M, K, generator=gen)
    a[5, 17], a[300, 299], a[650, 0], a[651, 100] = float("inf"), float("-inf"), float("inf"), float("nan")
    a[420, 8], a[420, 200] = float("inf"), float("-inf")                       # Inf - Inf inside one row
    a = a.to(dev)
    W = torch.randn(N, S * K, generator=gen) / 30
    W[:, 17] = 0.5                                                            # bf16-exact weights under the +Inf of row 5
    W[3, 299], W[9, 0], W[74, 2 * K + 17] = 0.0, 0.0, 0.0                     # Inf * 0 = NaN in fp32 as well (incl. the VALU column)
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


def test_wide_row_permutation_equivariance(wide):
    """A row's result does not depend on where the row sits (which wavefront, lane or tile gets it): bit-identical."""
    ops = wide
    M, K, N = 20000, 300, 75
    a, W, b, scales, tail = _inputs(M, K, N, 5)
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(1)).to(a.device)
    y = ops.posttrans(a, K, W, scales, b, arith="bf16x3", row_post=tail["row_post"], residual=tail["residual"], relu=True)
    yp = ops.posttrans(a[perm].contiguous(), K, W, [None if s is None else s[perm].contiguous() for s in scales], b, arith="bf16x3",
                       row_post=tail["row_post"][perm].contiguous(), residual=tail["residual"][perm].contiguous(), relu=True)
    assert torch.equal(y[perm], yp)


def test_wide_full_size_c3_rows_vs_float64(wide):
    """The C3 shape (1 M rows x 900 -> 75) through the wide kernel: sampled rows against float64."""
    ops = wide
    dev = torch.device("cuda:0")
    M, K, N = 1_000_000, 300, 75
    gen = torch.Generator(device=dev).manual_seed(3)
    a = torch.randn(M, K, device=dev, generator=gen)
    W = torch.randn(N, 3 * K, device=dev, generator=gen) / 30
    b = torch.randn(N, device=dev, generator=gen)
    scales = [None, torch.rand(M, device=dev, generator=gen) + 0.5, torch.rand(M, device=dev, generator=gen) + 0.5]
    y = ops.posttrans(a, K, W, scales, b, arith="bf16x3")
    rows = torch.cat([torch.arange(0, 300, device=dev), torch.randint(0, M, (3000,), device=dev, generator=gen), torch.arange(M - 300, M, device=dev)])
    y64, mass = _ref64(a[rows], K, W, [None if s is None else s[rows] for s in scales], b, {})
    assert ((y[rows].double() - y64).abs() / mass).max().item() <= 5e-7
