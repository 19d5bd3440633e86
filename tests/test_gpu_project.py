"""pna_project_f32 (pna_amd/csrc/pna_project.hip): the node-level source projection of all the towers of a PNALayer (reference
models/dgl/pna_layer.py:137-139, :36-44 factorised to node level) against float64 -- fp32 products and accumulation, so the bar is an
fp32 GEMM's: 1e-6 of sum_k |x_k| |w_k| (the same bar as the fp32 contraction kernels')."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(M, K, N, ldx=None, ldy=None, seed=0):
    from pna_amd import ops
    g = torch.Generator().manual_seed(seed)
    ldx, ldy = ldx or K, ldy or N
    xb = torch.randn(M, ldx, generator=g) * torch.exp(torch.randn(M, 1, generator=g) * 2)
    w = torch.randn(N, K, generator=g) * 0.3
    x = xb.cuda()[:, :K]
    yb = torch.full((M, ldy), 7.0, device="cuda")
    y = ops.project(x, K, w.cuda(), out=yb[:, :N])
    ref = xb[:, :K].double() @ w.double().t()
    floor = xb[:, :K].abs().double() @ w.abs().double().t()
    err = ((y.cpu().double() - ref).abs() / floor.clamp(min=1e-300)).max().item() if M else 0.0
    assert err <= 1e-6, (M, K, N, err)
    assert (yb[:, N:] == 7.0).all()                              # the pitch's padding is left alone
    return y


@pytest.mark.parametrize("M,K,N", [(1, 4, 1), (15, 5, 3), (17, 75, 400), (1000, 75, 400), (4097, 75, 480), (333, 16, 16), (260, 17, 33), (513, 64, 512),
                                   (300, 128, 240), (255, 127, 239), (100, 80, 320), (64, 7, 81), (5000, 50, 160), (0, 75, 400)])
def test_projection_matches_float64(M, K, N):
    _case(M, K, N)


def test_projection_reads_and_writes_pitched_rows():
    _case(777, 75, 400, ldx=80, ldy=416)
    _case(130, 13, 29, ldx=13 + 3, ldy=31)


def test_projection_is_the_library_gemm_up_to_summation_order():
    """Bitwise it differs from torch.mm only by the order of the K-sum: the same inputs twice give the same bits (no atomics), and the
    result is within fp32 rounding of the library's."""
    from pna_amd import ops
    torch.manual_seed(3)
    x, w = torch.randn(3000, 75, device="cuda"), torch.randn(400, 75, device="cuda")
    a, b = ops.project(x, 75, w), ops.project(x, 75, w)
    assert torch.equal(a, b)
    ref = x @ w.t()
    assert ((a - ref).abs() / (x.abs() @ w.abs().t())).max().item() <= 1e-6


def test_projection_rejects_what_it_cannot_hold():
    from pna_amd import ops
    x = torch.randn(8, 128, device="cuda")
    with pytest.raises(Exception):
        ops.project(x, 128, torch.randn(512, 128, device="cuda"))       # 128 x 516 floats > 160 KB of LDS
    with pytest.raises(Exception):
        ops.project(x[:, :3], 3, torch.randn(8, 3, device="cuda"))      # K < 4
