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


def _scaled_case(M, K, N, S, self_block, with_beta=True, ldy=None, seed=0):
    from pna_amd import ops
    g = torch.Generator().manual_seed(seed)
    B = S + int(self_block)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, B * K, generator=g) * 0.3
    scales = torch.rand(M, S, generator=g) * 3 if S else None
    if S and M > 3:
        scales[3] = 0.0                                           # a row without in-edges
    beta = torch.randn(S, N, generator=g) if (S and with_beta) else None
    ldy = ldy or N
    yb = torch.full((M, ldy), 7.0, device="cuda")
    y = ops.project_scaled(x.cuda(), K, w.cuda(), None if scales is None else scales.cuda(), None if beta is None else beta.cuda(), self_block, out=yb[:, :N])
    xd, wd = x.double(), w.double()
    ref = torch.zeros(M, N, dtype=torch.float64)
    floor = torch.zeros(M, N, dtype=torch.float64)
    b0 = 0
    if self_block:
        ref += xd @ wd[:, :K].t(); floor += xd.abs() @ wd[:, :K].abs().t(); b0 = 1
    for s in range(S):
        blk = wd[:, (b0 + s) * K:(b0 + s + 1) * K]
        t = xd @ blk.t() + (beta[s].double() if beta is not None else 0.0)
        ref += scales[:, s:s + 1].double() * t
        floor += scales[:, s:s + 1].double() * (xd.abs() @ blk.abs().t() + (beta[s].abs().double() if beta is not None else 0.0))
    err = ((y.cpu().double() - ref).abs() / floor.clamp(min=1e-300)).max().item() if M else 0.0
    assert err <= 1e-6, (M, K, N, S, self_block, err)
    assert (yb[:, N:] == 7.0).all()


@pytest.mark.parametrize("M,K,N,S,self_block", [(1000, 75, 75, 3, True), (4097, 75, 75, 3, False), (130, 75, 80, 2, True), (17, 16, 16, 1, False),
                                                (300, 128, 64, 2, True), (255, 100, 33, 3, False), (513, 80, 80, 3, True), (64, 5, 3, 1, True),
                                                (200, 40, 50, 0, True), (0, 75, 75, 3, True), (1, 4, 1, 3, True)])
def test_scaled_projection_matches_float64(M, K, N, S, self_block):
    _scaled_case(M, K, N, S, self_block)


def test_scaled_projection_without_beta_and_into_a_pitched_buffer():
    _scaled_case(777, 75, 75, 3, True, with_beta=False, ldy=80)


def test_scaled_projection_rejects_four_blocks_beyond_its_register_budget():
    from pna_amd import ops
    x = torch.randn(8, 96, device="cuda")
    assert not ops.project_scaled_applies(x, 96, 75, 4) and ops.project_scaled_applies(x[:, :75], 75, 75, 4)
    with pytest.raises(Exception):
        ops.project_scaled(x, 96, torch.randn(75, 4 * 96, device="cuda"), torch.rand(8, 3, device="cuda"), None, True)


@pytest.mark.parametrize("V,K,N,G,seed", [(5000, 75, 300, 7, 0), (1000, 16, 80, 3, 1), (3000, 75, 75, 1, 2), (777, 33, 130, 12, 3), (20000, 75, 300, 40, 4)])
def test_grouped_projection_matches_float64(V, K, N, G, seed):
    """pna_project_grouped_f32: virtual rows in 128-row tiles, each tile with its group's weight, -1 padding rows, rows that no virtual
    row names left alone, tiles listed in group order or as they come."""
    from pna_amd import ops
    g = torch.Generator().manual_seed(seed)
    ntiles = (V + 127) // 128 + 3
    tile_group = torch.randint(0, G, (ntiles,), generator=g, dtype=torch.int32)
    perm = torch.full((ntiles * 128,), -1, dtype=torch.int32)
    named = torch.randperm(V, generator=g)[:V - V // 10]                       # a tenth of the nodes are named by no virtual row
    slots = torch.randperm(ntiles * 128, generator=g)[:named.numel()]
    perm[slots] = named.to(torch.int32)
    x = torch.randn(V, K, generator=g)
    w = torch.randn(G, N, K, generator=g) * 0.3
    order = torch.sort(tile_group.long(), stable=True).indices
    for by_group in (True, False):                                               # tiles listed sorted by group, or as they come
        pm, tg = (perm.view(-1, 128)[order].reshape(-1), tile_group[order]) if by_group else (perm, tile_group)
        y = torch.full((V, N + 5), 7.0, device="cuda")
        ops.project_grouped(x.cuda(), K, w.cuda(), pm.contiguous().cuda(), tg.contiguous().cuda(), out=y[:, :N])
        y = y.cpu()
        grp = torch.full((V,), -1, dtype=torch.long)
        grp[perm[slots].long()] = tile_group[(slots // 128)].long()
        ref = torch.einsum("vk,vnk->vn", x[named].double(), w[grp[named]].double())
        floor = torch.einsum("vk,vnk->vn", x[named].abs().double(), w[grp[named]].abs().double())
        err = ((y[named, :N].double() - ref).abs() / floor).max().item()
        assert err <= 1e-6, err
        untouched = torch.ones(V, dtype=torch.bool); untouched[named] = False
        assert (y[untouched] == 7.0).all() and (y[:, N:] == 7.0).all()
