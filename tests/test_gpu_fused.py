"""pna_fused_simple_f32 (the whole PNASimpleLayer inference forward in one launch) against the two-kernel path,
which the golden fixtures pin (test_gpu_layers).  Summation order differs (k permutation, hub rows split across the
workgroup), so agreement is tolerance-level: 2e-5 of the output scale, 8x that where hub rows are split."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # V, E, F, N, S, heavy_threshold
    (100, 700, 75, 75, 3, 0), (33, 200, 8, 16, 1, 0), (1000, 9000, 75, 75, 3, 64), (257, 4000, 20, 70, 2, 32),
    (64, 500, 4, 5, 3, 0), (500, 3000, 80, 80, 3, 16), (95, 1000, 33, 40, 2, 0), (1, 5, 75, 75, 3, 0),
]


@pytest.mark.parametrize("V,E,F,N,S,ht", CASES)
def test_fused_simple_layer_matches_two_kernel_path(V, E, F, N, S, ht):
    from pna_amd import Graph, functional as PF, ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(V + E)
    src = torch.randint(0, V, (E,), generator=gen)
    dst = torch.randint(0, max(1, V - 3), (E,), generator=gen)        # the last rows have no in-edges
    if ht:
        dst[: E // 3] = 1                                              # a hub row
    g = Graph(src, dst, V).to(dev)
    x = torch.randn(V, F, generator=gen).to(dev)
    W = (torch.randn(N, S * 4 * F, generator=gen) / (S * 4 * F) ** 0.5).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    scales = [None] + [(torch.rand(V, generator=gen) + 0.5).to(dev) for _ in range(S - 1)]
    cs, ct = (torch.rand(N, generator=gen) + 0.5).to(dev), torch.randn(N, generator=gen).to(dev)
    res = torch.randn(V, N, generator=gen).to(dev)
    agg = PF.aggregate(g, x, F, ["mean", "max", "min", "std"])
    want = ops.posttrans(agg, 4 * F, W, scales, b, col_scale=cs, col_shift=ct, relu=True, residual=res)
    got = ops.fused_simple(g.csr.rowptr, g.csr.col, x, F, W, scales, b, col_scale=cs, col_shift=ct, relu=True,
                           residual=res, heavy_threshold=ht)
    tol = 2e-5 * max(1.0, want.abs().max().item()) * (8 if ht else 1)
    assert (got - want).abs().max().item() <= tol


def test_fused_rejects_unsupported_shapes():
    from pna_amd import Graph, ops
    dev = torch.device("cuda:0")
    g = Graph(torch.tensor([0, 1]), torch.tensor([1, 0]), 2).to(dev)
    x = torch.randn(2, 96, device=dev)
    W = torch.randn(8, 4 * 96, device=dev)
    with pytest.raises(RuntimeError, match="supported range"):
        ops.fused_simple(g.csr.rowptr, g.csr.col, x, 96, W, [None])
