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


def test_bf16x3_rejects_more_than_three_scalers():
    from pna_amd import ops
    dev = torch.device("cuda:0")
    a, W = torch.randn(8, 16, device=dev), torch.randn(4, 64, device=dev)
    with pytest.raises(ValueError, match="at most 3 scalers"):
        ops.posttrans(a, 16, W, [None] * 4, arith="bf16x3")


def _force_x3(monkeypatch):
    from pna_amd import ops
    monkeypatch.setattr(ops, "POSTTRANS_ARITH", "bf16x3")


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
