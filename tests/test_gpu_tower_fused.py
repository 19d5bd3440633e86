"""pna_small_linear_f32 / pna_tower_layer_f32 (the one-call tower layer of molecule-sized batches, pna_tower_fused.hip) against
torch in float64 and against the large-graph kernels on the same inputs: every shape class the kernel dispatches on (K split
over 1 / 2 / 4 / 8 wavefronts, several 16-column tiles per tower, more units than wavefronts, tower groups when the aggregate
tile does not fit the LDS at once, divided input), rows without in-edges, a hub row with more edges than the LDS id cache,
row counts that are not a multiple of 16."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

AGG, SCA = "mean max min std", "identity amplification attenuation"


@pytest.mark.parametrize("M,K,N,act,res", [(1, 75, 750, 0, False), (3000, 75, 750, 0, False), (777, 70, 70, 2, True), (100, 5, 3, 1, False),
                                           (4097, 300, 20, 2, True), (33, 1000, 17, 0, True)])
def test_small_linear_matches_float64(M, K, N, act, res):
    from pna_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K + 3, generator=gen).to(dev)[:, :K]            # non-contiguous rows
    W = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    r = torch.randn(M, N, generator=gen).to(dev) if res else None
    buf = torch.full((M, N + 5), 7.0, device=dev)
    out = buf[:, 2:2 + N]
    ops.small_linear(x, ops.pack_small(W), N, b, act=act, slope=0.01, residual=r, out=out)
    z = x.double() @ W.double().t() + b.double()
    if act == 1:
        z = torch.relu(z)
    elif act == 2:
        z = torch.nn.functional.leaky_relu(z, 0.01)
    if res:
        z = z + r.double()
    assert (out.double() - z).abs().max().item() <= 2e-6 * max(1.0, z.abs().max().item()) * max(1.0, K / 100)
    assert (buf[:, :2] == 7.0).all() and (buf[:, 2 + N:] == 7.0).all()


def _graph(V, E, seed, hub=None, isolated=()):
    from pna_amd import Graph
    rng = np.random.default_rng(seed)
    src = rng.integers(0, V, E)
    dst = rng.integers(0, V, E)
    if hub is not None:                                             # one row with more in-edges than the kernel's LDS id cache
        row, deg = hub
        src = np.concatenate([src, rng.integers(0, V, deg)])
        dst = np.concatenate([dst, np.full(deg, row)])
    keep = ~np.isin(dst, np.asarray(isolated, dtype=np.int64))
    src, dst = src[keep], dst[keep]
    return Graph(torch.from_numpy(src), torch.from_numpy(dst), V, [V])


def _layer(in_dim, out_dim, towers, divide_input, scalers, graph_norm, batch_norm, residual, avg, seed):
    from pna_amd.dgl.pna_layer import PNALayer
    layer = PNALayer(in_dim, out_dim, AGG, scalers, avg, 0.0, graph_norm, batch_norm, towers=towers, divide_input=divide_input,
                     residual=residual).eval()
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            p.copy_(torch.randn(p.shape, generator=gen) / (p.shape[-1] ** 0.5 if p.dim() == 2 else 3.0) +
                    (1.0 if "batchnorm" in n and n.endswith("weight") else 0.0))
        for n, bfr in layer.named_buffers():
            if n.endswith("running_mean"):
                bfr.copy_(torch.randn(bfr.shape, generator=gen) * 0.2)
            elif n.endswith("running_var"):
                bfr.copy_(torch.rand(bfr.shape, generator=gen) + 0.5)
    return layer


CASES = [  # V, E, in, out, towers, divide_input, scalers, graph_norm, batch_norm, residual, hub, isolated
    (2941, 6300, 75, 75, 5, False, SCA, True, True, True, None, (0, 17, 2940)),     # the ZINC layer (configs[1])
    (2941, 6300, 75, 70, 5, True, SCA, True, True, False, None, ()),                # ... its last layer (divided input, Fo = 14)
    (1000, 3000, 20, 20, 1, False, SCA, False, True, True, None, ()),               # one tower: K split over 4 wavefronts, 2 tiles
    (1000, 3000, 24, 16, 2, True, "identity", True, False, False, None, ()),        # 2 units x 4 wavefronts, one scaler
    (517, 4000, 30, 60, 3, False, "identity amplification", False, False, False, None, (5,)),   # Fo = 20: 6 units, two scalers
    (300, 900, 100, 70, 5, False, SCA, True, True, False, None, ()),                # MNIST last layer: 157 KB of LDS
    (300, 900, 96, 96, 6, False, SCA, False, True, True, None, ()),                 # aggregate tile > LDS: tower groups 5 + 1
    (5, 9, 75, 75, 5, False, SCA, True, True, True, None, ()),                      # a single partial tile
    (2000, 5000, 75, 75, 5, False, SCA, True, True, True, (1003, 5000), (1004,)),   # hub row past the LDS id cache
    (1, 0, 16, 16, 4, True, SCA, False, False, False, None, ()),                    # one node, no edges
]


@pytest.mark.parametrize("V,E,in_dim,out_dim,towers,divide,scalers,gn,bn,res,hub,isolated", CASES)
def test_tower_layer_small_vs_float64_and_large_graph_path(monkeypatch, V, E, in_dim, out_dim, towers, divide, scalers, gn, bn, res, hub, isolated):
    from pna_amd import functional as PF
    dev = torch.device("cuda:0")
    g = _graph(V, E, seed=V + E, hub=hub, isolated=isolated).to(dev)
    deg = g.in_degrees()
    avg = {"log": torch.log(deg.double() + 1).mean().float().cpu()}
    if float(avg["log"]) == 0.0:                 # a graph without edges: the reference's avg_d is 0 and its scalers 0 / 0;
        avg["log"] = torch.tensor(1.0)           # any finite value serves (every aggregate is 0 there)
    layer = _layer(in_dim, out_dim, towers, divide, scalers, gn, bn, res, avg, seed=E + 1).to(dev)
    h = torch.randn(V, in_dim, generator=torch.Generator().manual_seed(3)).to(dev)
    snorm = (torch.rand(V, 1, generator=torch.Generator().manual_seed(4)) + 0.5).to(dev)
    with torch.no_grad():
        assert layer._small_batch_path(g, h)
        y_small = layer(g, h, None, snorm)
        monkeypatch.setattr(PF, "SMALL_TOWER_ROWS", 0)
        assert not layer._small_batch_path(g, h)
        y_large = layer(g, h, None, snorm)
    # float64 restatement of models/dgl/pna_layer.py:35-75,:133-148 (eval mode) in torch ops
    T, Fi, Fo = towers, (in_dim // towers if divide else in_dim), out_dim // towers
    src, dst = g.csr.col.long(), g.csr.row.long()
    D = deg.double()
    hd, outs = h.double(), []
    amp = torch.log(D + 1) / avg["log"].double().to(dev)
    att = torch.where(D > 0, avg["log"].double().to(dev) / torch.log(D + 1), torch.zeros_like(D))
    for t, tower in enumerate(layer.towers):
        ht = hd[:, t * Fi:(t + 1) * Fi] if divide else hd
        lin = tower.pretrans.fully_connected[0].linear
        m = torch.cat([ht[src], ht[dst]], dim=1) @ lin.weight.double().t() + lin.bias.double()
        cnt = D.clamp(min=1)[:, None]
        mean = torch.zeros(V, Fi, dtype=torch.float64, device=dev).index_add_(0, dst, m) / cnt
        msq = torch.zeros(V, Fi, dtype=torch.float64, device=dev).index_add_(0, dst, m * m) / cnt
        mx = torch.full((V, Fi), -float("inf"), dtype=torch.float64, device=dev).scatter_reduce_(0, dst[:, None].expand(-1, Fi), m, "amax")
        mn = torch.full((V, Fi), float("inf"), dtype=torch.float64, device=dev).scatter_reduce_(0, dst[:, None].expand(-1, Fi), m, "amin")
        std = torch.sqrt(torch.relu(msq - mean * mean) + 1e-5)
        has = (D > 0)[:, None]
        a = torch.cat([torch.where(has, x, torch.zeros_like(x)) for x in (mean, mx, mn, std)], dim=1)
        blocks = [{"identity": a, "amplification": a * amp[:, None], "attenuation": a * att[:, None]}[s] for s in scalers.split()]
        lin = tower.posttrans.fully_connected[0].linear
        z = torch.cat([ht] + blocks, dim=1) @ lin.weight.double().t() + lin.bias.double()
        if gn:
            z = z * snorm.double()
        if bn:
            b_ = tower.batchnorm_h
            z = (z - b_.running_mean.double()) / torch.sqrt(b_.running_var.double() + b_.eps) * b_.weight.double() + b_.bias.double()
        outs.append(z)
    mix = layer.mixing_network
    y64 = torch.nn.functional.leaky_relu(torch.cat(outs, dim=1) @ mix.linear.weight.double().t() + mix.linear.bias.double(), 0.01)
    if layer.residual:
        y64 = hd + y64
    scale = max(1.0, y64.abs().max().item())
    e_small = (y_small.double() - y64).abs().max().item() / scale
    e_large = (y_large.double() - y64).abs().max().item() / scale
    assert e_small <= 2e-5, (e_small, e_large)
    assert e_small <= 4 * e_large + 2e-6, (e_small, e_large)             # not less accurate than the large-graph kernels
    assert torch.isfinite(y_small).all()


def test_tower_layer_small_is_deterministic_and_capturable():
    """Same bits on every call (the K-split partial tiles are summed in wavefront order) and under hipGraph replay."""
    from pna_amd.capture import GraphedForward
    dev = torch.device("cuda:0")
    g = _graph(1500, 4000, seed=9).to(dev)
    avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
    layer = _layer(20, 20, 1, False, SCA, True, True, True, avg, seed=2).to(dev)
    h = torch.randn(1500, 20, device=dev)
    sn = g.snorm_n()
    with torch.no_grad():
        y0 = layer(g, h, None, sn)
        for _ in range(5):
            assert torch.equal(layer(g, h, None, sn), y0)
        gf = GraphedForward(lambda x: layer(g, x, None, sn), h)
        assert torch.equal(gf(h), y0)
        h2 = torch.randn(1500, 20, device=dev)
        assert torch.equal(gf(h2), layer(g, h2, None, sn))


def test_tower_layer_small_tracks_weight_updates():
    """The cached weight images follow in-place parameter updates (optimizer steps, load_state_dict)."""
    dev = torch.device("cuda:0")
    g = _graph(400, 1200, seed=1).to(dev)
    avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
    layer = _layer(30, 30, 5, True, SCA, True, True, True, avg, seed=5).to(dev)
    h = torch.randn(400, 30, device=dev)
    sn = g.snorm_n()
    with torch.no_grad():
        y0 = layer(g, h, None, sn)
        layer.towers[2].posttrans.fully_connected[0].linear.weight.mul_(1.5)
        layer.mixing_network.linear.bias.add_(0.25)
        layer.towers[0].batchnorm_h.running_var.mul_(2.0)
        y1 = layer(g, h, None, sn)
        assert not torch.equal(y0, y1)
        fresh = _layer(30, 30, 5, True, SCA, True, True, True, avg, seed=5).to(dev)
        fresh.load_state_dict(layer.state_dict())
        assert torch.equal(fresh(g, h, None, sn), y1)


def test_small_simple_layer_keeps_a_non_finite_own_feature_out(cuda_device, monkeypatch):
    """PNASimpleLayer's posttrans never reads the node's own h (models/dgl/pna_layer.py:206).  The one-call small-batch kernel used to
    multiply h with a ZERO self panel (0 * Inf = NaN); with `no_self_panel` it multiplies zeros: a node whose own
    feature is Inf / NaN but whose neighbours are finite gets the three-kernel path's finite output."""
    from pna_amd import Graph, functional as PF
    from pna_amd.dgl.pna_layer import PNASimpleLayer
    from pna_amd.synth import molecule_batch
    src, dst, sizes = molecule_batch(32, seed=3)
    V = int(sum(sizes))
    g = Graph(src, dst, V, sizes).to(cuda_device)
    torch.manual_seed(1)
    layer = PNASimpleLayer(40, 40, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(1.1)}, 0.0, True, False).to(cuda_device).eval()
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(torch.randn_like(p) / (p.shape[-1] ** 0.5 if p.dim() == 2 else 2.0))
    # a leaf-free pick: node 5's own features are poisoned; only rows that have node 5 as a SOURCE may turn non-finite
    h = torch.randn(V, 40, device=cuda_device)
    h[5, 3], h[5, 7] = float("inf"), float("nan")
    with torch.no_grad():
        y_small = layer(g, h)
        monkeypatch.setattr(PF, "SMALL_SIMPLE_ROWS", 0)
        y_ref = layer(g, h)
    has5 = torch.zeros(V, dtype=torch.bool, device=cuda_device)
    has5[g.csr.row.long()[g.csr.col.long() == 5]] = True
    assert not bool(has5[5]) and torch.isfinite(y_ref[5]).all()
    assert torch.isfinite(y_small[~has5]).all()
    assert torch.equal(torch.isfinite(y_small), torch.isfinite(y_ref))
    ok = torch.isfinite(y_ref)
    torch.testing.assert_close(y_small[ok], y_ref[ok], rtol=1e-5, atol=1e-5)
