"""Degree-grouped posttrans (pna_amd/degree_groups.py): for the rows of one in-degree the three scaler blocks of the posttrans
weight collapse into one combined weight -- same result as the ordinary three-block path to fp32 noise, every row written
exactly once, the plan's bookkeeping consistent."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _layer(F, N, dev, residual=True, seed=0, scalers="identity amplification attenuation"):
    from pna_amd.dgl.pna_layer import PNASimpleLayer
    torch.manual_seed(seed)
    layer = PNASimpleLayer(F, N, "mean max min std", scalers, {"log": torch.tensor(2.3)}, 0.0, True, residual)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(torch.randn_like(p) / (p.shape[-1] ** 0.5 if p.dim() == 2 else 3.0))
        layer.batchnorm_h.running_mean.normal_()
        layer.batchnorm_h.running_var.uniform_(0.5, 2.0)
    return layer.to(dev).eval()


# (F = 128, 96: the 128-column block -- one workgroup per CU, the rest rows through the ordinary kernel and an index scatter)
@pytest.mark.parametrize("V,E,F", [(200_000, 2_000_000, 75), (140_000, 700_000, 80), (300_000, 4_000_000, 66), (200_000, 2_000_000, 128),
                                   (150_000, 1_200_000, 96)])
def test_grouped_layer_equals_plain_layer(cuda_device, V, E, F):
    from pna_amd import Graph, degree_groups as DG
    from pna_amd.synth import powerlaw_graph
    src, dst = powerlaw_graph(V, E, seed=V % 97, device=cuda_device)
    g = Graph(src, dst, V)
    layer = _layer(F, F, cuda_device)
    h = torch.randn(V, F, device=cuda_device)
    with torch.no_grad():
        assert layer._degree_grouped_path(g, h)
        y_grouped = layer(g, h)
        for _ in range(3):                                   # run-to-run identical bits (two workgroups per CU share LDS-DMA'd weights)
            assert torch.equal(layer(g, h), y_grouped)
        keep, DG.ENABLED = DG.ENABLED, False
        try:
            assert not layer._degree_grouped_path(g, h)
            y_plain = layer(g, h)
        finally:
            DG.ENABLED = keep
    assert torch.isfinite(y_grouped).all()
    scale = y_plain.abs().max().item()
    assert (y_grouped - y_plain).abs().max().item() <= 2e-6 * scale          # (measured: ~1e-7; the combined weights are rounded once)
    # the plan: every node exactly once, groups are whole tiles of one degree, hub rows in the compacted rest
    plan = DG.plan_of(g)
    deg = g.in_degrees()
    covered = torch.cat([plan.perm[plan.perm >= 0], plan.perm_rest[plan.perm_rest >= 0]]).long()
    assert covered.numel() == V and torch.equal(torch.sort(covered).values, torch.arange(V, device=cuda_device))
    assert plan.NV % DG.TILE == 0 and plan.NRp % DG.TILE_REST == 0 and plan.tile_image.numel() == plan.NV // DG.TILE
    tiles = plan.perm.view(-1, DG.TILE)
    d = torch.where(tiles >= 0, deg[tiles.clamp(min=0).long()], torch.full_like(tiles, -1, dtype=torch.long))
    dmax = d.max(dim=1).values
    assert ((d == dmax[:, None]) | (d < 0)).all()                                 # one degree per tile
    assert torch.equal(dmax, plan.group_degree[plan.tile_image.long()])
    assert (deg[plan.perm_rest[plan.perm_rest >= 0].long()] > g.heavy_schedule().threshold).sum() == g.heavy_schedule().n_heavy


def test_grouped_layer_rows_vs_float64(cuda_device):
    """Sampled rows (frequent degrees, rare degrees, hubs) of the grouped layer against a float64 restatement."""
    from pna_amd import Graph
    from pna_amd.synth import powerlaw_graph
    V, E, F = 250_000, 2_500_000, 75
    src, dst = powerlaw_graph(V, E, seed=5, device=cuda_device)
    g = Graph(src, dst, V)
    layer = _layer(F, F, cuda_device, seed=3)
    h = torch.randn(V, F, device=cuda_device)
    with torch.no_grad():
        assert layer._degree_grouped_path(g, h)
        y = layer(g, h)
    deg = g.in_degrees()
    rows = torch.cat([torch.arange(0, 2000, device=cuda_device), torch.topk(deg, 50).indices, torch.nonzero(deg == 0).flatten()[:20]])
    csr = g.csr
    amp, att = g.degree_scalers(2.3)
    lin, bn = layer.posttrans.fully_connected[0].linear, layer.batchnorm_h
    W, b = lin.weight.double(), lin.bias.double()
    worst = 0.0
    for v in rows.tolist():
        lo, hi = int(csr.rowptr[v]), int(csr.rowptr[v + 1])
        if hi > lo:
            m = h[csr.col[lo:hi].long()].double()
            a = torch.cat([m.mean(0), m.max(0).values, m.min(0).values, torch.sqrt(torch.relu((m * m).mean(0) - m.mean(0) ** 2) + 1e-5)])
        else:
            a = torch.zeros(4 * F, dtype=torch.float64, device=cuda_device)
        z = b + W @ torch.cat([a, a * amp[v].double(), a * att[v].double()])
        z = (z - bn.running_mean.double()) / torch.sqrt(bn.running_var.double() + bn.eps) * bn.weight.double() + bn.bias.double()
        ref = h[v].double() + torch.relu(z)
        worst = max(worst, (y[v].double() - ref).abs().max().item() / max(1.0, ref.abs().max().item()))
    assert worst <= 2e-5, worst


@pytest.mark.parametrize("F,N,scalers,residual", [
    (32, 32, "identity amplification attenuation", True),          # narrow outputs: still the 80-column block (predicated columns)
    (50, 50, "identity amplification attenuation", True),          # (32 and the two-scaler 75 are off by default -- no gain
                                                                   #  measured -- and enabled here for the code path)
    (75, 75, "identity amplification", True),                      # two scalers: W_D = W_0 + amp(D) W_1, rest rows by index scatter
    (75, 75, "amplification attenuation", True),                   # no identity block
    (40, 100, "identity amplification attenuation", False),        # in_dim != out_dim (no residual), the 128-column block
    (90, 70, "identity attenuation", False),
])
def test_grouped_layer_other_shapes(cuda_device, F, N, scalers, residual):
    from pna_amd import Graph, degree_groups as DG
    from pna_amd.synth import powerlaw_graph
    V, E = 160_000, 1_500_000
    src, dst = powerlaw_graph(V, E, seed=F + N, device=cuda_device)
    g = Graph(src, dst, V)
    layer = _layer(F, N, cuda_device, residual=residual, seed=N, scalers=scalers)
    h = torch.randn(V, F, device=cuda_device)
    keep = (DG.ENABLED, DG.MIN_OUT, DG.TWO_SCALER_MIN_OUT)
    DG.MIN_OUT, DG.TWO_SCALER_MIN_OUT = 1, 1
    try:
        with torch.no_grad():
            assert layer._degree_grouped_path(g, h)
            y_grouped = layer(g, h)
            DG.ENABLED = False
            y_plain = layer(g, h)
    finally:
        DG.ENABLED, DG.MIN_OUT, DG.TWO_SCALER_MIN_OUT = keep
    assert y_grouped.shape == (V, N) and torch.isfinite(y_grouped).all()
    assert (y_grouped - y_plain).abs().max().item() <= 2e-6 * y_plain.abs().max().item()


def test_grouping_is_on_where_it_was_measured_to_pay(cuda_device):
    from pna_amd import degree_groups as DG
    from pna_amd import Graph
    from pna_amd.synth import powerlaw_graph
    V = DG.MIN_ROWS
    src, dst = powerlaw_graph(V, 4 * V, seed=1, device=cuda_device)
    g = Graph(src, dst, V)
    aggr = ("mean", "max", "min", "std")
    assert DG.applies(g, V, 75, 3, aggr) and DG.applies(g, V, 128, 3, aggr) and DG.applies(g, V, 128, 2, aggr) and DG.applies(g, V, 50, 3, aggr)
    assert not DG.applies(g, V, 32, 3, aggr) and not DG.applies(g, V, 75, 2, aggr) and not DG.applies(g, V, 75, 1, aggr)
    assert not DG.applies(g, V - 1, 75, 3, aggr) and not DG.applies(g, V, 75, 3, ("mean", "max")) and not DG.applies(g, V, 129, 3, aggr)


@pytest.mark.parametrize("path", ["degree-grouped", "ordinary"])
@pytest.mark.parametrize("name", __import__("conftest").golden_names("dgl_simple_groups"))
def test_grouped_layer_vs_reference_golden(cuda_device, name, path):
    """The REFERENCE's own output (models/dgl/pna_layer.py::PNASimpleLayer run by oracle/make_golden_degree_groups.py) on graphs
    where degree tiles exist: through the grouped path (thresholds lowered so that it takes these few thousand rows) and through
    the ordinary one."""
    from conftest import load_golden
    from pna_amd import Graph, degree_groups as DG, functional as PF
    from pna_amd.dgl.pna_layer import PNASimpleLayer
    meta, a, sd = load_golden(name)
    layer = PNASimpleLayer(meta["F"], meta["out_dim"], meta["aggregators"], meta["scalers"], {"log": a["avg_log"]}, 0.0, True, meta["residual"])
    layer.load_state_dict(sd)
    layer = layer.to(cuda_device).eval()
    g = Graph(a["src"].long(), a["dst"].long(), meta["N"]).to(cuda_device)
    h = a["h"].to(cuda_device)
    keep = (DG.ENABLED, DG.MIN_ROWS, DG.MIN_OUT, DG.TWO_SCALER_MIN_OUT, PF.SMALL_SIMPLE_ROWS)
    PF.SMALL_SIMPLE_ROWS = 0
    DG.ENABLED, DG.MIN_ROWS, DG.MIN_OUT, DG.TWO_SCALER_MIN_OUT = path == "degree-grouped", 1, 1, 1
    try:
        with torch.no_grad():
            assert layer._degree_grouped_path(g, h) == (path == "degree-grouped")
            out = layer(g, h).cpu()
        if path == "degree-grouped":
            plan = DG.plan_of(g)
            assert plan.G >= meta["degrees_with_128_rows"] - 1 and plan.G > 0 and plan.NR > 0
    finally:
        DG.ENABLED, DG.MIN_ROWS, DG.MIN_OUT, DG.TWO_SCALER_MIN_OUT, PF.SMALL_SIMPLE_ROWS = keep
    torch.testing.assert_close(out, a["out"], rtol=1e-5, atol=1e-5)


def test_grouped_layer_is_deterministic_at_full_size(cuda_device):
    """The C3 shape, ten runs of the degree-grouped layer: identical bits (an experimental fused kernel with the same weight
    pipeline was NOT, with two workgroups per CU -- DESIGN.md 4.7 point 7; the shipped kernels are checked for the same symptom)."""
    from pna_amd import Graph
    from pna_amd.synth import powerlaw_graph
    V, E, F = 1_000_000, 10_000_000, 75
    src, dst = powerlaw_graph(V, E, seed=1234, device=cuda_device)
    g = Graph(src, dst, V)
    layer = _layer(F, F, cuda_device, seed=7)
    h = torch.randn(V, 80, device=cuda_device)[:, :F]
    with torch.no_grad():
        assert layer._degree_grouped_path(g, h)
        y0 = layer(g, h).clone()
        for _ in range(10):
            assert torch.equal(layer(g, h), y0)


def test_grouped_path_on_a_graph_without_any_group(cuda_device):
    """No degree value fills a tile: the grouping would be pure overhead (every row in the rest list), so the layer takes the
    ordinary path; the plan itself is still well-formed -- all rest rows -- and its two-kernel path still computes the
    layer when called directly."""
    from pna_amd import Graph, degree_groups as DG
    from pna_amd.synth import powerlaw_graph
    V, E, F = 700, 7000, 75
    src, dst = powerlaw_graph(V, E, seed=2, device=cuda_device)
    g = Graph(src, dst, V)
    layer = _layer(F, F, cuda_device, seed=1)
    h = torch.randn(V, F, device=cuda_device)
    keep = (DG.ENABLED, DG.MIN_ROWS)
    from pna_amd import functional as PF
    keep_small, PF.SMALL_SIMPLE_ROWS = PF.SMALL_SIMPLE_ROWS, 0
    try:
        with torch.no_grad():
            DG.ENABLED, DG.MIN_ROWS = True, 1
            plan = DG.plan_of(g)
            assert plan.G == 0 and plan.NR == V and not layer._degree_grouped_path(g, h)
            y_grouped = PF.degree_grouped_posttrans(layer, g, h, PF.degree_grouped_aggregate(layer, g, h, plan), plan)
            DG.ENABLED = False
            y_plain = layer(g, h)
    finally:
        DG.ENABLED, DG.MIN_ROWS = keep
        PF.SMALL_SIMPLE_ROWS = keep_small
    assert (y_grouped - y_plain).abs().max().item() <= 2e-6 * y_plain.abs().max().item()


def _pitched(x, mult=4):
    """The same values in rows of a pitch that is a multiple of `mult` floats (what the one-kernel paths read in 16-byte strips)."""
    P = (x.shape[1] + mult - 1) // mult * mult
    buf = torch.full((x.shape[0], P), float("nan"), dtype=x.dtype, device=x.device)
    buf[:, :x.shape[1]] = x
    return buf[:, :x.shape[1]]


@pytest.mark.parametrize("path", ["one-kernel", "degree-grouped", "ordinary"])
@pytest.mark.parametrize("name", __import__("conftest").golden_names("dgl_tower_groups"))
def test_grouped_tower_layer_vs_reference_golden(cuda_device, name, path):
    """The REFERENCE's own PNALayer output (towers, graph norm, BatchNorm, mixing network; oracle/make_golden_degree_groups.py) on
    graphs with degree tiles: through the degree-grouped tower path -- gather in degree order, ONE contraction with the collapsed
    posttrans . BatchNorm . mixing weight (functional.tower_layer_degree_grouped) -- and through the ordinary kernels."""
    from conftest import load_golden
    from pna_amd import Graph, degree_groups as DG, functional as PF
    from pna_amd.dgl.pna_layer import PNALayer
    meta, a, sd = load_golden(name)
    layer = PNALayer(meta["in_dim"], meta["out_dim"], meta["aggregators"], meta["scalers"], {"log": a["avg_log"]}, 0.0, meta["graph_norm"],
                     meta["batch_norm"], towers=meta["towers"], pretrans_layers=1, posttrans_layers=1, divide_input=meta["divide_input"],
                     residual=meta["residual"], edge_features=False, edge_dim=0)
    layer.load_state_dict(sd)
    layer = layer.to(cuda_device).eval()
    g = Graph(a["src"].long(), a["dst"].long(), meta["N"], meta["sizes"]).to(cuda_device)
    h, snorm = a["h"].to(cuda_device), a["snorm_n"].to(cuda_device)
    one = path == "one-kernel"
    if one and not (49 <= meta["in_dim"] <= 80 and meta["out_dim"] <= 80):
        # (one tower, T towers with divide_input -- ONE gather of in_dim message features -- or, round 6, T towers over the whole input:
        # one launch per tower, FusedMultiTowerCall)
        pytest.skip("the one-kernel tower layer takes gathers of 49..80 message features")
    if one:
        h = _pitched(h)                                   # (pitch 75 rows are not 16-byte aligned: the path would decline)
    keep = (DG.ENABLED, DG.MIN_ROWS, DG.MIN_OUT, PF.SMALL_TOWER_ROWS, DG.FUSED)
    DG.ENABLED, DG.MIN_ROWS, DG.MIN_OUT, PF.SMALL_TOWER_ROWS, DG.FUSED = path != "ordinary", 1, 1, 0, one
    try:
        with torch.no_grad():
            assert PF.tower_layer_degree_grouped_applies(layer, g, h) == (path != "ordinary")
            if path != "ordinary":
                assert PF.tower_layer_degree_fused_applies(layer, g, h) == one
            out = layer(g, h, None, snorm).cpu()
        if path != "ordinary":
            plan = DG.plan_of(g)
            assert plan.G > 0 and plan.NR > 0
    finally:
        DG.ENABLED, DG.MIN_ROWS, DG.MIN_OUT, PF.SMALL_TOWER_ROWS, DG.FUSED = keep
    # 1e-5 relative + 1e-5 absolute, plus -- for the few elements where a row's terms cancel -- 4 x the REFERENCE's own fp32 error on
    # that row (its fp32 output against the oracle's float64 evaluation of the same formulas): the collapsed weight rounds the two
    # Linears' product once where the reference rounds twice, neither is closer to the exact value
    from oracle import torch_oracle as O
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    ref64 = O.dgl_layer_forward(sd64, a["src"].long(), a["dst"].long(), meta["N"], a["h"].double(), torch.zeros(a["src"].numel(), 0, dtype=torch.float64),
                                a["snorm_n"].double(), meta["aggregators"].split(), meta["scalers"].split(), a["avg_log"].double(), meta["towers"],
                                meta["divide_input"], meta["graph_norm"], meta["batch_norm"], meta["residual"], False)
    floor = 4.0 * (a["out"].double() - ref64).abs().max(dim=1, keepdim=True).values
    err = (out.double() - a["out"].double()).abs()
    plain = 1e-5 * a["out"].double().abs() + 1e-5
    assert bool((err <= plain + floor).all()), float((err - plain - floor).max())
    assert (err <= plain).double().mean().item() >= 0.999


@pytest.mark.parametrize("T,Fi,out,divide", [(1, 75, 75, False), (5, 75, 75, False), (4, 16, 64, True)])
def test_grouped_tower_layer_equals_ordinary_kernels(cuda_device, T, Fi, out, divide):
    from pna_amd import Graph, degree_groups as DG, functional as PF
    from pna_amd.dgl.pna_layer import PNALayer
    from pna_amd.synth import powerlaw_graph
    V, E = 150_000, 1_200_000
    src, dst = powerlaw_graph(V, E, seed=T + Fi, device=cuda_device)
    sizes = [V // 2, V - V // 2]
    g = Graph(src, dst, V, sizes)
    in_dim = T * Fi if divide else Fi
    torch.manual_seed(T)
    layer = PNALayer(in_dim, out, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.2)}, 0.0, True, True,
                     towers=T, divide_input=divide, residual=in_dim == out).to(cuda_device).eval()
    with torch.no_grad():
        for t in layer.towers:
            t.batchnorm_h.running_mean.normal_()
            t.batchnorm_h.running_var.uniform_(0.5, 2.0)
    h = torch.randn(V, in_dim, device=cuda_device)
    snorm = g.snorm_n()
    keep = (DG.ENABLED, DG.MIN_ROWS, DG.MIN_OUT, PF.SMALL_TOWER_ROWS)
    DG.MIN_ROWS, DG.MIN_OUT, PF.SMALL_TOWER_ROWS = 1, 1, 0
    try:
        with torch.no_grad():
            DG.ENABLED = True
            assert PF.tower_layer_degree_grouped_applies(layer, g, h)
            y_g = layer(g, h, None, snorm)
            assert torch.equal(layer(g, h, None, snorm), y_g)
            DG.ENABLED = False
            y_p = layer(g, h, None, snorm)
    finally:
        DG.ENABLED, DG.MIN_ROWS, DG.MIN_OUT, PF.SMALL_TOWER_ROWS = keep
    assert torch.isfinite(y_g).all()
    assert (y_g - y_p).abs().max().item() <= 1e-5 * y_p.abs().max().item()       # (two Linears collapsed into one: rounded differently)


@pytest.mark.parametrize("Fi,out,gn,bn,res", [(75, 75, True, True, True), (64, 64, False, True, True), (80, 48, True, False, False),
                                              (49, 80, True, True, False), (56, 20, False, False, False), (72, 72, True, True, True)])
def test_one_kernel_tower_layer_equals_two_kernel_grouped_path(cuda_device, Fi, out, gn, bn, res):
    """pna_fused_degree_f32 in tower mode (gather over x_src, destination term and self features as extra K panels) against the
    two-kernel grouped path (gather with the destination term added per edge -> aggregate in HBM -> grouped contraction), which
    the reference goldens pin: same collapsed weight, fp32-level arithmetic in both (fp16 x 2 in the one-kernel layer since round 5, bf16x3 in the two-kernel path); they differ in WHERE x_dst is added (after the
    statistics instead of per edge) and in the summation order over K.  Every shape class: half block or not, partial column
    windows, with / without graph norm, BatchNorm, residual; isolated nodes and hub rows are in the graph."""
    from pna_amd import Graph, degree_groups as DG, functional as PF
    from pna_amd.dgl.pna_layer import PNALayer
    from pna_amd.synth import powerlaw_graph
    V, E = 120_000, 900_000
    src, dst = powerlaw_graph(V, E, seed=Fi + out, device=cuda_device)
    keep_e = dst >= 500                                    # nodes 0..499 lose their in-edges: rows without any message
    g = Graph(src[keep_e], dst[keep_e], V, [V // 2, V - V // 2])
    torch.manual_seed(Fi)
    layer = PNALayer(Fi, out, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.2)}, 0.0, gn, bn,
                     towers=1, divide_input=False, residual=res and Fi == out).to(cuda_device).eval()
    with torch.no_grad():
        if bn:
            layer.towers[0].batchnorm_h.running_mean.normal_()
            layer.towers[0].batchnorm_h.running_var.uniform_(0.5, 2.0)
    h = _pitched(torch.randn(V, Fi, device=cuda_device), 8)      # (a full last block is read in 32-byte strips: pitch >= round_up(F, 8))
    snorm = g.snorm_n()
    keep = (DG.ENABLED, DG.MIN_ROWS, DG.MIN_OUT, PF.SMALL_TOWER_ROWS, DG.FUSED)
    DG.ENABLED, DG.MIN_ROWS, DG.MIN_OUT, PF.SMALL_TOWER_ROWS = True, 1, 1, 0
    try:
        with torch.no_grad():
            DG.FUSED = True
            assert PF.tower_layer_degree_fused_applies(layer, g, h)
            y_f = layer(g, h, None, snorm)
            assert torch.equal(layer(g, h, None, snorm), y_f)
            DG.FUSED = False
            assert not PF.tower_layer_degree_fused_applies(layer, g, h) and PF.tower_layer_degree_grouped_applies(layer, g, h)
            y_g = layer(g, h, None, snorm)
    finally:
        DG.ENABLED, DG.MIN_ROWS, DG.MIN_OUT, PF.SMALL_TOWER_ROWS, DG.FUSED = keep
    plan = DG.plan_of(g)
    assert plan.G > 0 and plan.NR > 0 and int((g.in_degrees() == 0).sum()) >= 500
    assert torch.isfinite(y_f).all()
    # per element: 1e-5 relative + 2e-6 of the row's largest output (cancelling terms)
    tol = 1e-5 * y_g.abs() + 2e-6 * y_g.abs().max(dim=1, keepdim=True).values + 1e-6
    bad = (y_f - y_g).abs() > tol
    assert not bool(bad.any()), (int(bad.sum()), float(((y_f - y_g).abs() / tol).max()))
