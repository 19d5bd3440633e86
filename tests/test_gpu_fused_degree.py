"""pna_fused_degree_f32 (pna_amd/csrc/pna_fused_degree.hip): PNASimpleLayer.forward (models/dgl/pna_layer.py:186-216) with gather,
aggregators, scalers and posttrans in ONE kernel on degree-ordered rows.  Parity chain: the statistics its contraction consumes
are the production gather's BITS (which are the C oracle's for every row one lane group walks alone); the layer output matches
the reference's own golden outputs (graphs with degree tiles, oracle/make_golden_degree_groups.py), the two-kernel grouped path,
the ordinary path and a float64 restatement; 200 repeats at full size give identical bits."""
import ctypes

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


def _features(V, F, dev, seed=0):
    """(V, F) view of a table whose rows are 16-byte aligned (pitch = round_up(F, 8) floats): what the one-kernel path needs."""
    g = torch.Generator(device=dev).manual_seed(seed)
    pitch = 128 if F > 96 else (F + 7) // 8 * 8           # (the wide shapes read every row in four 128-byte strips)
    return torch.randn(V, pitch, device=dev, generator=g)[:, :F]


class _Knobs:
    """Degree-path switches for one test: thresholds lowered so that small graphs take the path, FUSED on / off."""

    def __init__(self, fused, small_graphs=False, enabled=True):
        self.fused, self.small, self.enabled = fused, small_graphs, enabled

    def __enter__(self):
        from pna_amd import degree_groups as DG, functional as PF
        self.keep = (DG.ENABLED, DG.FUSED, DG.MIN_ROWS, DG.MIN_OUT, DG.TWO_SCALER_MIN_OUT, PF.SMALL_SIMPLE_ROWS)
        DG.ENABLED, DG.FUSED = self.enabled, self.fused
        if self.small:
            DG.MIN_ROWS, DG.MIN_OUT, DG.TWO_SCALER_MIN_OUT, PF.SMALL_SIMPLE_ROWS = 1, 1, 1, 0
        return self

    def __exit__(self, *a):
        from pna_amd import degree_groups as DG, functional as PF
        DG.ENABLED, DG.FUSED, DG.MIN_ROWS, DG.MIN_OUT, DG.TWO_SCALER_MIN_OUT, PF.SMALL_SIMPLE_ROWS = self.keep


# every instantiation: (feature blocks) F = 75: 2 full + half, 64: 2 full, 40: 1 full + half, 20 / 32: 1 full, 48: 1 + half, 80: 2 + half
@pytest.mark.parametrize("V,E,F,N,scalers,residual", [
    (200_000, 2_000_000, 75, 75, "identity amplification attenuation", True),
    (131_072, 600_000, 64, 64, "identity amplification attenuation", True),
    (160_000, 1_000_000, 40, 72, "identity amplification attenuation", False),
    (140_000, 1_400_000, 50, 50, "identity amplification attenuation", True),
    (150_000, 900_000, 80, 80, "identity amplification attenuation", True),
    (140_000, 700_000, 32, 40, "identity amplification attenuation", False),
    (140_000, 800_000, 20, 64, "amplification attenuation", False),
    (135_000, 900_000, 48, 48, "identity amplification", True),
    (135_000, 500_000, 17, 17, "identity amplification attenuation", True),
    # the wide shapes (two gather passes x two column panels): BASELINE configs[4]'s layer, and partial last blocks / panels
    (150_000, 1_200_000, 128, 128, "identity amplification attenuation", True),
    (140_000, 900_000, 120, 100, "identity amplification attenuation", False),
    (130_000, 700_000, 113, 81, "identity amplification", False),
    (130_000, 800_000, 128, 64, "identity amplification attenuation", False),     # two passes, one panel
    (130_000, 600_000, 64, 96, "identity amplification attenuation", False),      # one pass (2 full), two panels
])
def test_fused_layer_equals_two_kernel_paths(cuda_device, V, E, F, N, scalers, residual):
    from pna_amd import Graph, degree_groups as DG, functional as PF
    from pna_amd.synth import powerlaw_graph
    src, dst = powerlaw_graph(V, E, seed=V % 89, device=cuda_device)
    g = Graph(src, dst, V)
    layer = _layer(F, N, cuda_device, residual=residual, seed=N, scalers=scalers)
    h = _features(V, F, cuda_device, seed=F)
    with torch.no_grad():
        with _Knobs(fused=True, small_graphs=True):
            assert layer._degree_grouped_path(g, h) and DG.fused_applies(g, h, F, N)
            plan = DG.plan_of(g)
            # the statistics the contraction consumed == the production gather's aggregate, bit for bit
            dump = torch.zeros(plan.NV, 4 * F, device=cuda_device)
            y_dump = PF.simple_layer_degree_fused(layer, g, h, agg_out=dump)
            y_f = layer(g, h)
            for _ in range(3):
                assert torch.equal(layer(g, h), y_f)
            ref_agg = PF.degree_grouped_aggregate(layer, g, h, plan)[:plan.NV]
            real = plan.perm >= 0
            assert torch.equal(dump[real], ref_agg[real]), int((dump[real] != ref_agg[real]).any(dim=1).sum())
        with _Knobs(fused=False, small_graphs=True):
            assert layer._degree_grouped_path(g, h)
            y_g = layer(g, h)
        with _Knobs(fused=False, enabled=False):
            assert not layer._degree_grouped_path(g, h)
            y_p = layer(g, h)
    assert torch.isfinite(y_f).all()
    s = y_p.abs().max().item()
    assert (y_dump - y_f).abs().max().item() <= 1e-6 * s       # (the verification instantiation runs one workgroup per CU: same arithmetic)
    assert (y_f - y_g).abs().max().item() <= 2e-6 * s          # same W_D up to the order of its fp32 combination; fp16 x 2 against bf16 x 3 contraction: the same accuracy class
    assert (y_f - y_p).abs().max().item() <= 2e-5 * s


@pytest.mark.parametrize("name", [n for n in __import__("conftest").golden_names("dgl_simple_groups")])
def test_fused_layer_vs_reference_golden(cuda_device, name):
    """The REFERENCE's own PNASimpleLayer output (oracle/make_golden_degree_groups.py) on graphs where degree tiles exist, through
    the one-kernel path (group rows) + the two-kernel rest path (rare degrees, three 300-edge hubs)."""
    from conftest import load_golden
    from pna_amd import Graph, degree_groups as DG
    from pna_amd.dgl.pna_layer import PNASimpleLayer
    meta, a, sd = load_golden(name)
    F, N = meta["F"], meta["out_dim"]
    wide = F > 96                                          # (groups_f128: BASELINE configs[4]'s layer shape)
    if not ((17 <= F <= 80 or 113 <= F <= 128) and (N <= 80 or 49 <= F <= 64 or F >= 113)):
        pytest.skip("shape outside the one-kernel path (covered by test_gpu_degree_groups.py)")
    layer = PNASimpleLayer(F, N, meta["aggregators"], meta["scalers"], {"log": a["avg_log"]}, 0.0, True, meta["residual"])
    layer.load_state_dict(sd)
    layer = layer.to(cuda_device).eval()
    g = Graph(a["src"].long(), a["dst"].long(), meta["N"]).to(cuda_device)
    h = torch.zeros(meta["N"], 128 if wide else (F + 7) // 8 * 8, device=cuda_device)[:, :F]
    h.copy_(a["h"])
    with torch.no_grad(), _Knobs(fused=True, small_graphs=True):
        assert layer._degree_grouped_path(g, h) and DG.fused_applies(g, h, F, N)
        plan = DG.plan_of(g)
        assert plan.G > 0 and plan.NR > 0
        out = layer(g, h).cpu()
    torch.testing.assert_close(out, a["out"], rtol=1e-5, atol=1e-5)


def test_fused_layer_isolated_nodes_and_nonfinite_features(cuda_device):
    """A degree-0 group (rows without in-edges: the aggregate is 0, DGL's zero initialiser) and Inf / NaN features: the one-kernel
    path has the two-kernel path's Inf / NaN pattern."""
    from pna_amd import Graph
    from pna_amd.synth import powerlaw_graph
    V0, E, F, iso = 140_000, 900_000, 75, 400
    src, dst = powerlaw_graph(V0, E, seed=3, device=cuda_device)
    V = V0 + iso                                                   # `iso` nodes that only send
    src = torch.cat([src, torch.arange(V0, V, device=cuda_device)])
    dst = torch.cat([dst, torch.arange(0, iso, device=cuda_device)])
    g = Graph(src, dst, V)
    assert int((g.in_degrees() == 0).sum()) >= iso
    layer = _layer(F, F, cuda_device, seed=2)
    h = _features(V, F, cuda_device, seed=9)
    h[5, 3] = float("inf"); h[77, 70] = float("-inf"); h[1234, 74] = float("nan"); h[V0 + 1, 0] = float("inf")
    with torch.no_grad():
        with _Knobs(fused=True, small_graphs=True):
            y_f = layer(g, h)
        with _Knobs(fused=False, small_graphs=True):
            y_g = layer(g, h)
    assert torch.equal(torch.isnan(y_f), torch.isnan(y_g)) and torch.equal(torch.isinf(y_f), torch.isinf(y_g))
    fin = torch.isfinite(y_g)
    assert torch.equal(torch.sign(y_f[~fin & ~torch.isnan(y_g)]), torch.sign(y_g[~fin & ~torch.isnan(y_g)]))
    s = y_g[fin].abs().max().item()
    assert (y_f[fin] - y_g[fin]).abs().max().item() <= 2e-6 * s
    zero_rows = torch.nonzero(g.in_degrees() == 0).flatten()[:iso]
    lin, bn = layer.posttrans.fully_connected[0].linear, layer.batchnorm_h
    z = (lin.bias - bn.running_mean) / torch.sqrt(bn.running_var + bn.eps) * bn.weight + bn.bias
    ref = h[zero_rows] + torch.relu(z)[None, :]
    ok = torch.isfinite(ref)
    assert (y_f[zero_rows][ok] - ref[ok]).abs().max().item() <= 1e-5


@pytest.mark.parametrize("V,E,F,N", [(150_000, 1_200_000, 75, 75), (140_000, 1_000_000, 128, 128), (130_000, 700_000, 40, 72)])
def test_fused_layer_fp16_scales_over_a_wide_dynamic_range(cuda_device, V, E, F, N):
    """The power-of-two scales of the fp16 x 2 contraction (round 5; pna_x3_split.h): feature rows from 1e-15 to 1e+15 (a destination's
    statistics share ONE scale, set by its largest message), six decades of spread inside a row, the two halves of a wide row six
    decades apart either way (the second gather pass re-scales the accumulator), weight columns over eight decades (one scale per
    column).  Checked PER ELEMENT against the float64 contraction of the kernel's own fp32 statistics (its agg_out dump): within
    2e-6 of sum_k |a_k| |W_D[n][k]| carried through BatchNorm's scale (+ the fp32 floor of the epilogue's O(1) bias / BatchNorm terms) --
    measured 5e-7, the two-kernel path's bf16 x 3 contraction 7e-7 on the same inputs; a lost second fp16 term would be 5e-4 of the
    dominant product, a wrong scale a factor of two."""
    from pna_amd import Graph, degree_groups as DG, functional as PF
    from pna_amd.synth import powerlaw_graph
    src, dst = powerlaw_graph(V, E, seed=11, device=cuda_device)
    g = Graph(src, dst, V)
    layer = _layer(F, N, cuda_device, residual=False, seed=4)
    gen = torch.Generator(device=cuda_device).manual_seed(5)
    h = _features(V, F, cuda_device, seed=6)
    with torch.no_grad():
        # rows: 30 decades (the wide rows: 20 + the halves' six either way -- sums of squares stay inside fp32)
        h.mul_(10.0 ** torch.empty(V, 1, device=cuda_device).uniform_(-15 if F <= 96 else -10, 15 if F <= 96 else 10, generator=gen))
        h.mul_(10.0 ** torch.empty(V, F, device=cuda_device).uniform_(-6, 0, generator=gen))            # inside a row: 6 decades
        if F > 96:
            h[:, 64:].mul_(10.0 ** (6.0 * torch.randint(-1, 2, (V, 1), device=cuda_device, generator=gen).float()))
        lin, bn = layer.posttrans.fully_connected[0].linear, layer.batchnorm_h
        lin.weight.mul_((10.0 ** (torch.arange(N, device=cuda_device) % 9 - 4).float())[:, None])       # columns: 8 decades
        with _Knobs(fused=True, small_graphs=True):
            assert DG.fused_applies(g, h, F, N)
            y_f = layer(g, h)
            plan = DG.plan_of(g)
            dump = torch.zeros(plan.NV, 4 * F, device=cuda_device)
            PF.simple_layer_degree_fused(layer, g, h, agg_out=dump)
        with _Knobs(fused=False, small_graphs=True):
            y_g = layer(g, h)
        assert torch.isfinite(y_g).all() and torch.isfinite(y_f).all()
        live = plan.perm >= 0
        nodes = plan.perm[live].long()
        assert nodes.numel() > V // 2
        a = dump[live].double()                               # [mean | max | min | std] x F: the fp32 statistics the contraction consumed
        amp, att = (t[nodes].double()[:, None] for t in g.degree_scalers(2.3))
        W, b, K = lin.weight.double(), lin.bias.double(), 4 * F
        z = a @ W[:, :K].t() + amp * (a @ W[:, K:2 * K].t()) + att * (a @ W[:, 2 * K:].t())
        mass = a.abs() @ W[:, :K].abs().t() + amp.abs() * (a.abs() @ W[:, K:2 * K].abs().t()) + att.abs() * (a.abs() @ W[:, 2 * K:].abs().t())
        bn_scale = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
        ref = torch.relu((z + b - bn.running_mean.double()) * bn_scale + bn.bias.double())
        tol = 2e-6 * (mass * bn_scale.abs() + 10.0)
        worst = {}
        for name, y in (("one-kernel, fp16 x 2", y_f), ("two-kernel, bf16 x 3", y_g)):
            worst[name] = ((y[nodes].double() - ref).abs() / tol).max().item()
        assert worst["one-kernel, fp16 x 2"] <= 1.0 and worst["two-kernel, bf16 x 3"] <= 1.0, worst
        assert worst["one-kernel, fp16 x 2"] <= 2.0 * worst["two-kernel, bf16 x 3"] + 0.1, worst     # (the same accuracy class)
        # the bar bites: most products dwarf the O(1) floor, and the outputs are far from zero on the scale of the bar
        assert ((mass * bn_scale.abs() > 1e3).float().mean().item() > 0.3) and ((ref > 50 * tol).float().mean().item() > 0.1)


def _adversarial_case(cuda_device, V, E, F, N, ratio, w_big, arith, seed=21):
    """One statistic family (feature `f0`: its mean / max / min / std) `ratio` x the rest, with weight `w_big` x the others on it in every
    scaler block (0: pruned) -- the input class on which a row-scaled fp16 x 2 split is normwise- but not componentwise-accurate
    (VERDICT r5 weak #1).  Returns (worst error over the per-element bar 1e-5 |ref| + 2e-6 sum_k |w_k a_k| |bn scale|, tiles handed over,
    tiles) for the one-kernel layer under `arith`; the reference is the float64 contraction of the kernel's own fp32 statistics."""
    from pna_amd import Graph, degree_groups as DG, functional as PF
    from pna_amd.synth import powerlaw_graph
    src, dst = powerlaw_graph(V, E, seed=seed, device=cuda_device)
    g = Graph(src, dst, V)
    layer = _layer(F, N, cuda_device, residual=False, seed=seed + 1)
    h = _features(V, F, cuda_device, seed=seed + 2)
    f0 = 7
    keep = DG.FUSED_ARITH
    with torch.no_grad():
        h[:, f0].mul_(ratio)
        lin, bn = layer.posttrans.fully_connected[0].linear, layer.batchnorm_h
        lin.bias.zero_(); bn.running_mean.zero_(); bn.bias.zero_()          # (no O(1) epilogue terms: the bar is the contraction's alone)
        K = 4 * F
        cols = torch.tensor([s * K + a * F + f0 for s in range(len(layer.scalers)) for a in range(4)], device=cuda_device)
        lin.weight[:, cols] *= w_big
        try:
            DG.FUSED_ARITH = arith
            with _Knobs(fused=True, small_graphs=True):
                assert DG.fused_applies(g, h, F, N)
                plan = DG.plan_of(g)
                if arith == "guarded":
                    DG.guard_stats(plan, cuda_device, reset=True)
                y = layer(g, h)
                handed = DG.guard_stats(plan, cuda_device)[0] if arith == "guarded" else 0
                dump = torch.zeros(plan.NV, 4 * F, device=cuda_device)
                PF.simple_layer_degree_fused(layer, g, h, agg_out=dump)
        finally:
            DG.FUSED_ARITH = keep
        live = plan.perm >= 0
        nodes = plan.perm[live].long()
        a = dump[live].double()
        amp, att = (t[nodes].double()[:, None] for t in g.degree_scalers(2.3))
        W = lin.weight.double()
        blocks = [(1.0, W[:, :K])] + ([(amp, W[:, K:2 * K]), (att, W[:, 2 * K:3 * K])] if len(layer.scalers) == 3 else [])
        z = sum(sc * (a @ w.t()) for sc, w in blocks)
        mass = sum(abs(sc) * (a.abs() @ w.abs().t()) if not isinstance(sc, float) else a.abs() @ w.abs().t() for sc, w in blocks)
        bn_scale = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
        ref = torch.relu(z * bn_scale)
        tol = 1e-5 * ref.abs() + 2e-6 * mass * bn_scale.abs()
        assert torch.isfinite(y).all()
        worst = ((y[nodes].double() - ref).abs() / tol.clamp(min=1e-300)).max().item()
    return worst, handed, plan.NV // L_tile_rows(F, N)


def L_tile_rows(F, N):
    from pna_amd import _lib
    return _lib.lib().pna_fused_degree_tile_rows(F, N)


@pytest.mark.parametrize("ratio,w_big", [(1e4, 0.0), (1e6, 0.0), (1e7, 0.0), (1e7, 1e-8), (1e8, 0.0), (1e10, 1e-8), (1e12, 0.0), (1e12, 1e-8)])
def test_guarded_contraction_is_componentwise_accurate_on_adversarial_rows(cuda_device, ratio, w_big):
    """one statistic per row 1e4 .. 1e12 x the rest with a zero / 1e-8 weight on it.  The guarded fp16 x 2 layer (the
    default) holds the per-element bar like bf16 x 3 does; the rows it cannot certify went to the bf16 x 3 launch (counted); and the
    test BITES: round 5's unguarded fp16 x 2 leaves the bar from 1e7 on."""
    V, E, F, N = 140_000, 1_100_000, 75, 75
    worst_g, handed, tiles = _adversarial_case(cuda_device, V, E, F, N, ratio, w_big, "guarded")
    worst_3, _, _ = _adversarial_case(cuda_device, V, E, F, N, ratio, w_big, "bf16x3")
    assert worst_3 <= 1.0, worst_3
    assert worst_g <= 1.0, (worst_g, handed, tiles)
    if ratio >= 1e7:
        worst_u, _, _ = _adversarial_case(cuda_device, V, E, F, N, ratio, w_big, "fp16x2")
        assert worst_u > 1.0, ("the unguarded form was expected to leave the bar here", worst_u)
        assert handed > 0.5 * tiles, (handed, tiles)


@pytest.mark.parametrize("F,N", [(128, 128), (40, 72), (64, 96)])
def test_guarded_contraction_other_instantiations_adversarial(cuda_device, F, N):
    """The same on the wide shapes (two gather passes: the scale of the second pass; two panels) and a one-block shape."""
    worst_g, handed, tiles = _adversarial_case(cuda_device, 135_000, 900_000, F, N, 1e9, 0.0, "guarded", seed=5)
    assert worst_g <= 1.0 and handed > 0, (worst_g, handed, tiles)
    worst_u, _, _ = _adversarial_case(cuda_device, 135_000, 900_000, F, N, 1e9, 0.0, "fp16x2", seed=5)
    assert worst_u > 1.0, worst_u


def test_guard_hands_over_inside_a_replayed_hipgraph_and_leaves_its_words_zero(cuda_device):
    """The guard needs no host round trip: captured once (pna_amd.capture.GraphedForward), the layer is replayed on a benign input and on an
    adversarial one written into the same static buffer -- the second replay hands its tiles to the bf16 x 3 launch and equals the eager,
    guarded result bit for bit; afterwards the working words of the guard block and the dynamic schedule's counter pair are zero (ADVICE r5:
    a launch that left them non-zero would make every later launch on the plan skip tiles)."""
    from pna_amd import Graph, degree_groups as DG
    from pna_amd.capture import GraphedForward
    from pna_amd.synth import powerlaw_graph
    V, E, F = 140_000, 1_100_000, 75
    src, dst = powerlaw_graph(V, E, seed=77, device=cuda_device)
    g = Graph(src, dst, V)
    layer = _layer(F, F, cuda_device, residual=True, seed=78)
    h = _features(V, F, cuda_device, seed=79)
    with torch.no_grad():
        lin = layer.posttrans.fully_connected[0].linear
        lin.weight[:, [s_ * 4 * F + a * F + 7 for s_ in range(3) for a in range(4)]] = 0.0
        with _Knobs(fused=True, small_graphs=True):
            assert DG.fused_applies(g, h, F, F)
            plan = DG.plan_of(g)
            gf = GraphedForward(lambda x: layer(g, x), h, alias_inputs=True)
            DG.guard_stats(plan, cuda_device, reset=True)
            gf.graph.replay()
            y_benign = gf.static_out.clone()
            handed_benign = DG.guard_stats(plan, cuda_device, reset=True)[0]
            h[:, 7].mul_(1e9)                                   # the static input, in place: feature 7 (zero weight) dwarfs the row
            gf.graph.replay()
            y_adv = gf.static_out.clone()
            handed_adv = DG.guard_stats(plan, cuda_device)[0]
            y_eager = layer(g, h)
            for ws in plan.__dict__["_guard_ws"].values():
                assert ws[:2].tolist() == [0, 0], ws[:4].tolist()
            for c in plan.__dict__.get("_tile_counters", {}).values():
                assert c.tolist() == [0, 0], c.tolist()
    assert handed_benign <= 2 and handed_adv > 0.5 * (plan.NV // 64), (handed_benign, handed_adv)
    assert torch.equal(y_adv, y_eager)
    live = plan.perm[plan.perm >= 0].long()
    assert (y_adv[live][:, :7] - y_benign[live][:, :7]).abs().max().item() <= 1e-4 * y_benign.abs().max().item()   # (feature 7 has no weight: only its own residual column moves)


def test_guard_leaves_benign_inputs_on_the_fast_path(cuda_device):
    """Gaussian features and weights (the benchmark's input class), ReLU-like features with exact zeros, and a constant feature six decades
    below the others: the guard hands over at most a few tiles in a thousand / a few per cent -- and holds the bar either way."""
    V, E, F, N = 200_000, 2_000_000, 75, 75
    worst, handed, tiles = _adversarial_case(cuda_device, V, E, F, N, 1.0, 1.0, "guarded", seed=31)
    assert worst <= 1.0 and handed <= max(2, tiles // 200), (worst, handed, tiles)
    worst, handed, tiles = _adversarial_case(cuda_device, V, E, F, N, 1e-6, 1.0, "guarded", seed=32)       # one feature tiny
    assert worst <= 1.0 and handed <= tiles // 4, (worst, handed, tiles)


@pytest.fixture(scope="module")
def c3(cuda_device):
    from pna_amd import Graph
    from pna_amd.synth import powerlaw_graph
    V, E, F = 1_000_000, 10_000_000, 75
    src, dst = powerlaw_graph(V, E, seed=1234, device=cuda_device)
    g = Graph(src, dst, V)
    return g, _layer(F, F, cuda_device, seed=7), _features(V, F, cuda_device, seed=1)


def test_fused_layer_rows_vs_float64_at_full_size(cuda_device, c3):
    """BASELINE configs[2] (V = 1 M, E = 10 M, F = 75): sampled rows -- frequent degrees, rare degrees, the largest hubs -- of the
    shipped layer (one-kernel path) against a float64 restatement of models/dgl/pna_layer.py:189-216."""
    from pna_amd import degree_groups as DG
    g, layer, h = c3
    F = 75
    with torch.no_grad():
        assert layer._degree_grouped_path(g, h) and DG.fused_applies(g, h, F, F)
        y = layer(g, h)
    deg = g.in_degrees()
    plan = DG.plan_of(g)
    rows = torch.cat([torch.arange(0, 1500, device=cuda_device), torch.topk(deg, 40).indices, plan.rest_rows[:200],
                      plan.perm[plan.perm >= 0][-300:].long()])
    csr = g.csr
    amp, att = g.degree_scalers(2.3)
    lin, bn = layer.posttrans.fully_connected[0].linear, layer.batchnorm_h
    W, b = lin.weight.double(), lin.bias.double()
    # the north star's bar PER ELEMENT (not normalised by the row's largest value): 1e-5 relative + the fp32
    # rounding floor of a K = 12F sum evaluated in another order, 2e-6 x sum_k |w_k a_k| carried through BatchNorm's scale
    # (bench.py::sampled_check's bar).  The statistics here are float64 of the fp32 inputs: the std of the fp32 pipeline
    # (E[x^2] - E[x]^2 in fp32, models/dgl/aggregators.py:18-19) carries its own cancellation error, 2e-7 (E[x^2] + E[x]^2) / (2 std)
    # per feature -- added to the floor through |w|.
    worst = 0.0
    bn_scale = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
    for v in rows.tolist():
        lo, hi = int(csr.rowptr[v]), int(csr.rowptr[v + 1])
        m = h[csr.col[lo:hi].long()].double()
        mean, msq = m.mean(0), (m * m).mean(0)
        std = torch.sqrt(torch.relu(msq - mean ** 2) + 1e-5)
        a = torch.cat([mean, m.max(0).values, m.min(0).values, std])
        a3 = torch.cat([a, a * amp[v].double(), a * att[v].double()])
        z = b + W @ a3
        z = (z - bn.running_mean.double()) * bn_scale + bn.bias.double()
        ref = h[v].double() + torch.relu(z)
        std_err = 2e-7 * (msq + mean ** 2) / (2 * std)
        e = torch.cat([torch.zeros_like(mean), torch.zeros_like(mean), torch.zeros_like(mean), std_err])
        e3 = torch.cat([e, e * amp[v].double().abs(), e * att[v].double().abs()])
        mass = (W.abs() @ a3.abs() + b.abs()) * bn_scale.abs()
        tol = 1e-5 * ref.abs() + 2e-6 * mass + (W.abs() @ e3) * bn_scale.abs()
        worst = max(worst, ((y[v].double() - ref).abs() / tol).max().item())
    assert worst <= 1.0, worst


def test_fused_layer_200_runs_identical_bits_at_full_size(cuda_device, c3):
    """The round-2 experiment of this kernel produced wrong sums in a few wavefront tiles per launch, differently from run to run,
    with its running sums folded by packed fp32 instructions.  The shipped fold is single v_add / v_mul: 200
    launches at the C3 shape, identical bits -- and the statistics are the production gather's."""
    from pna_amd import degree_groups as DG, functional as PF
    g, layer, h = c3
    with torch.no_grad():
        assert DG.fused_applies(g, h, 75, 75)
        y0 = layer(g, h).clone()
        bad = 0
        for _ in range(200):
            bad += int(not torch.equal(layer(g, h), y0))
        assert bad == 0, f"{bad} of 200 runs differ"
        plan = DG.plan_of(g)
        dump = torch.zeros(plan.NV, 300, device=cuda_device)
        PF.simple_layer_degree_fused(layer, g, h, agg_out=dump)
        ref = PF.degree_grouped_aggregate(layer, g, h, plan)[:plan.NV]
        real = plan.perm >= 0
        assert torch.equal(dump[real], ref[real])


def test_rest_rows_beside_the_kernel_same_bits_as_behind_it_at_full_size(cuda_device, c3):
    """functional.run_fused_call: on a large graph the rest-row launches run on a second stream BESIDE the persistent kernel, which
    leaves DG.FUSED_SPARE_WGS workgroups out for them (pna_fused_degree_args.spare_workgroups, ABI 17).  Same kernels over the same
    rows: the bits of the serial order, every time -- eager, and replayed from a hipGraph (the fork / join events are captured)."""
    from pna_amd import degree_groups as DG, functional as PF
    from pna_amd.capture import GraphedForward
    g, layer, h = c3
    plan = DG.plan_of(g)
    assert plan.rest_overlap_applies(75) and not plan.rest_overlap_applies(32) and sum(plan.edge_split()) == g.number_of_edges()
    spare = DG.FUSED_SPARE_WGS
    with torch.no_grad():
        try:
            DG.FUSED_SPARE_WGS = 0
            assert not plan.rest_overlap_applies(75)
            y_serial = PF.simple_layer_degree_fused(layer, g, h).clone()
        finally:
            DG.FUSED_SPARE_WGS = spare
        bad = 0
        for _ in range(50):
            bad += int(not torch.equal(PF.simple_layer_degree_fused(layer, g, h), y_serial))
        assert bad == 0, f"{bad} of 50 runs differ from the serial order"
        # ... with work queued on the caller's stream before and after (the call must order itself against both)
        for _ in range(10):
            h2 = torch.empty(h.shape[0], h.stride(0), dtype=h.dtype, device=h.device)[:, :h.shape[1]].copy_(h)   # (the 16-byte row pitch)
            assert DG.fused_applies(g, h2, 75, 75)
            y2 = PF.simple_layer_degree_fused(layer, g, h2)
            z = y2 + 1.0
            del h2
            assert torch.equal(z, y_serial + 1.0)
    graphed = GraphedForward(lambda t: PF.simple_layer_degree_fused(layer, g, t), h, alias_inputs=True)
    for _ in range(5):
        assert torch.equal(graphed(h), y_serial)


@pytest.mark.gpu
@pytest.mark.parametrize("V,E,F,N", [(600_000, 5_000_000, 128, 128), (560_000, 6_000_000, 64, 100)])
def test_rest_rows_beside_the_kernel_wide_shapes(cuda_device, V, E, F, N):
    """The same for the wide instantiations (two gather passes and / or two column panels): beside = behind, bit for bit."""
    from pna_amd import Graph, degree_groups as DG, functional as PF
    from pna_amd.synth import powerlaw_graph
    src, dst = powerlaw_graph(V, E, seed=V % 97, device=cuda_device)
    g = Graph(src, dst, V)
    layer, h = _layer(F, N, cuda_device, residual=F == N, seed=2), _features(V, F, cuda_device, seed=4)
    plan = DG.plan_of(g)
    with torch.no_grad():
        assert DG.fused_applies(g, h, F, N) and plan.rest_overlap_applies(F)
        spare = DG.FUSED_SPARE_WGS
        try:
            DG.FUSED_SPARE_WGS = 0
            y_serial = PF.simple_layer_degree_fused(layer, g, h).clone()
        finally:
            DG.FUSED_SPARE_WGS = spare
        bad = sum(int(not torch.equal(PF.simple_layer_degree_fused(layer, g, h), y_serial)) for _ in range(20))
    assert bad == 0, f"{bad} of 20 runs differ from the serial order"


@pytest.mark.parametrize("F,N", [(75, 75), (64, 64), (40, 72), (50, 50), (20, 64), (33, 40), (65, 70), (128, 128), (117, 100)])
def test_contiguous_table_takes_the_one_kernel_path_with_the_same_bits(cuda_device, F, N):
    """PNASimpleLayer.forward(g, h) with a CONTIGUOUS (V, F) tensor -- what a caller of the reference API passes -- must
    land on the one-kernel path.  Rows of 4 F bytes are read through 64-bit lane addresses; the last feature block's window slides
    back to end at F, so no 16-byte strip reaches past a row (not even the table's last row, whose storage ends with it: the
    tensor here is allocated exactly).  Same bits as the 16-byte aligned table."""
    from pna_amd import Graph, degree_groups as DG
    from pna_amd.synth import powerlaw_graph
    V, E = 140_000, 1_000_000
    src, dst = powerlaw_graph(V, E, seed=8, device=cuda_device)
    g = Graph(src, dst, V)
    layer = _layer(F, N, cuda_device, seed=4, residual=F == N)
    ha = _features(V, F, cuda_device, seed=2)
    hp = torch.empty(V * F, device=cuda_device).view(V, F)                 # storage of exactly V * F floats
    hp.copy_(ha)
    with torch.no_grad(), _Knobs(fused=True, small_graphs=True):
        assert DG.fused_applies(g, ha, F, N) and DG.fused_applies(g, hp, F, N)
        assert hp.is_contiguous() and hp.untyped_storage().nbytes() == V * F * 4
        ya, yp = layer(g, ha), layer(g, hp)
    assert torch.equal(ya, yp)


def test_fused_entry_point_rejects_bad_arguments(cuda_device):
    from pna_amd import _lib
    L = _lib.lib()
    a = _lib.PnaFusedDegreeArgs()
    assert L.pna_fused_degree_f32(ctypes.byref(a), None) == 0            # M = 0: nothing to do
    a.M = 64
    assert L.pna_fused_degree_f32(ctypes.byref(a), None) == -1 and b"non-null" in L.pna_last_error()
    assert L.pna_fused_degree_image_bytes(96, 80) == 0 and L.pna_fused_degree_image_bytes(75, 81) == 0 and L.pna_fused_degree_image_bytes(16, 16) == 0
    assert L.pna_fused_degree_image_bytes(75, 75) == 10 * 10240 + 1024 and L.pna_fused_degree_image_bytes(64, 80) == 8 * 10240 + 1024   # (chunks of two fp16 terms + the columns' scales and guard thresholds)
    assert L.pna_fused_image_bytes(75, 75, 0, 1) == 10 * 15360 and L.pna_fused_image_bytes(75, 75, 1, 1) == 15 * 15360 and L.pna_fused_image_bytes(75, 75, 1, 0) == 15 * 10240 + 1024
    assert L.pna_fused_image_bytes(128, 128, 0, 0) == 16 * 2 * 8192 + 1024 and L.pna_fused_image_bytes(128, 128, 1, 0) == 0
    # ABI 21: the images lie exactly pna_fused_image_bytes apart (a padded stride would misplace the tails), the arithmetic is one
    # of PNA_FD_ARITH_*, the guarded form needs its workspace
    t = torch.zeros(1 << 16, dtype=torch.int32, device=cuda_device)
    f = torch.zeros(1 << 16, dtype=torch.float32, device=cuda_device)
    for k in ("tile_desc", "tile_ids", "row_perm", "w_img", "w_img_x3"):
        setattr(a, k, t.data_ptr())
    a.x = a.y = f.data_ptr()
    a.n_records, a.ldx, a.x_rows, a.F, a.N, a.n_nodes, a.ldy = 4, 80, 16, 75, 75, 16, 80
    a.image_stride, a.image_stride_x3 = L.pna_fused_image_bytes(75, 75, 0, 0) + 512, L.pna_fused_image_bytes(75, 75, 0, 1)
    a.arith = _lib.FD_ARITH_H2
    assert L.pna_fused_degree_f32(ctypes.byref(a), None) == -1 and b"image_stride" in L.pna_last_error()
    a.image_stride -= 512
    a.arith = 7
    assert L.pna_fused_degree_f32(ctypes.byref(a), None) == -1 and b"arith" in L.pna_last_error()
    a.arith = _lib.FD_ARITH_GUARDED
    assert L.pna_fused_degree_f32(ctypes.byref(a), None) == -1 and b"guard_ws" in L.pna_last_error()
    assert L.pna_fused_degree_guard_bytes(64) == 64 + 9 * 64
    # ABI 21, y_cols_writable: 0, or between N and min(ldy, 16 ceil(N / 16))
    a.arith = _lib.FD_ARITH_H2
    for bad in (74, 81, 96):
        a.y_cols_writable = bad
        assert L.pna_fused_degree_f32(ctypes.byref(a), None) == -1 and b"y_cols_writable" in L.pna_last_error(), bad


@pytest.mark.parametrize("F,N", [(75, 75), (40, 72), (64, 96), (128, 100), (20, 13)])
def test_rows_of_the_layers_own_output_buffer_are_written_as_whole_sectors(cuda_device, F, N, monkeypatch):
    """pna_fused_degree_args.y_cols_writable (round 6): into the padding of the buffer the layer allocates itself the kernel writes ZEROS up to
    the next multiple of 16 columns -- same bits in the N columns as without, nothing beyond the writable columns; a caller's own `out` (rows
    of exactly N floats here) is never padded."""
    from pna_amd import Graph, functional as PF
    from pna_amd.synth import powerlaw_graph
    V, E = 140_000, 1_000_000
    src, dst = powerlaw_graph(V, E, seed=9, device=cuda_device)
    g = Graph(src, dst, V)
    layer = _layer(F, N, cuda_device, seed=5, residual=F == N)
    h = _features(V, F, cuda_device, seed=3)
    marks = {}
    real_empty = torch.empty

    def marked_empty(*size, **kw):                              # the layer's buffers start as 7.0 everywhere: what the kernel leaves alone stays 7.0
        t = real_empty(*size, **kw)
        if t.dtype == torch.float32 and t.dim() == 2 and t.shape[0] == V:
            t.fill_(7.0)
        return t
    with torch.no_grad(), _Knobs(fused=True, small_graphs=True):
        out = {}
        for pad in (True, False):
            monkeypatch.setattr(PF, "WRITE_PADDING", pad)
            monkeypatch.setattr(torch, "empty", marked_empty)
            call = PF.FusedDegreeCall(layer, g, h)
            monkeypatch.setattr(torch, "empty", real_empty)
            call.set_spare(False)
            call.group_rows()
            call.rest_rows()
            torch.cuda.synchronize()
            out[pad] = (call.y.clone(), torch.as_strided(call.y, (V, call.y.stride(0)), (call.y.stride(0), 1)).clone(), call.plan)
    assert torch.equal(out[True][0], out[False][0])
    full, plan = out[True][1], out[True][2]
    pitch, w = full.shape[1], min(full.shape[1], (N + 15) // 16 * 16)
    rows = plan.perm[plan.perm >= 0].long()                     # the rows of the degree groups: the one-kernel launch's
    assert bool((full[rows][:, N:w] == 0).all()) and bool((full[rows][:, w:] == 7.0).all())
    assert bool((out[False][1][:, N:] == 7.0).all())
    if plan.NR:                                                  # the rest rows (two-kernel path) leave their padding alone
        assert bool((full[plan.rest_rows.long()][:, N:] == 7.0).all())


def test_tower_mode_rows_vs_float64_and_100_identical_runs_at_full_size(cuda_device, c3):
    """The tower mode of the one-kernel layer (PNALayer with one tower, models/dgl/pna_layer.py:33-76 + :130-145) at the C3 size:
    sampled rows -- frequent degrees, rare degrees, the largest hubs, isolated rows -- against a float64 restatement of the
    reference's formulas (per-edge pretrans Linear of [h_u | h_v], aggregators, scalers, posttrans Linear, graph norm, eval
    BatchNorm, mixing Linear + LeakyReLU, residual), and 100 launches with identical bits."""
    from pna_amd import degree_groups as DG, functional as PF
    from pna_amd.dgl.pna_layer import PNALayer
    g, _, h = c3
    F = 75
    torch.manual_seed(11)
    layer = PNALayer(F, F, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.3)}, 0.0, True, True,
                     towers=1, divide_input=False, residual=True).to(cuda_device).eval()
    tw = layer.towers[0]
    with torch.no_grad():
        tw.batchnorm_h.running_mean.normal_()
        tw.batchnorm_h.running_var.uniform_(0.5, 2.0)
    snorm = torch.rand(g.num_nodes, 1, device=cuda_device, generator=torch.Generator(device=cuda_device).manual_seed(2)) + 0.5
    with torch.no_grad():
        assert PF.tower_layer_degree_grouped_applies(layer, g, h) and PF.tower_layer_degree_fused_applies(layer, g, h)
        y = layer(g, h, None, snorm).clone()
        bad = sum(int(not torch.equal(layer(g, h, None, snorm), y)) for _ in range(100))
    assert bad == 0, f"{bad} of 100 runs differ"
    deg = g.in_degrees()
    plan = DG.plan_of(g)
    rows = torch.cat([torch.arange(0, 800, device=cuda_device), torch.topk(deg, 30).indices, plan.rest_rows[:150],
                      plan.perm[plan.perm >= 0][-200:].long(), torch.nonzero(deg == 0).flatten()[:20]])
    csr = g.csr
    amp, att = g.degree_scalers(2.3)
    pre, post, bn, mix = (tw.pretrans.fully_connected[0].linear, tw.posttrans.fully_connected[0].linear, tw.batchnorm_h, layer.mixing_network)
    Wp, bp, Wo, bo = pre.weight.double(), pre.bias.double(), post.weight.double(), post.bias.double()
    Wm, bm = mix.linear.weight.double(), mix.linear.bias.double()
    slope = mix.activation.negative_slope
    worst = 0.0
    for v in rows.tolist():
        lo, hi = int(csr.rowptr[v]), int(csr.rowptr[v + 1])
        hv = h[v].double()
        if hi > lo:
            hu = h[csr.col[lo:hi].long()].double()
            m = torch.cat([hu, hv.expand(hi - lo, F)], dim=1) @ Wp.t() + bp
            a = torch.cat([m.mean(0), m.max(0).values, m.min(0).values, torch.sqrt(torch.relu((m * m).mean(0) - m.mean(0) ** 2) + 1e-5)])
        else:
            a = torch.zeros(4 * F, dtype=torch.float64, device=cuda_device)
        cat = torch.cat([hv, a, a * amp[v].double(), a * att[v].double()])
        z = bo + Wo @ cat
        z = z * snorm[v].double()
        bn_scale = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
        z = (z - bn.running_mean.double()) * bn_scale + bn.bias.double()
        z = Wm @ z + bm
        ref = hv + torch.where(z >= 0, z, z * slope)
        # per element: 1e-5 relative + the fp32 rounding floor of the chain's sums carried to the output --
        # 2e-6 x the absolute mass of pretrans (in the aggregates), posttrans and mixing products; the fp32 std's own
        # cancellation error, 2e-7 (E[x^2] + E[x]^2) / (2 std), enters through |W| like in the simple layer's test
        if hi > lo:
            pm = (torch.cat([hu, hv.expand(hi - lo, F)], dim=1).abs() @ Wp.abs().t() + bp.abs()).max(0).values
            mean, msq = m.mean(0), (m * m).mean(0)
            std = torch.sqrt(torch.relu(msq - mean ** 2) + 1e-5)
            ea = torch.cat([2e-6 * pm, 2e-6 * pm, 2e-6 * pm, (2e-7 * (msq + mean ** 2) + 4e-6 * pm * m.abs().max(0).values) / (2 * std)])
        else:
            ea = torch.zeros(4 * F, dtype=torch.float64, device=cuda_device)
        ecat = torch.cat([torch.zeros_like(hv), ea, ea * amp[v].double().abs(), ea * att[v].double().abs()])
        mass_z = (Wo.abs() @ (2e-6 * cat.abs() + ecat) + 2e-6 * bo.abs()) * snorm[v].double().abs() * bn_scale.abs()
        zb = ((bo + Wo @ cat) * snorm[v].double() - bn.running_mean.double()) * bn_scale + bn.bias.double()
        tol = 1e-5 * ref.abs() + Wm.abs() @ (mass_z + 2e-6 * zb.abs()) + 2e-6 * bm.abs()
        worst = max(worst, ((y[v].double() - ref).abs() / tol).max().item())
    assert worst <= 1.0, worst


@pytest.mark.parametrize("F", [75, 64])
def test_tower_mode_fp16_scales_over_a_wide_dynamic_range(cuda_device, F):
    """Tower mode on the fp16 x 2 contraction: the row's scale is set by its neighbours' statistics first and LOWERED mid-tile when the row's
    own x_dst / h strips land (panel_rescale in pna_fused_degree.hip) -- with node rows spread over 16 decades the two bounds differ by
    many powers of two either way on most rows.  The one-kernel layer against the two-kernel grouped path (same statistics; bf16 x 3
    contraction, no scales), per element, normalised by the ROW's largest output: 2e-5 (a lost second term would be 5e-4, a wrong
    scale a factor of two)."""
    from pna_amd import Graph, degree_groups as DG, functional as PF
    from pna_amd.dgl.pna_layer import PNALayer
    from pna_amd.synth import powerlaw_graph
    V, E = 150_000, 1_200_000
    src, dst = powerlaw_graph(V, E, seed=13, device=cuda_device)
    g = Graph(src, dst, V)
    torch.manual_seed(12)
    layer = PNALayer(F, F, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.3)}, 0.0, True, True,
                     towers=1, divide_input=False, residual=True).to(cuda_device).eval()
    gen = torch.Generator(device=cuda_device).manual_seed(3)
    h = _features(V, F, cuda_device, seed=8)
    snorm = torch.rand(V, 1, device=cuda_device, generator=gen) + 0.5
    with torch.no_grad():
        layer.towers[0].batchnorm_h.running_mean.normal_()
        layer.towers[0].batchnorm_h.running_var.uniform_(0.5, 2.0)
        h.mul_(10.0 ** torch.empty(V, 1, device=cuda_device).uniform_(-8, 8, generator=gen))
        with _Knobs(fused=True, small_graphs=True):
            assert PF.tower_layer_degree_fused_applies(layer, g, h)
            y_f = layer(g, h, None, snorm)
        with _Knobs(fused=False, small_graphs=True):
            assert PF.tower_layer_degree_grouped_applies(layer, g, h)
            y_g = layer(g, h, None, snorm)
    assert torch.isfinite(y_g).all() and torch.isfinite(y_f).all()
    scale = y_g.abs().amax(1, keepdim=True)
    ratio = ((y_f - y_g).abs() / (2e-5 * scale + 1e-5)).max().item()
    assert ratio <= 1.0, ratio
    assert (scale > 1e3).float().mean().item() > 0.3            # (the bar bites: a third of the rows are far above the O(1) floor)


def _tower_layer_float64(layer, g, h, snorm):
    """PNALayer.forward (models/dgl/pna_layer.py:33-76, :130-145; eval) in float64 for every node at once, and the mass
    sum |w_k| |a_k| of its chain of sums carried to the output (the per-element bar's floor): per edge pretrans Linear of [h_u | h_v],
    the four aggregators, the three scalers, posttrans Linear, graph norm, eval BatchNorm, mixing Linear + LeakyReLU, residual."""
    dev = h.device
    V = h.shape[0]
    csr = g.csr
    dst = torch.repeat_interleave(torch.arange(V, device=dev), (csr.rowptr[1:] - csr.rowptr[:-1]).long())
    src = csr.col.long()
    deg = (csr.rowptr[1:] - csr.rowptr[:-1]).double()
    amp, att = (t.double()[:, None] for t in g.degree_scalers(2.3))
    towers, mix = list(layer.towers), layer.mixing_network
    T, Fi = len(towers), towers[0].in_dim
    outs, masses = [], []
    for t, tw in enumerate(towers):
        ht = (h[:, t * Fi:(t + 1) * Fi] if layer.divide_input else h).double()
        pre, post, bn = tw.pretrans.fully_connected[0].linear, tw.posttrans.fully_connected[0].linear, tw.batchnorm_h
        Wp, bp = pre.weight.double(), pre.bias.double()
        m = ht[src] @ Wp[:, :Fi].t() + ht[dst] @ Wp[:, Fi:2 * Fi].t() + bp
        dcl = deg.clamp(min=1)[:, None]
        s1 = torch.zeros(V, Fi, dtype=torch.float64, device=dev).index_add_(0, dst, m) / dcl
        s2 = torch.zeros(V, Fi, dtype=torch.float64, device=dev).index_add_(0, dst, m * m) / dcl
        mx = torch.full((V, Fi), -float("inf"), dtype=torch.float64, device=dev).scatter_reduce_(0, dst[:, None].expand(-1, Fi), m, "amax")
        mn = torch.full((V, Fi), float("inf"), dtype=torch.float64, device=dev).scatter_reduce_(0, dst[:, None].expand(-1, Fi), m, "amin")
        a = torch.cat([s1, mx, mn, torch.sqrt(torch.relu(s2 - s1 * s1) + 1e-5)], dim=1)
        a = torch.where((deg > 0)[:, None], a, torch.zeros_like(a))
        cat = torch.cat([ht, a, a * amp, a * att], dim=1)
        Wo, bo = post.weight.double(), post.bias.double()
        z, mz = cat @ Wo.t() + bo, cat.abs() @ Wo.abs().t() + bo.abs()
        if tw.graph_norm:
            z, mz = z * snorm.double(), mz * snorm.double().abs()
        if tw.batch_norm:
            sc = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
            z, mz = (z - bn.running_mean.double()) * sc + bn.bias.double(), (mz + bn.running_mean.double().abs()) * sc.abs() + bn.bias.double().abs()
        outs.append(z); masses.append(mz)
    z, mz = torch.cat(outs, dim=1), torch.cat(masses, dim=1)
    Wm, bm = mix.linear.weight.double(), mix.linear.bias.double()
    z, mz = z @ Wm.t() + bm, mz @ Wm.abs().t() + bm.abs()
    slope = mix.activation.negative_slope
    y = torch.where(z >= 0, z, z * slope)
    if layer.residual:
        y, mz = y + h.double(), mz + h.double().abs()
    return y, mz


def _tower_case(cuda_device, towers, divide_input, mutate, arith, seed=41):
    """The one-kernel tower layer under `arith` on inputs `mutate(layer, h)` has made adversarial, against float64: (worst error over the
    per-element bar 1e-5 |ref| + 2e-6 x the chain's mass, tiles handed over, tiles)."""
    from pna_amd import Graph, degree_groups as DG, functional as PF
    from pna_amd.dgl.pna_layer import PNALayer
    from pna_amd.synth import powerlaw_graph
    V, E, F = 140_000, 1_100_000, 75
    src, dst = powerlaw_graph(V, E, seed=seed, device=cuda_device)
    g = Graph(src, dst, V)
    torch.manual_seed(seed)
    layer = PNALayer(F, F, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.3)}, 0.0, True, True,
                     towers=towers, divide_input=divide_input, residual=False).to(cuda_device).eval()
    h = _features(V, F, cuda_device, seed=seed + 1)
    snorm = torch.rand(V, 1, device=cuda_device, generator=torch.Generator(device=cuda_device).manual_seed(seed + 2)) + 0.5
    keep = DG.FUSED_ARITH
    with torch.no_grad():
        for tw in layer.towers:
            tw.batchnorm_h.running_var.uniform_(0.5, 2.0)
            tw.batchnorm_h.running_mean.zero_(); tw.batchnorm_h.bias.zero_()
            tw.pretrans.fully_connected[0].linear.bias.zero_(); tw.posttrans.fully_connected[0].linear.bias.zero_()
        layer.mixing_network.linear.bias.zero_()             # (no O(1) terms anywhere in the chain: the bar is the products' alone)
        mutate(layer, h)
        keep_small = PF.SMALL_TOWER_ROWS
        try:
            DG.FUSED_ARITH, PF.SMALL_TOWER_ROWS = arith, 0         # (not the one-call kernel of molecule batches: divided input up to 262 k rows)
            with _Knobs(fused=True, small_graphs=True):
                assert PF.tower_layer_degree_fused_applies(layer, g, h) and not layer._small_batch_path(g, h)
                plan = DG.plan_of(g)
                if arith == "guarded":
                    DG.guard_stats(plan, cuda_device, reset=True)
                y = layer(g, h, None, snorm)
                handed = DG.guard_stats(plan, cuda_device)[0] if arith == "guarded" else 0
        finally:
            DG.FUSED_ARITH, PF.SMALL_TOWER_ROWS = keep, keep_small
        ref, mass = _tower_layer_float64(layer, g, h, snorm)
        rows = plan.perm[plan.perm >= 0].long()
        assert torch.isfinite(y).all()
        tol = 1e-5 * ref.abs() + 2e-6 * mass
        worst = ((y[rows].double() - ref[rows]).abs() / tol[rows].clamp(min=1e-300)).max().item()
    return worst, handed, plan.NV // 64


def _towers_apart(layer, h):
    """Five towers over divided input, tower t's input slice 100^t x the first one's (1e8 between the extremes), and a block-structured
    mixing network: output block t listens to tower t alone -- the collapsed image is block-diagonal over the statistics."""
    T, Fi = len(layer.towers), layer.towers[0].in_dim
    for t in range(T):
        h[:, t * Fi:(t + 1) * Fi].mul_(100.0 ** t)
    W = layer.mixing_network.linear.weight
    Fo = W.shape[1] // T
    mask = torch.zeros_like(W)
    for t in range(T):
        mask[t * (W.shape[0] // T):(t + 1) * (W.shape[0] // T), t * Fo:(t + 1) * Fo] = 1.0
    W.mul_(mask)


def _huge_own_feature(layer, h):
    """One tower; the node's own feature 7 is 1e9 x the rest, with zero weight wherever it enters: pretrans (both halves) and the
    posttrans self panel -- the row's scale must cover the huge strip (panel_rescale), the statistics sit 1e9 below it."""
    h[:, 7].mul_(1e9)
    tw = layer.towers[0]
    Fi = tw.in_dim
    tw.pretrans.fully_connected[0].linear.weight[:, [7, Fi + 7]] = 0.0
    tw.posttrans.fully_connected[0].linear.weight[:, 7] = 0.0


@pytest.mark.parametrize("towers,divide_input,mutate", [(5, True, _towers_apart), (1, False, _huge_own_feature)])
def test_guarded_tower_mode_on_adversarial_inputs(cuda_device, towers, divide_input, mutate):
    """VERDICT r5 item 1a, tower twin: block-diagonal divide_input=True images with towers 1e8 apart; a huge own feature in tower mode.
    Guarded (default) and bf16 x 3 hold the per-element bar against float64; round 5's unguarded fp16 x 2 does not."""
    worst_3, _, _ = _tower_case(cuda_device, towers, divide_input, mutate, "bf16x3")
    worst_g, handed, tiles = _tower_case(cuda_device, towers, divide_input, mutate, "guarded")
    worst_u, _, _ = _tower_case(cuda_device, towers, divide_input, mutate, "fp16x2")
    assert worst_3 <= 1.0, worst_3
    assert worst_g <= 1.0 and handed > 0, (worst_g, handed, tiles)
    assert worst_u > 1.0, ("the unguarded form was expected to leave the bar here", worst_u)


def test_guarded_tower_mode_benign(cuda_device):
    worst_g, handed, tiles = _tower_case(cuda_device, 1, False, lambda layer, h: None, "guarded")
    assert worst_g <= 1.0 and handed <= max(2, tiles // 100), (worst_g, handed, tiles)


def test_source_table_beyond_4_gib_and_2_pow_24_rows(cuda_device):
    """VERDICT r3 item 3 / BASELINE configs[4] at 8 ranks: a shard's [local | halo] table has > 2^24 rows and > 4 GiB, which round 3's
    32-bit byte offsets (__umul24) could not address -- that configuration fell to the two-kernel path.  Here: the same graph twice,
    once over a compact table and once with every source id spread over a 17.8 M-row / 5.7 GB table (id -> 17 id + 3); the
    one-kernel layer must give the same BITS (statistics through agg_out and outputs)."""
    from pna_amd import Graph, degree_groups as DG, functional as PF
    from pna_amd.synth import powerlaw_graph
    free, _ = torch.cuda.mem_get_info()
    if free < 12 << 30:
        pytest.skip("needs 12 GB of free device memory")
    V, E, F, K = 1 << 20, 6_000_000, 75, 17
    src, dst = powerlaw_graph(V, E, seed=21, device=cuda_device)
    g0 = Graph(src, dst, V)
    layer = _layer(F, F, cuda_device, seed=9)
    big = torch.empty(V * K, 80, device=cuda_device)                      # 17.8 M rows x 320 bytes = 5.7 GB
    assert big.shape[0] >= (1 << 24) and big.numel() * 4 > (1 << 32)
    big.normal_(generator=torch.Generator(device=cuda_device).manual_seed(5))
    xb = big[:, :F]
    h0 = xb[3::K][:V]                                                     # the rows the spread ids point at, as a strided view ...
    hc = torch.empty(V, 80, device=cuda_device)[:, :F]
    hc.copy_(h0)                                                          # ... and compact
    g1 = Graph(src, dst, V)
    csr = g1.csr
    csr.col.copy_((csr.col.long() * K + 3).to(csr.col.dtype))            # the same edges, sources in the big table (before any plan exists)
    assert int(csr.col.max()) >= (1 << 24)
    with torch.no_grad(), _Knobs(fused=True, small_graphs=True):
        plan0, plan1 = DG.plan_of(g0), DG.plan_of(g1)
        a0, a1 = torch.zeros(plan0.NV, 4 * F, device=cuda_device), torch.zeros(plan1.NV, 4 * F, device=cuda_device)
        c0 = PF.FusedDegreeCall(layer, g0, hc, x=hc, agg_out=a0)
        c1 = PF.FusedDegreeCall(layer, g1, hc, x=xb, agg_out=a1)
        y0, y1 = c0.group_rows().clone(), c1.group_rows().clone()
        assert torch.equal(plan0.perm, plan1.perm)
        live = plan0.perm >= 0
        assert torch.equal(a0[live], a1[live])
        rows = plan0.perm[live].long()
        assert torch.equal(y0[rows], y1[rows])
        # ... and the production instantiation (no agg_out) at full speed -- unguarded fp16 x 2 like the verification instantiation for the
        # bit comparison (the guard hands a few tiles in ten thousand to the bf16 x 3 launch: other last bits) ...
        keep_arith, DG.FUSED_ARITH = DG.FUSED_ARITH, "fp16x2"
        try:
            c2 = PF.FusedDegreeCall(layer, g1, hc, x=xb)
        finally:
            DG.FUSED_ARITH = keep_arith
        y2 = c2.group_rows().clone()
        assert torch.equal(y2[rows], y0[rows])
        # ... and the default, guarded: the same values up to the arithmetic of the few tiles handed over
        c2g = PF.FusedDegreeCall(layer, g1, hc, x=xb)
        y2g = c2g.group_rows().clone()
        differ = (y2g[rows] != y0[rows]).any(1)
        assert int(differ.sum()) <= rows.numel() // 100 and (y2g[rows] - y0[rows]).abs().max().item() <= 2e-5 * y0[rows].abs().max().item()
        # the rest rows (hub rows, rare degrees): the hand-scheduled gather's 64-bit-address instantiation over the big table
        assert plan0.NR > 0
        c3 = PF.FusedDegreeCall(layer, g0, hc, x=hc)
        y3 = c3.rest_rows().clone()
        y2 = c2.rest_rows().clone()
        assert torch.equal(y2[plan0.rest_rows], y3[plan0.rest_rows])


# WHICH path a shape takes is part of the contract -- a shape outside the one-kernel layer's instantiations must
# land on the two-kernel grouped path (or, beyond 128 outputs, on the ordinary kernels) visibly, not silently; and whatever ran
# must match the oracle.  "one" = pna_fused_degree_f32 over the group rows, "two" = gather in degree order + grouped contraction,
# "ordinary" = gather + three-block contraction in node order.
@pytest.mark.parametrize("F,N,path,aggregators,scalers", [(f, n, p, "mean max min std", "identity amplification attenuation") for f, n, p in [
    (75, 75, "one"), (64, 64, "one"), (80, 80, "one"), (17, 40, "one"), (40, 72, "one"),             # one gather pass, one panel
    (128, 128, "one"), (120, 100, "one"), (113, 81, "one"), (128, 64, "one"),                        # two gather passes
    (64, 96, "one"), (50, 128, "one"),                                                               # one pass of two full blocks, two panels
    (100, 100, "one"), (112, 112, "one"), (97, 64, "one"), (110, 110, "one"),                        # round 6: 97 <= F <= 112 as four blocks in two passes
    (96, 96, "one"), (81, 64, "one"), (90, 90, "one"), (95, 95, "one"),                              # round 6: 81 <= F <= 96 as feature panels [0, 64) + [64, F), partial sums between the launches
    (75, 96, "one"), (80, 128, "one"), (40, 100, "one"), (75, 160, "one"),                           # round 6: wider than an instantiation -> column panels, one launch each
    (16, 64, "two"),                                                                                 # F < 17
    (128, 192, "one"), (90, 160, "one"), (75, 250, "ordinary"),                                      # more than three panels: the ordinary kernels
]] + [
    # round 6: the operator sets of the reference's README ablations on the one-kernel layer
    (75, 75, "one", "mean max min std", "identity"),                                                 # "PNA (no scalers)" (README.md:76-77)
    (100, 100, "one", "mean max min std", "identity"),
    (75, 75, "one", "mean max min std", "identity amplification"),
    (110, 110, "one", "sum", "identity"), (100, 100, "one", "max", "identity"),                      # MPNN (sum) / (max) (README.md:91-92)
    (95, 95, "one", "mean max min std", "identity"), (90, 90, "one", "mean max min std", "identity"),  # "PNA (no scalers)" at its hidden sizes (README.md:76-77)
    (75, 75, "one", "mean", "identity amplification attenuation"),
    (64, 64, "one", "sum max", "identity attenuation"), (75, 80, "one", "mean sum max min std", "identity amplification attenuation"),
    (75, 75, "ordinary", "mean max min var", "identity amplification attenuation"),                  # var: not a linear image of the kernel's statistics
])
def test_which_path_a_shape_takes_and_that_it_matches_the_oracle(cuda_device, F, N, path, aggregators, scalers):
    from oracle import torch_oracle as O
    from pna_amd import Graph, degree_groups as DG, functional as PF
    from pna_amd.dgl.pna_layer import PNASimpleLayer
    from pna_amd.synth import powerlaw_graph
    V, E = 6000, 48_000
    src, dst = powerlaw_graph(V, E, seed=F + N)
    g = Graph(src, dst, V).to(cuda_device)
    torch.manual_seed(N)
    layer = PNASimpleLayer(F, N, aggregators, scalers, {"log": torch.tensor(2.3)}, 0.0, True, F == N)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(torch.randn_like(p) / (p.shape[-1] ** 0.5 if p.dim() == 2 else 3.0))
        layer.batchnorm_h.running_mean.normal_()
        layer.batchnorm_h.running_var.uniform_(0.5, 2.0)
    layer = layer.eval()
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    layer = layer.to(cuda_device)
    h = _features(V, F, cuda_device, seed=F)
    ran = []
    keep = (PF.run_fused_call, PF.degree_grouped_posttrans)
    # (recorded when the call has RETURNED: a path that raises "the hand-scheduled kernel was required" half-way hands the layer to the
    # ordinary kernels -- round 6: the rest rows of a non-standard aggregator list did exactly that, unseen, while the entry was recorded up front)
    PF.run_fused_call = lambda call: (keep[0](call), ran.append("one"))[0]
    PF.degree_grouped_posttrans = lambda *a, **k: (keep[1](*a, **k), ran.append("two"))[0]
    try:
        with torch.no_grad(), _Knobs(fused=True, small_graphs=True):
            y = layer(g, h).cpu()
    finally:
        PF.run_fused_call, PF.degree_grouped_posttrans = keep
    assert (ran or ["ordinary"]) == [path], (F, N, ran)
    # the north star's bar per element (bench.py's parity_check): the oracle's fp32 aggregate (reduce_func in fp32, like the reference:
    # a float64 std would part from BOTH at rows whose variance cancels, E[x^2] - E[x]^2), contracted in float64 -- 1e-5 relative + the
    # fp32 rounding floor of a K = 12 F sum in another order, 2e-6 x sum_k |w_k a_k| carried through BatchNorm's scale
    hc = h.cpu().contiguous()
    agg = O.reduce_bucketed(hc[src], src, dst, V, aggregators.split(), scalers.split(), torch.tensor(2.3)).double()
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    W, b = sd64["posttrans.fully_connected.0.linear.weight"], sd64["posttrans.fully_connected.0.linear.bias"]
    bn_scale = sd64["batchnorm_h.weight"] / torch.sqrt(sd64["batchnorm_h.running_var"] + 1e-5)
    z = ((agg @ W.t() + b) - sd64["batchnorm_h.running_mean"]) * bn_scale + sd64["batchnorm_h.bias"]
    ref = torch.relu(z) + (hc.double() if F == N else 0.0)
    mass = (agg.abs() @ W.abs().t() + b.abs()) * bn_scale.abs()
    # ... plus the fp32 rounding of the epilogue's own operands (4 ulp of |BatchNorm shift| + |residual|: a column whose BatchNorm
    # scale is tiny has next to no mass, its output is the shift -- rounded in fp32 like everything else)
    shift = (sd64["batchnorm_h.bias"] - sd64["batchnorm_h.running_mean"] * bn_scale).abs()
    eps_floor = 2.4e-7 * (shift + (hc.double().abs() if F == N else 0.0) + ref.abs())
    err, tol = (y.double() - ref).abs(), 1e-5 * ref.abs() + 2e-6 * mass + eps_floor
    if not bool((err <= tol).all()):
        ratio = err / tol
        r, c = divmod(int(ratio.argmax()), ratio.shape[1])
        deg = torch.bincount(dst, minlength=V)
        raise AssertionError(f"F={F} N={N} path={path}: worst err/tol {ratio.max().item():.3f} at row {r} (in-degree {int(deg[r])}) col {c}: got {y[r, c].item():.8g} "
                             f"ref {ref[r, c].item():.8g} tol {tol[r, c].item():.3g}; rows over: {int((ratio > 1).any(1).sum())}, their in-degrees "
                             f"{sorted(set(deg[(ratio > 1).any(1)].tolist()))[:12]}")


def _lib_image_bytes(F, N):
    from pna_amd import _lib
    return _lib.lib().pna_fused_degree_image_bytes(F, N)


@pytest.mark.parametrize("F,N", [(75, 75), (128, 128), (40, 72)])
def test_balanced_tile_order_gives_the_same_bits(cuda_device, F, N):
    """Round 5: the load-balanced tile list (DegreePlan.fused_balance: the same tiles dealt to the persistent workgroups in another
    order) against the plan's own order -- identical output BITS, simple layer and (F = 75) the one-tower layer with graph norm (its
    per-row factor travels with the rows)."""
    from pna_amd import Graph, degree_groups as DG, functional as PF
    from pna_amd.synth import powerlaw_graph
    V, E = 140_000, 1_200_000
    src, dst = powerlaw_graph(V, E, seed=11, device=cuda_device)
    layer = _layer(F, N, cuda_device, residual=(F == N), seed=3)
    h = _features(V, F, cuda_device, seed=2)
    outs = {}
    keep = DG.FUSED_BALANCE
    try:
        for mode in ("off", "lpt", "cheap_last", "dynamic"):
            DG.FUSED_BALANCE = mode
            g = Graph(src, dst, V)                               # (a fresh plan per mode)
            with torch.no_grad(), _Knobs(fused=True, small_graphs=True):
                assert DG.fused_applies(g, h, F, N)
                call = PF.FusedDegreeCall(layer, g, h, x=h)
                assert (DG.plan_of(g).fused_balance(PF._fused_grid(cuda_device, 0, DG.plan_of(g).NV // 64)) is not None) == (mode != "off")
                assert bool(call.args.tile_counter) == (mode == "dynamic")
                outs[mode] = PF.run_fused_call(call).clone()
                for i in range(4):                               # (the kernel leaves the counter pair zero: any grid may follow any)
                    call.set_spare(i % 2 == 1)
                    assert torch.equal(call.group_rows(), outs[mode])
                if mode == "dynamic":
                    torch.cuda.synchronize()
                    assert all(c.tolist() == [0, 0] for c in DG.plan_of(g).__dict__["_tile_counters"].values())
            if F == 75:
                from pna_amd.dgl.pna_layer import PNALayer
                torch.manual_seed(5)
                tl = PNALayer(F, F, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.3)}, 0.0, True, True, towers=1,
                              divide_input=False, residual=True).to(cuda_device).eval()
                sn = torch.rand(V, 1, device=cuda_device) + 0.5
                with torch.no_grad(), _Knobs(fused=True, small_graphs=True):
                    keep_t, PF.SMALL_TOWER_ROWS = PF.SMALL_TOWER_ROWS, 0
                    try:
                        assert PF.tower_layer_degree_fused_applies(tl, g, h)
                        outs["tower " + mode] = tl(g, h, None, sn).clone()
                    finally:
                        PF.SMALL_TOWER_ROWS = keep_t
    finally:
        DG.FUSED_BALANCE = keep
    assert torch.equal(outs["off"], outs["lpt"]) and torch.equal(outs["off"], outs["cheap_last"]) and torch.equal(outs["off"], outs["dynamic"])
    if F == 75:
        assert torch.equal(outs["tower off"], outs["tower lpt"]) and torch.equal(outs["tower off"], outs["tower cheap_last"])
        assert torch.equal(outs["tower off"], outs["tower dynamic"])
