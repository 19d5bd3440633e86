"""Run-to-run identical bits for every kernel family that keeps matrix-core work and vector arithmetic in flight on the same SIMD.
Why this file exists: an experimental fused kernel (tools/ubench/degree_fused.hip, DESIGN.md 4.7 point 7) produced wrong running
sums in a few wavefront tiles per launch, different ones every run, whenever two of its wavefronts shared a SIMD -- and was exact
again once its accumulation was written as single (non-packed) VALU instructions.  None of the shipped kernels shows the symptom;
these tests keep looking for it (every result is also checked against float64 / the oracle elsewhere)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
REPEATS = 6


def _same(fn):
    first = fn()
    first = [t.clone() for t in (first if isinstance(first, (tuple, list)) else (first,))]
    for _ in range(REPEATS):
        again = fn()
        again = again if isinstance(again, (tuple, list)) else (again,)
        for a, b in zip(first, again):
            assert torch.equal(a, b)


@pytest.mark.parametrize("arith", ["f32", "bf16x3"])
@pytest.mark.parametrize("M,F,N,S", [(400_000, 75, 75, 3), (300_000, 128, 128, 3), (200_000, 40, 20, 2)])
def test_contraction_kernels(cuda_device, arith, M, F, N, S):
    from pna_amd import ops
    gen = torch.Generator(device=cuda_device).manual_seed(M + N)
    K = 4 * F
    a = torch.randn(M, K, device=cuda_device, generator=gen)
    W = torch.randn(N, S * K, device=cuda_device, generator=gen) / (S * K) ** 0.5
    b = torch.randn(N, device=cuda_device, generator=gen)
    scales = [None] + [torch.rand(M, device=cuda_device, generator=gen) + 0.5 for _ in range(S - 1)]
    res = torch.randn(M, N, device=cuda_device, generator=gen)
    _same(lambda: ops.posttrans(a, K, W, scales, b, arith=arith, relu=True, residual=res))


def test_gather_kernels(cuda_device):
    from pna_amd import Graph, functional as PF
    from pna_amd.synth import powerlaw_graph
    V, E, F = 400_000, 4_000_000, 75
    src, dst = powerlaw_graph(V, E, seed=9, device=cuda_device)
    g = Graph(src, dst, V)
    h = torch.randn(V, 80, device=cuda_device)[:, :F]
    _same(lambda: PF.aggregate(g, h, F, ["mean", "max", "min", "std"]))


def test_tower_layer_large_and_small(cuda_device):
    from pna_amd import Graph
    from pna_amd.dgl.pna_layer import PNALayer
    from pna_amd.synth import molecule_batch, powerlaw_graph
    torch.manual_seed(0)
    layer = PNALayer(75, 75, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.0)}, 0.0, True, True,
                     towers=5, divide_input=False, residual=True).to(cuda_device).eval()
    V, E = 150_000, 1_500_000
    src, dst = powerlaw_graph(V, E, seed=4, device=cuda_device)
    g = Graph(src, dst, V)
    h = torch.randn(V, 75, device=cuda_device)
    snorm = torch.rand(V, 1, device=cuda_device) + 0.5
    with torch.no_grad():
        _same(lambda: layer(g, h, None, snorm))
        s2, d2, sizes = molecule_batch(128, seed=3)
        gm = Graph(s2, d2, int(sum(sizes)), sizes).to(cuda_device)
        hm = torch.randn(gm.num_nodes, 75, device=cuda_device)
        sm = torch.rand(gm.num_nodes, 1, device=cuda_device) + 0.5
        _same(lambda: layer(gm, hm, None, sm))


def test_full_size_50_repeats_every_large_graph_path(cuda_device):
    """BASELINE configs[2] (V = 1 M, E = 10 M, F = 75), 50 repeats each, identical bits: the three-block bf16x3 contraction over all
    rows, the degree-grouped two-kernel path (gather in plan order + one-block GRP contraction + three-block GRP rest) and the
    one-kernel path (pna_fused_degree_f32; its own 200-repeat test is tests/test_gpu_fused_degree.py).  VERDICT r2 asked for >= 50
    full-size repeats of the x3 / GRP kernels after the packed-fp32 observation (root cause: DESIGN.md 4.8.6,
    tools/ubench/pk_opsel_mfma_repro.hip -- one op_sel form of the packed-fp32 instructions beside MFMA wavefronts; the shipped
    kernels do not contain it: tests/test_build_resources.py)."""
    from pna_amd import Graph, degree_groups as DG
    from pna_amd.dgl.pna_layer import PNASimpleLayer
    from pna_amd.synth import powerlaw_graph
    V, E, F = 1_000_000, 10_000_000, 75
    src, dst = powerlaw_graph(V, E, seed=1234, device=cuda_device)
    g = Graph(src, dst, V)
    torch.manual_seed(5)
    layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.3)}, 0.0, True, True).to(cuda_device).eval()
    h = torch.randn(V, 80, device=cuda_device)[:, :F]
    keep = (DG.ENABLED, DG.FUSED)
    try:
        with torch.no_grad():
            for enabled, fused in ((False, False), (True, False), (True, True)):
                DG.ENABLED, DG.FUSED = enabled, fused
                y0 = layer(g, h).clone()
                bad = sum(int(not torch.equal(layer(g, h), y0)) for _ in range(50))
                assert bad == 0, (enabled, fused, bad)
    finally:
        DG.ENABLED, DG.FUSED = keep
