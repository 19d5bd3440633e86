"""pna_fused_roles_f32 (pna_amd/csrc/pna_fused_roles.hip): the one-kernel layer with gather / multiply wavefront ROLES -- round 4's
experiment towards VERDICT r3 item 1a.  It is parity-green and 2.4x slower than pna_fused_degree_f32 at C3 (the
multiply role is slower than the gather role, and per SIMD the roles cost max(G, M) where alternating wavefronts cost (G + M) / 2), so
it ships switched off (degree_groups.ROLES); these tests keep it correct: statistics bit-identical to pna_fused_degree_f32's (through
agg_out), outputs identical bits (the same bf16x3 contraction over the same weight images), contiguous rows, the give-up flag clean."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("V,E,F,N,pitch", [(40_000, 400_000, 75, 75, 80), (40_000, 400_000, 75, 75, 75), (30_000, 300_000, 64, 64, 64),
                                           (30_000, 300_000, 40, 50, 40), (30_000, 240_000, 20, 24, 24), (30_000, 300_000, 50, 75, 56),
                                           (30_000, 300_000, 33, 40, 33)])
def test_roles_kernel_equals_the_alternating_kernel(cuda_device, V, E, F, N, pitch):
    from pna_amd import Graph, degree_groups as DG, functional as PF
    from pna_amd.dgl.pna_layer import PNASimpleLayer
    from pna_amd.synth import powerlaw_graph
    src, dst = powerlaw_graph(V, E, seed=1234, device=cuda_device)
    g = Graph(src, dst, V)
    torch.manual_seed(0)
    layer = PNASimpleLayer(F, N, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.1)}, 0.0, True, F == N)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(torch.randn_like(p) / (p.shape[-1] ** 0.5 if p.dim() == 2 else 3.0))
    layer = layer.to(cuda_device).eval()
    h = torch.randn(V, pitch, device=cuda_device)[:, :F] if pitch != F else torch.randn(V * F, device=cuda_device).view(V, F)
    plan = DG.plan_of(g)
    assert plan.G > 0 and PF.roles_applies(g, h, F, N) or not DG.ROLES
    with torch.no_grad():
        y0 = torch.full((V, (N + 3) // 4 * 4), float("nan"), device=cuda_device)[:, :N]
        y1 = torch.full((V, (N + 3) // 4 * 4), float("nan"), device=cuda_device)[:, :N]
        a0, a1 = torch.zeros(plan.NV, 4 * F, device=cuda_device), torch.zeros(plan.NV, 4 * F, device=cuda_device)
        c0 = PF.FusedDegreeCall(layer, g, h, x=h, out=y0, agg_out=a0)
        c1 = PF.FusedRolesCall(layer, g, h, x=h, out=y1, agg_out=a1)
        c0.group_rows(); c1.group_rows()
        c0.rest_rows(); c1.rest_rows()
        c2 = PF.FusedRolesCall(layer, g, h, x=h)                       # the production instantiation (no agg_out)
        y2 = c2.group_rows().clone()
        c2.rest_rows()
    assert int(c1.err.item()) == 0
    live = plan.perm >= 0
    assert torch.equal(a0[live], a1[live])
    assert not torch.isnan(y1).any() and torch.equal(y0, y1)
    rows = plan.perm[live].long()
    assert torch.equal(c2.y[rows], y0[rows])
