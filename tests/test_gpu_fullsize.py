"""Parity at BASELINE.json's full roofline size (|V| = 1M, |E| = 10M, F = 75) on the GPU: the whole aggregate
against the C oracle, plus size-independent properties (determinism, exact scaling, in-row permutation
invariance of max/min)."""
import numpy as np
import pytest
import torch

from conftest import check_blocks
from oracle import c_oracle
from pna_amd import Graph, functional as PF
from pna_amd.synth import powerlaw_graph

pytestmark = pytest.mark.gpu
AGGS = ["mean", "max", "min", "std"]
V, E, F = 1_000_000, 10_000_000, 75


@pytest.fixture(scope="module")
def c3(cuda_device):
    src, dst = powerlaw_graph(V, E, seed=1234, device=cuda_device)
    g = Graph(src, dst, V)
    x = torch.randn(V, F, generator=torch.Generator().manual_seed(1234))
    return g, x


def test_full_size_aggregate_matches_oracle(cuda_device, c3):
    g, x = c3
    xd = x.to(cuda_device)
    got = PF.aggregate(g, xd, F, AGGS).cpu().numpy()
    rp, col = g.csr.rowptr.cpu().numpy(), g.csr.col.cpu().numpy()
    assert g.csr.max_degree > 3000 and g.heavy_schedule().n_heavy > 100          # the hub path is exercised
    xn = x.numpy()
    ref = c_oracle.segreduce(rp, col, xn, F, AGGS)
    ref64 = c_oracle.segreduce(rp, col, xn, F, AGGS, acc_double=True)
    # rounding-floor masses from the oracle itself: sum |m| and sum m^2 per row / feature, in float64
    m1 = c_oracle.segreduce(rp, col, np.abs(xn), F, ["sum"], acc_double=True).astype(np.float64)
    m2 = c_oracle.segreduce(rp, col, xn * xn, F, ["sum"], acc_double=True).astype(np.float64)
    wsum = np.repeat(np.diff(rp).astype(np.float64)[:, None], F, axis=1)
    check_blocks(got, ref, ref64, AGGS, 1, F, "C3", (m1, m2, wsum))


def test_full_size_properties(cuda_device, c3):
    g, x = c3
    xd = x.to(cuda_device)
    a = PF.aggregate(g, xd, F, AGGS)
    # determinism: same launch twice, and a different launch geometry, are bit-identical
    assert torch.equal(a, PF.aggregate(g, xd, F, AGGS))
    from pna_amd import ops
    c = g.csr
    b = ops.segreduce(c.rowptr, c.col, xd, F, AGGS, heavy=g.heavy_schedule(), workspace=g.workspace, items=g.work_items(),
                      tune=dict(unroll=2, rows_per_group=16))
    assert torch.equal(a, b)
    # exact scaling by a power of two: mean/max/min scale exactly (fp32 has no rounding under *4)
    a4 = PF.aggregate(g, xd * 4.0, F, AGGS)
    assert torch.equal(a4[:, :3 * F], a[:, :3 * F] * 4.0)
    # max/min do not depend on the order of the in-edges of a row: reverse every row's edge list
    rp = c.rowptr.long()
    pos = torch.arange(E, device=cuda_device)
    row = c.row.long()
    rev = rp[row] + (rp[row + 1] - 1 - pos)
    col_rev = c.col[rev].contiguous()
    r = ops.segreduce(c.rowptr, col_rev, xd, F, AGGS, heavy=g.heavy_schedule(), workspace=g.workspace, items=g.work_items())
    assert torch.equal(r[:, F:3 * F], a[:, F:3 * F])
    torch.testing.assert_close(r[:, :F], a[:, :F], rtol=1e-4, atol=1e-5)


def test_full_size_contraction_bf16x3_properties(cuda_device, c3):
    """The posttrans contraction at the roofline size (1M x 900 -> 75), where the layers use the bf16x3 kernel:
    sampled rows against float64, agreement with the exact f32-MFMA kernel on every row, determinism, and row
    permutation equivariance (a row's result does not depend on which tile / wavefront / lane computes it)."""
    from pna_amd import ops
    g, x = c3
    xd = x.to(cuda_device)
    agg = PF.aggregate(g, xd, F, AGGS)
    gen = torch.Generator().manual_seed(7)
    W = (torch.randn(F, 12 * F, generator=gen) / (12 * F) ** 0.5).to(cuda_device)
    b = torch.randn(F, generator=gen).to(cuda_device)
    amp, att = g.degree_scalers(2.2488)
    scales = [None, amp, att]
    assert V >= ops.X3_MIN_ROWS
    y = ops.posttrans(agg, 4 * F, W, scales, b, arith="bf16x3")
    # determinism
    assert torch.equal(y, ops.posttrans(agg, 4 * F, W, scales, b, arith="bf16x3"))
    # every row against the exact-fp32 kernel: differences are fp32 summation-order noise relative to the output scale
    y32 = ops.posttrans(agg, 4 * F, W, scales, b, arith="f32")
    assert (y - y32).abs().max().item() <= 2e-5 * y32.abs().max().item()
    # 4096 sampled rows against float64, error relative to the mass sum |a_k w_k|
    idx = torch.randint(0, V, (4096,), generator=gen).to(cuda_device)
    a64 = agg[idx].double()
    ref = b.double()[None, :].repeat(idx.numel(), 1)
    mass = b.abs().double()[None, :].repeat(idx.numel(), 1)
    for s, sc in enumerate(scales):
        f = torch.ones(idx.numel(), dtype=torch.float64, device=cuda_device) if sc is None else sc[idx].double()
        Ws = W[:, s * 4 * F:(s + 1) * 4 * F].double()
        ref = ref + f[:, None] * (a64 @ Ws.t())
        mass = mass + f.abs()[:, None] * (a64.abs() @ Ws.abs().t())
    assert ((y[idx].double() - ref).abs() / mass).max().item() <= 5e-7
    # row permutation equivariance, bit for bit
    perm = torch.randperm(V, generator=gen).to(cuda_device)
    yp = ops.posttrans(agg[perm].contiguous(), 4 * F, W, [None, amp[perm].contiguous(), att[perm].contiguous()], b, arith="bf16x3")
    assert torch.equal(yp, y[perm])
