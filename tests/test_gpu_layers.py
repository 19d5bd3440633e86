"""GPU parity of the drop-in layers against the golden vectors produced by the reference's own source
(tests/golden/, see oracle/make_golden.py).  Tolerance: 1e-5 relative (north_star) plus an absolute floor
of 1e-5 on O(1)-O(10) activations for the dot products' cancellation (the CPU reference itself moves by
that much when its GEMM runs with a different thread count, tests/test_oracle_golden.py)."""
import pytest
import torch

from conftest import golden_names, load_golden
from pna_amd import Graph
from pna_amd.dgl.pna_layer import PNALayer, PNASimpleLayer
from pna_amd.dgl.aggregators import AGGREGATORS as DGL_AGG
from pna_amd.dgl.scalers import SCALERS as DGL_SCALERS
from pna_amd.pytorch.pna.layer import PNALayer as DensePNALayer
from pna_amd.pytorch.pna.aggregators import AGGREGATORS as DENSE_AGG
from pna_amd.pytorch.pna.scalers import SCALERS as DENSE_SCALERS

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", golden_names("dgl_simple"))
def test_simple_layer_golden(cuda_device, name):
    meta, a, sd = load_golden(name)
    layer = PNASimpleLayer(meta["F"], meta["out_dim"], meta["aggregators"], meta["scalers"], {"log": a["avg_log"]}, 0.0,
                           True, meta["residual"], posttrans_layers=meta["posttrans_layers"])
    layer.load_state_dict(sd)
    layer = layer.to(cuda_device).eval()
    g = Graph(a["src"], a["dst"], meta["N"]).to(cuda_device)
    with torch.no_grad():
        out = layer(g, a["h"].to(cuda_device)).cpu()
        agg = layer.aggregate(g, a["h"].to(cuda_device)).cpu()
    # the (V, A*S*F) tensor of reduce_func vs the REFERENCE's own output: max/min blocks (all scalers)
    # bit-exact, the rest 1e-5 relative + the fp32 cancellation floor of conftest.check_blocks
    import numpy as np
    from conftest import check_blocks, mass_stats
    from oracle import c_oracle, torch_oracle as O
    aggs, scalers, F = meta["aggregators"].split(), meta["scalers"].split(), meta["F"]
    rowptr, order, _ = O.csr_by_dst(a["src"], a["dst"], meta["N"])
    rp, col = rowptr.numpy().astype(np.int32), a["src"][order].numpy().astype(np.int32)
    amp, att = c_oracle.degree_scalers(rp, float(a["avg_log"]))
    scales = [{"identity": None, "amplification": amp, "attenuation": att}[s] for s in scalers]
    ref64 = c_oracle.segreduce(rp, col, a["h"].numpy(), F, aggs, scales, acc_double=True)
    check_blocks(agg.numpy(), a["agg"].numpy(), ref64, aggs, len(scalers), F, name,
                 mass_stats(rp, a["h"].numpy()[col]), scales)
    torch.testing.assert_close(out, a["out"], **TOL)


@pytest.mark.parametrize("name", golden_names("dgl_tower"))
def test_tower_layer_golden(cuda_device, name):
    meta, a, sd = load_golden(name)
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
        torch.testing.assert_close(g.snorm_n().cpu(), a["snorm_n"])
    torch.testing.assert_close(out, a["out"], **TOL)


@pytest.mark.parametrize("name", golden_names("dense"))
def test_dense_layer_golden(cuda_device, name):
    meta, a, sd = load_golden(name)
    avg_d = {"log": a["avg_log"].to(cuda_device), "lin": a["avg_lin"].to(cuda_device)}
    layer = DensePNALayer(meta["in_features"], meta["out_features"], meta["aggregators"], meta["scalers"], avg_d,
                          towers=meta["towers"], self_loop=meta["self_loop"], divide_input=meta["divide_input"],
                          device=cuda_device)
    layer.load_state_dict(sd)
    layer = layer.to(cuda_device).eval()
    with torch.no_grad():
        out = layer(a["x"].to(cuda_device), a["adj"].to(cuda_device)).cpu()
    torch.testing.assert_close(out, a["out"], **TOL)


def test_dgl_registry_operators_on_a_mailbox(cuda_device):
    from oracle import torch_oracle as O
    h = torch.randn(37, 6, 20, generator=torch.Generator().manual_seed(0))
    for name, fn in DGL_AGG.items():
        got = fn(h.to(cuda_device)).cpu()
        ref = O.MAILBOX_AGGREGATORS[name](h)
        if name in ("max", "min"):
            assert torch.equal(got, ref)
        else:
            torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-6)
    avg = {"log": torch.tensor(1.3, device=cuda_device)}
    x = torch.randn(5, 8, device=cuda_device)
    for name, fn in DGL_SCALERS.items():
        ref = O.mailbox_scale(name, x.cpu(), 6, torch.tensor(1.3))
        assert torch.equal(fn(x, D=6, avg_d=avg).cpu(), ref)


def test_dense_registry_operators(cuda_device):
    from oracle import torch_oracle as O
    gen = torch.Generator().manual_seed(1)
    B, N, F = 3, 7, 5
    X = torch.randn(B, N, N, F, generator=gen)
    adj = (torch.rand(B, N, N, generator=gen) < 0.5).float()
    adj = torch.maximum(adj, torch.eye(N).roll(1, 0).unsqueeze(0))        # every row and column non-empty
    adj = torch.maximum(adj, torch.eye(N).roll(1, 1).unsqueeze(0))
    for name in O.DENSE_AGGREGATORS:
        fn = DENSE_AGG[name]
        for sl in (False, True):
            got = fn(X.to(cuda_device), adj.to(cuda_device), self_loop=sl, device=cuda_device).cpu()
            ref = O.DENSE_AGGREGATORS[name](X, adj + torch.eye(N).unsqueeze(0) if sl else adj)
            if name in ("max", "min"):
                assert torch.equal(got, ref), name
            else:
                torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-6)
    avg_d = {"log": torch.tensor(1.4), "lin": torch.tensor(3.1)}
    m = torch.randn(B, N, 12, generator=gen)
    for name, fn in DENSE_SCALERS.items():
        got = fn(m.to(cuda_device), adj.to(cuda_device), {k: v.to(cuda_device) for k, v in avg_d.items()}).cpu()
        torch.testing.assert_close(got, O.dense_scale(name, m, adj, avg_d), rtol=1e-6, atol=1e-7)


def test_sharded_layer_single_rank_rccl(cuda_device):
    """The multi-GPU code path (dst-range shard, halo all-to-all over RCCL, layer over [local | halo]) on a
    1-rank process group: exercises the NCCL/RCCL calls and shows the sharded layer equals the plain one."""
    import os
    import torch.distributed as dist
    from pna_amd.shard import shard_graph
    from pna_amd.synth import powerlaw_graph
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda_device)
    try:
        V, E, F = 3000, 30000, 20
        src, dst = powerlaw_graph(V, E, seed=5, device=cuda_device)
        gs = shard_graph(src, dst, V)
        g = Graph(src, dst, V)
        assert gs.n_halo == 0 and gs.num_nodes == V
        layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.0)},
                               0.0, True, True).to(cuda_device).eval()
        h = torch.randn(V, F, device=cuda_device)
        with torch.no_grad():
            assert torch.equal(layer(gs, h), layer(g, h))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", golden_names("net_molecules"))
def test_molecules_net_golden(cuda_device, name):
    """Whole PNANet (embeddings, L tower layers, GRU, per-graph readout through the segment-reduce kernel,
    MLPReadout) against the output of the reference's own net."""
    from pna_amd.nets import PNANet
    from test_host_logic import _net_params
    meta, a, sd = load_golden(name)
    net = PNANet(_net_params(meta, a))
    net.load_state_dict(sd)
    net = net.to(cuda_device).eval()
    g = Graph(a["src"], a["dst"], meta["N"], meta["sizes"]).to(cuda_device)
    with torch.no_grad():
        out = net(g, a["atoms"].to(cuda_device), a["bonds"].to(cuda_device), a["snorm_n"].to(cuda_device), None).cpu()
    torch.testing.assert_close(out, a["out"], **TOL)


def test_hiv_net_runs_and_trains(cuda_device):
    from pna_amd.nets import PNANetHIV
    from pna_amd.synth import molecule_batch
    src, dst, sizes = molecule_batch(12, mean_nodes=25.5, sd_nodes=12, lo=6, hi=60, seed=2, lognormal=True)
    V = sum(sizes)
    g = Graph(src, dst, V, sizes).to(cuda_device)
    avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
    net = PNANetHIV(dict(hidden_dim=32, out_dim=32, in_feat_dropout=0.0, dropout=0.0, L=3, readout="mean", batch_norm=True,
                         residual=True, aggregators="mean max min std", scalers="identity amplification attenuation",
                         avg_d=avg, posttrans_layers=1, device=cuda_device)).to(cuda_device)
    gen = torch.Generator().manual_seed(0)
    x = torch.stack([torch.randint(0, d, (V,), generator=gen) for d in (119, 4, 12, 12, 10, 6, 6, 2, 2)], dim=1).to(cuda_device)
    y = torch.randint(0, 2, (len(sizes),), generator=gen)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    losses = []
    for _ in range(10):
        opt.zero_grad()
        loss = net.loss(net(g, x), y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]
    net.eval()
    with torch.no_grad():
        assert net(g, x).shape == (len(sizes), 1)


def test_dense_registry_every_entry_vs_reference_golden(cuda_device):
    """All 13 aggregators x self_loop and all 5 scalers of the dense registries against the values the reference's
    own functions produced (tests/golden/dense_registry_all.npz)."""
    meta, a, _ = load_golden("dense_registry_all")
    X, adj = a["X"].to(cuda_device), a["adj"].to(cuda_device)
    assert set(meta["aggregators"]) == set(DENSE_AGG) and set(meta["scalers"]) == set(DENSE_SCALERS)
    for name in meta["aggregators"]:
        for sl in (0, 1):
            got = DENSE_AGG[name](X, adj, self_loop=bool(sl), device=cuda_device).cpu()
            ref = a[f"agg/{name}/{sl}"]
            if name in ("max", "min", "identity"):
                assert torch.equal(got, ref), (name, sl)
            else:
                torch.testing.assert_close(got, ref, rtol=2e-5, atol=5e-6, msg=lambda m: f"{name} self_loop={sl}: {m}")
    avg_d = {"log": a["avg_log"].to(cuda_device), "lin": a["avg_lin"].to(cuda_device)}
    for name in meta["scalers"]:
        got = DENSE_SCALERS[name](a["m"].to(cuda_device), adj, avg_d).cpu()
        torch.testing.assert_close(got, a[f"sca/{name}"], rtol=1e-6, atol=1e-7)


def test_dense_layer_with_exotic_aggregators(cuda_device):
    from oracle import torch_oracle as O
    gen = torch.Generator().manual_seed(2)
    B, N, Fi = 2, 6, 4
    adj = (torch.rand(B, N, N, generator=gen) < 0.5).float()
    adj = torch.maximum(adj, torch.eye(N).roll(1, 0).unsqueeze(0))
    adj = torch.maximum(adj, torch.eye(N).roll(1, 1).unsqueeze(0))
    avg_d = {"log": torch.tensor(1.2, device=cuda_device), "lin": torch.tensor(3.0, device=cuda_device)}
    layer = DensePNALayer(Fi, 4, ["mean", "softmax", "moment3", "max"], ["identity", "amplification"], avg_d, towers=1,
                          device=cuda_device).to(cuda_device).eval()
    x = torch.randn(B, N, Fi, generator=gen)
    with torch.no_grad():
        out = layer(x.to(cuda_device), adj.to(cuda_device))
    assert out.shape == (B, N, 4) and torch.isfinite(out).all()


def test_hipgraph_capture_replays_the_layer(cuda_device):
    """pna_amd.capture.GraphedForward: the whole tower-layer forward as one hipGraph, bit-identical to eager."""
    from pna_amd.capture import GraphedForward
    from pna_amd.synth import molecule_batch
    src, dst, sizes = molecule_batch(16, seed=3)
    V = sum(sizes)
    g = Graph(src, dst, V, sizes).to(cuda_device)
    avg = {"log": torch.tensor(1.1)}
    layer = PNALayer(30, 30, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True, towers=5,
                     divide_input=False, residual=True).to(cuda_device).eval()
    snorm = g.snorm_n()
    h1, h2 = torch.randn(V, 30, device=cuda_device), torch.randn(V, 30, device=cuda_device)
    with torch.no_grad():
        gf = GraphedForward(lambda x: layer(g, x, None, snorm), h1)
        assert torch.equal(gf(h1).clone(), layer(g, h1, None, snorm))
        assert torch.equal(gf(h2).clone(), layer(g, h2, None, snorm))
