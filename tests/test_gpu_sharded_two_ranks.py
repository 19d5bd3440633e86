"""The multi-GPU layer path with REAL halos on one GPU: two processes share cuda:0 and exchange over gloo (which stages
device tensors through the host), so that the destination-range shard, the de-duplicated halo all-to-all (both the
autograd exchange and the resident-table one bench.py uses) and the kernels over the extended [local | halo] table run
end to end.  Each rank's rows must equal the corresponding rows of the unsharded layer (same kernels, same edge order
within a row: bit-identical aggregation; the contraction runs on identical rows, so it is bit-identical too)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, V, E, F):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pna_amd import Graph, functional as PF
        PF.SMALL_SIMPLE_ROWS = 0      # the unsharded reference below runs the large-graph kernels too (bit-identity is between those)
        from pna_amd.dgl.pna_layer import PNALayer, PNASimpleLayer
        from pna_amd.shard import shard_graph
        from pna_amd.synth import powerlaw_graph
        src, dst = powerlaw_graph(V, E, seed=11, device=dev)
        gs = shard_graph(src, dst, V)
        assert gs.n_halo > 0
        g = Graph(src, dst, V)
        lo, hi = gs.lo, gs.hi
        torch.manual_seed(0)
        simple = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.0)},
                                0.0, True, True).to(dev).eval()
        tower = PNALayer(F, F, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.0)}, 0.0,
                         True, True, towers=5, divide_input=False, residual=True).to(dev).eval()
        h = torch.randn(V, F, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        snorm = torch.rand(V, 1, device=dev, generator=torch.Generator(device=dev).manual_seed(4)) + 0.5
        with torch.no_grad():
            want_s = simple(g, h)[lo:hi]
            want_t = tower(g, h, None, snorm)[lo:hi]
            # (a) features passed as an ordinary tensor: autograd-aware exchange + concatenation
            got_s = simple(gs, h[lo:hi].clone())
            got_t = tower(gs, h[lo:hi].clone(), None, snorm[lo:hi])
            # (b) features living in the shard's resident [local | halo] table (bench.py's N > 1 path), padded pitch
            hr = gs.alloc_features(F, pitch=F + 5)
            hr.copy_(h[lo:hi])
            got_r = simple(gs, hr)
        assert torch.equal(got_s, want_s)
        assert torch.equal(got_r, want_s)
        torch.testing.assert_close(got_t, want_t, rtol=1e-6, atol=1e-6)      # projected rows: library GEMM on a different row count
        # (c) the overlap path: (b) started the exchange asynchronously and aggregated the interior rows (only local sources)
        #     while it was in flight, then the boundary rows + hub segments -- two launches of the hand-scheduled kernel
        #     over disjoint work lists.  Same bits as the one-launch unsharded kernel, whatever the interior fraction.
        interior, items_in, items_bd = gs.split_work_lists()
        assert gs._pending is None and items_in.shape[0] + items_bd.shape[0] >= gs.num_nodes
        # a graph WITH locality (mostly the ring v <-> v+1): almost every row is interior
        src2, dst2 = powerlaw_graph(V, 2 * V + 600, seed=12, device=dev)
        gs2, g2 = shard_graph(src2, dst2, V, balance="edges"), Graph(src2, dst2, V)
        lo2, hi2 = gs2.lo, gs2.hi
        hr2 = gs2.alloc_features(F)                                          # dense pitch: exactly F floats per halo row travel
        hr2.copy_(h[lo2:hi2])
        with torch.no_grad():
            got2 = simple(gs2, hr2)
            want2 = simple(g2, h)[lo2:hi2]
            agg_split = __import__("pna_amd.functional", fromlist=["aggregate"]).aggregate(
                gs2, gs2.source_features(hr2, defer=True), F, ["mean", "max", "min", "std"])
            agg_whole = __import__("pna_amd.functional", fromlist=["aggregate"]).aggregate(
                gs2, gs2.source_features(hr2), F, ["mean", "max", "min", "std"])
        in2 = gs2.interior_mask()
        assert int(in2.sum()) > 0.8 * gs2.num_nodes and int((~in2).sum()) > 0
        assert torch.equal(got2, want2) and torch.equal(agg_split, agg_whole)
        # the HIP pack kernel = index_select
        from pna_amd import ops
        idx = torch.randint(0, V, (5000,), device=dev, dtype=torch.int32)
        for width, pitch in ((F, F), (75, 80), (3, 7), (128, 128)):
            tab = torch.randn(V, pitch, device=dev)[:, :width]
            assert torch.equal(ops.pack_rows(tab, idx), tab[idx.long()])
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_layers_on_one_gpu():
    mp.spawn(_worker, args=(2, _free_port(), 4000, 40000, 20), nprocs=2, join=True)


def _worker_grouped(rank, world, port, V, E, F):
    """The degree-grouped contraction on shards: every rank plans its OWN rows (a row's in-degree is global: the shard holds all
    its in-edges), the interior rows are gathered in degree order while the exchange is in flight, boundary rows and hub
    segments after it.  Against the unsharded three-block path: equal to the rounding of the combined weights (2e-6 of max|y|,
    the bar of tests/test_gpu_degree_groups.py); the aggregate itself stays bit-identical."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pna_amd import Graph, functional as PF, degree_groups as DG
        PF.SMALL_SIMPLE_ROWS = 0
        from pna_amd.dgl.pna_layer import PNASimpleLayer
        from pna_amd.shard import shard_graph
        from pna_amd.synth import powerlaw_graph
        torch.manual_seed(0)
        layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.0)},
                               0.0, True, True).to(dev).eval()
        h = torch.randn(V, F, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        for seed, edges, balance in ((11, E, "nodes"), (12, 2 * V + 600, "edges")):      # locality-free | mostly interior rows
            src, dst = powerlaw_graph(V, edges, seed=seed, device=dev)
            gs, g = shard_graph(src, dst, V, balance=balance), Graph(src, dst, V)
            lo, hi = gs.lo, gs.hi
            hr = gs.alloc_features(F)
            hr.copy_(h[lo:hi])
            with torch.no_grad():
                DG.ENABLED = False
                want = layer(g, h)[lo:hi]
                agg_want = PF.aggregate(g, h, F, layer.aggregators)[lo:hi]
                DG.ENABLED, DG.MIN_ROWS = True, 1
                DG.FUSED = False                      # (the TWO-kernel overlap path is the subject here; since round 4 the one-kernel layer
                                                      # would take any pitch: _worker_fused_shard covers it)
                assert layer._degree_grouped_path(gs, hr)
                got = layer(gs, hr)
                plan = DG.plan_of(gs)
                assert plan._split is not None and gs._pending is None           # the two-launch overlap path ran and was drained
                agg = PF.degree_grouped_aggregate(layer, gs, hr, plan)
                got_plain_tensor = layer(gs, h[lo:hi].clone())                   # not in the resident table: synchronous exchange
            assert plan.G > 0 and plan.NR > 0
            tol = 2e-6 * want.abs().max().item()          # (measured on one GPU: ~1e-7)
            assert (got - want).abs().max().item() <= tol and (got_plain_tensor - want).abs().max().item() <= tol
            # the plan-ordered aggregate holds the natural-order rows, bit for bit
            real = plan.perm >= 0
            assert torch.equal(agg[:plan.NV][real], agg_want[plan.perm[real].long()])
            assert torch.equal(agg[plan.NV:plan.NV + plan.NR], agg_want[plan.rest_rows])
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_degree_grouped_layer_on_one_gpu():
    mp.spawn(_worker_grouped, args=(2, _free_port(), 12000, 120000, 75), nprocs=2, join=True)


def _worker_pipeline(rank, world, port, V, E, F, L, n_blocks, fused=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pna_amd import Graph, functional as PF, degree_groups as DG
        PF.SMALL_SIMPLE_ROWS = 0
        DG.ENABLED = False            # the reference below: the ordinary three-block path (what the row blocks run), bit for bit
        from pna_amd.dgl.pna_layer import PNASimpleLayer
        from pna_amd.shard import BlockPipeline, shard_graph
        from pna_amd.synth import powerlaw_graph
        src, dst = powerlaw_graph(V, E, seed=21, device=dev)
        gs, g = shard_graph(src, dst, V), Graph(src, dst, V)
        assert gs.n_halo > 0 and int((g.in_degrees() > 128).sum()) > 0          # hub rows ride with block 0
        torch.manual_seed(0)
        layers = [PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.0)}, 0.0, True, True).to(dev).eval()
                  for _ in range(L)]
        h = torch.randn(V, F, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        with torch.no_grad():
            want = h
            for layer in layers:
                want = layer(g, want)
            pipe = BlockPipeline(gs, n_blocks)
            P = (F + 3) // 4 * 4
            ta = torch.zeros(gs.num_nodes + gs.n_halo, P, device=dev)
            tb = torch.full_like(ta, float("nan"))
            ta[: gs.num_nodes, :F] = h[gs.lo:gs.hi]
            rows = PF.SimpleLayerRows(layers, gs, n_blocks, fused=fused)
            made = []
            ctor = PF.FusedDegreeCall.__init__
            PF.FusedDegreeCall.__init__ = lambda self, *a, **k: (made.append(k.get("plan")), ctor(self, *a, **k))[1]
            try:
                res = pipe.run(rows, L, ta, tb)
            finally:
                PF.FusedDegreeCall.__init__ = ctor
            torch.cuda.synchronize()
        got, ref = res[: gs.num_nodes, :F], want[gs.lo:gs.hi]
        if not fused:
            assert not made and torch.equal(got, ref)
        else:
            # every block of every layer through the one-kernel layer with the BLOCK's plan (its own degree groups; rows whose degree
            # fills no tile inside the block take the three-block rest path: another rounding than the unsharded one-block result)
            assert len(made) == L * n_blocks and all(p is not None and p.row_range is not None for p in made)
            assert sum(p.with_heavy for p in made) == L
            assert not torch.isnan(got).any()
            torch.testing.assert_close(got, ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_blocks", [1, 4])
def test_two_rank_block_pipelined_layers_on_one_gpu(n_blocks):
    """Three PNASimpleLayers on two shards with the inter-layer halo exchange cut into row blocks (shard.BlockPipeline: block b of
    layer L is packed and sent while blocks b+1.. are still being computed): every rank's rows equal the unsharded three-layer
    result bit for bit."""
    mp.spawn(_worker_pipeline, args=(2, _free_port(), 6000, 70000, 20, 3, n_blocks), nprocs=2, join=True)


@pytest.mark.parametrize("n_blocks", [1, 4])
def test_two_rank_block_pipelined_layers_take_the_one_kernel_layer_per_block(n_blocks):
    """the row blocks of shard.BlockPipeline run pna_fused_degree_f32 over their own degree plans
    (DegreePlan(row_range=...)); hub rows ride with block 0; against the unsharded three-layer result (which runs the ordinary
    three-block path here) to 2e-5."""
    mp.spawn(_worker_pipeline, args=(2, _free_port(), 60000, 600000, 75, 3, n_blocks, True), nprocs=2, join=True)


def _worker_fused_shard(rank, world, port, V, E, F):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pna_amd import Graph, functional as PF, degree_groups as DG
        PF.SMALL_SIMPLE_ROWS = 0
        DG.MIN_ROWS = 1
        from pna_amd.dgl.pna_layer import PNASimpleLayer
        from pna_amd.shard import shard_graph
        from pna_amd.synth import powerlaw_graph
        src, dst = powerlaw_graph(V, E, seed=31, device=dev)
        gs, g = shard_graph(src, dst, V), Graph(src, dst, V)
        assert gs.n_halo > 0 and gs.interior_fraction() < DG.FUSED_HALO_MAX_INTERIOR
        torch.manual_seed(0)
        layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(2.0)}, 0.0, True, True).to(dev).eval()
        P = (F + 7) // 8 * 8
        hg = torch.zeros(V, P, device=dev)[:, :F]
        hg.copy_(torch.randn(V, F, device=dev, generator=torch.Generator(device=dev).manual_seed(3)))
        hr = gs.alloc_features(F, pitch=P)
        hr.copy_(hg[gs.lo:gs.hi])
        with torch.no_grad():
            assert DG.fused_applies(g, hg, F, F)
            want = layer(g, hg)[gs.lo:gs.hi]                         # unsharded: the one-kernel path
            calls = []
            orig = PF.simple_layer_degree_fused
            PF.simple_layer_degree_fused = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
            try:
                got = layer(gs, hr)
            finally:
                PF.simple_layer_degree_fused = orig
        assert calls, "the shard did not take the one-kernel path"
        # group rows: the same statistics, the same combined weight, the same contraction -- every row's bits; the shard's plan
        # differs from the whole graph's in WHICH rows are rest rows (a degree that fills a tile globally may not on a shard):
        # those go through the three-block contraction instead of W_D, equal to its rounding
        same = (got == want).all(dim=1)
        assert same.float().mean().item() > 0.9
        assert (got - want).abs().max().item() <= 2e-6 * want.abs().max().item()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_shards_without_locality_take_the_one_kernel_layer():
    """On a graph without locality the exchange overlap covers ~1 % of the rows: a shard whose features live in the resident
    table at an aligned pitch exchanges first and runs pna_fused_degree_f32 over the extended [local | halo] table."""
    mp.spawn(_worker_fused_shard, args=(2, _free_port(), 60000, 600000, 75), nprocs=2, join=True)
