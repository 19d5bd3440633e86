"""Host logic of the degree-grouped contraction (pna_amd/degree_groups.py) on CPU tensors: the plan's bookkeeping, the work
list it hands the gather, and the algebra the grouped kernels rely on -- in float64, with the plan's own tables:
    y[perm[v]] = W_D[tile_image[v // TILE]] a[perm[v]],  W_D = W_0 + amp(D) W_1 + att(D) W_2,  scalers taken at group_first_row
equals the reference formulation Linear([a | amp a | att a]) (models/dgl/pna_layer.py:203-206) row for row."""
import pytest
import torch

from pna_amd import Graph, degree_groups as DG
from pna_amd.synth import powerlaw_graph


@pytest.fixture(scope="module")
def graph_and_plan():
    V, E = 30_000, 240_000
    src, dst = powerlaw_graph(V, E, seed=5)
    g = Graph(src, dst, V)
    return g, DG.DegreePlan(g)


def test_plan_covers_every_node_once_in_single_degree_tiles(graph_and_plan):
    g, plan = graph_and_plan
    V = g.num_nodes
    deg = g.in_degrees().long()
    covered = torch.cat([plan.perm[plan.perm >= 0], plan.perm_rest[plan.perm_rest >= 0]]).long()
    assert covered.numel() == V and torch.equal(torch.sort(covered).values, torch.arange(V))
    assert plan.G > 0 and plan.NR > 0
    assert plan.NV % DG.TILE == 0 and plan.NRp % DG.TILE_REST == 0 and plan.rows == plan.NV + plan.NRp
    assert plan.tile_image.numel() == plan.NV // DG.TILE and plan.perm.numel() == plan.NV and plan.perm_rest.numel() == plan.NRp
    tiles = plan.perm.view(-1, DG.TILE).long()
    d = torch.where(tiles >= 0, deg[tiles.clamp(min=0)], torch.full_like(tiles, -1))
    assert bool(((d == d.max(dim=1, keepdim=True).values) | (d < 0)).all())                 # one in-degree per tile
    assert torch.equal(d.max(dim=1).values, plan.group_degree[plan.tile_image.long()].long())
    assert bool((tiles[:, 0] >= 0).all())                                                  # padding only trails a group
    assert torch.equal(deg[plan.group_first_row.long()], plan.group_degree.long())         # the row whose scalers stand for the group
    assert bool((torch.bincount(plan.tile_image.long(), minlength=plan.G) > 0).all())
    hs = g.heavy_schedule()
    rest = plan.perm_rest[plan.perm_rest >= 0].long()
    assert int((deg[rest] > hs.threshold).sum()) == hs.n_heavy                            # every hub row is a rest row
    in_group = plan.perm[plan.perm >= 0].long()
    counts = torch.bincount(deg[in_group])
    assert bool((counts[counts > 0] >= DG.TILE).all())                                     # only degrees with a whole tile of rows


def test_plan_work_list_is_the_graphs_with_output_rows(graph_and_plan):
    g, plan = graph_and_plan
    base = g.work_items()
    hs = g.heavy_schedule()
    n_seg = hs.n_seg if hs.n_heavy > 0 else 0
    assert plan.items.shape == base.shape and torch.equal(plan.items[:, 1:], base[:, 1:])
    assert torch.equal(plan.items[:n_seg], base[:n_seg])
    where = torch.cat([plan.perm, plan.perm_rest]).long()
    vrows = plan.items[n_seg:, 0].long()
    assert torch.equal(where[vrows], base[n_seg:, 0].long()) and torch.unique(vrows).numel() == vrows.numel()
    if hs.n_heavy:
        assert torch.equal(where[plan.heavy_out.long()], hs.heavy_rows.long()) and bool((plan.heavy_out >= plan.NV).all())


def test_combined_weight_algebra_with_the_plans_tables(graph_and_plan):
    g, plan = graph_and_plan
    V, F, N = g.num_nodes, 6, 5
    K = 4 * F
    gen = torch.Generator().manual_seed(0)
    a = torch.randn(V, K, generator=gen, dtype=torch.float64)
    W = torch.randn(N, 3 * K, generator=gen, dtype=torch.float64)
    deg = g.in_degrees().double()
    delta = torch.log(deg + 1).mean()
    amp = torch.log(deg + 1) / delta                                                      # models/dgl/scalers.py:12-14
    att = torch.where(deg > 0, delta / torch.log(deg + 1), torch.zeros_like(deg))        # :17-19 (rows without in-edges: aggregate is 0)
    want = torch.cat([a, a * amp[:, None], a * att[:, None]], dim=1) @ W.t()
    first = plan.group_first_row.long()
    W_D = W[None, :, :K] + amp[first, None, None] * W[None, :, K:2 * K] + att[first, None, None] * W[None, :, 2 * K:]
    got = torch.full((V, N), float("nan"), dtype=torch.float64)
    img = plan.tile_image.long().repeat_interleave(DG.TILE)                               # image of every virtual row
    real = plan.perm >= 0
    nodes = plan.perm[real].long()
    got[nodes] = torch.einsum("vnk,vk->vn", W_D[img[real]], a[nodes])
    rest = plan.perm_rest[plan.perm_rest >= 0].long()
    sc = plan.rest_scales(("t",), [None, amp.float(), att.float()])
    assert sc[0] is None and torch.equal(sc[1][:plan.NR], amp.float()[rest]) and bool((sc[1][plan.NR:] == 0).all())
    got[rest] = want[rest]
    assert not bool(torch.isnan(got).any())
    assert (got - want).abs().max().item() <= 1e-12 * want.abs().max().item()


def test_aggregate_pitch_and_gating():
    assert DG.agg_pitch(300) == 320 and DG.agg_pitch(512) == 512 and DG.agg_pitch(4) == 32
    V = DG.MIN_ROWS
    src, dst = powerlaw_graph(V, 4 * V, seed=1)
    g = Graph(src, dst, V)
    aggr = ("mean", "max", "min", "std")
    assert DG.applies(g, V, 75, 3, aggr) and DG.applies(g, V, 128, 3, aggr) and DG.applies(g, V, 128, 2, aggr) and DG.applies(g, V, 50, 3, aggr)
    assert not DG.applies(g, V, 32, 3, aggr) and not DG.applies(g, V, 75, 2, aggr) and not DG.applies(g, V, 75, 1, aggr)
    assert not DG.applies(g, V - 1, 75, 3, aggr) and not DG.applies(g, V, 75, 3, ("mean", "max")) and not DG.applies(g, V, 129, 3, aggr)


def test_output_pitch_bounds_the_grouped_paths_row_offsets():
    """the one-kernel layers allocate y at a 128-byte-line pitch (N = 75 -> 96 floats); the 4 GiB guard of applies() must
    use THAT pitch, or a graph of 11.2-13.4 M nodes with 65 <= N <= 80 is accepted here and refused by pna_fused_degree_f32."""
    from pna_amd import functional as PF
    for N in (40, 64, 65, 75, 80, 81, 96, 128):
        assert DG.out_pitch_floats(N) >= PF.out_pitch(N) and DG.out_pitch_floats(N) >= (80 if N <= 80 else 128)
    assert DG.out_pitch_floats(75) == 96 and DG.out_pitch_floats(64) == 80 and DG.out_pitch_floats(128) == 128
    V = DG.MIN_ROWS
    src, dst = powerlaw_graph(V, 4 * V, seed=1)
    g = Graph(src, dst, V)
    aggr = ("mean", "max", "min", "std")
    lim = (1 << 32) // (96 * 4)                              # rows of 96 floats below 4 GiB
    assert DG.applies(g, lim - 1, 75, 3, aggr) and not DG.applies(g, lim + 1, 75, 3, aggr)     # (V is only the caller's row count here)
    assert DG.applies(g, (1 << 32) // (80 * 4) - 1, 64, 3, aggr)


def test_plan_without_any_group():
    """No in-degree value fills a 128-row tile (a small graph): the plan is all rest rows, nothing crashes."""
    src, dst = powerlaw_graph(500, 5000, seed=1)
    g = Graph(src, dst, 500)
    plan = DG.DegreePlan(g)
    assert plan.G == 0 and plan.NV == 0 and plan.NR == 500 and plan.NRp == 576 and plan.perm.numel() == 0 and plan.tile_image.numel() == 0
    assert torch.equal(torch.sort(plan.perm_rest[:500].long()).values, torch.arange(500)) and bool((plan.perm_rest[500:] < 0).all())
    assert torch.equal(torch.sort(plan.items[:, 0].long()).values, torch.arange(500))
    assert DG.DegreePlan(g).serial != plan.serial                                          # cache keys tell two plans apart


def test_rest_rows_run_beside_the_kernel_only_on_large_graphs(graph_and_plan, monkeypatch):
    """DegreePlan.rest_overlap_applies (functional.run_fused_call): the rest-row launches go beside the one-kernel layer only when
    the kernel is long enough to hide them -- enough group rows, few rest edges."""
    g, plan = graph_and_plan
    e_g, e_r = plan.edge_split()
    deg = g.in_degrees().long()
    assert e_g + e_r == g.number_of_edges() and e_g == int(deg[plan.perm[plan.perm >= 0].long()].sum())
    assert not plan.rest_overlap_applies(75)                                               # 30 k rows: far too short a kernel
    monkeypatch.setattr(DG, "FUSED_OVERLAP_MIN_ROWS", 1024)
    assert plan.rest_overlap_applies(75) == (e_r <= DG.FUSED_OVERLAP_MAX_REST_EDGES * e_g)
    monkeypatch.setattr(DG, "FUSED_OVERLAP_MAX_REST_EDGES", 1.0)
    assert plan.rest_overlap_applies(75) and plan.rest_overlap_applies(128) and not plan.rest_overlap_applies(40)
    monkeypatch.setattr(DG, "FUSED_SPARE_WGS", 0)
    assert not plan.rest_overlap_applies(75)


@pytest.mark.parametrize("n_blocks", [1, 3, 8])
def test_block_plans_partition_the_nodes(n_blocks):
    """The per-block plans of shard.BlockPipeline's row blocks (round 4: DegreePlan(row_range, with_heavy, extra_rows, drop_rest), as
    functional.SimpleLayerRows.block_plan builds them): blocks 1.. hold only rows of their own range in single-degree tiles and no
    rest rows; block 0 takes its range, the hub rows of the WHOLE graph and every block's leftovers; together: every node once."""
    V, E = 40_000, 400_000
    src, dst = powerlaw_graph(V, E, seed=11)
    g = Graph(src, dst, V)
    deg = g.in_degrees().long()
    hs = g.heavy_schedule()
    heavy = deg > hs.threshold
    assert int(heavy.sum()) > 0
    bounds = [(V * i) // n_blocks for i in range(n_blocks + 1)]
    plans, left = {}, []
    for i in range(1, n_blocks):
        plans[i] = DG.DegreePlan(g, row_range=(bounds[i], bounds[i + 1]), with_heavy=False, drop_rest=True)
        left.append(plans[i].dropped_rest)
    plans[0] = DG.DegreePlan(g, row_range=(bounds[0], bounds[1]), with_heavy=True, extra_rows=torch.cat(left) if left else None)
    seen = []
    for i, plan in plans.items():
        rows_g = plan.perm[plan.perm >= 0].long()
        rows_r = plan.perm_rest[plan.perm_rest >= 0].long()
        seen += [rows_g, rows_r]
        tiles = plan.perm.view(-1, DG.TILE).long()
        d = torch.where(tiles >= 0, deg[tiles.clamp(min=0)], torch.full_like(tiles, -1))
        assert bool(((d == d.max(dim=1, keepdim=True).values) | (d < 0)).all())             # one in-degree per tile
        assert not bool(heavy[rows_g].any())                                                 # hub rows are never group rows
        assert plan.items is None and plan.row_range == (bounds[i], bounds[i + 1])
        if i > 0:
            assert plan.NR == 0 and not plan.with_heavy
            assert bool(((rows_g >= bounds[i]) & (rows_g < bounds[i + 1])).all())
            dr = plan.dropped_rest.long()
            assert bool(((dr >= bounds[i]) & (dr < bounds[i + 1]) & ~heavy[dr]).all())
            items, hout, hsched = plan.rest_items(g)
            assert items.shape[0] == 0 and hout is None and hsched is None
        else:
            assert bool(heavy[rows_r].sum() == heavy.sum())                                  # all hub rows, wherever they lie
            items, hout, hsched = plan.rest_items(g)
            n_seg = hsched.n_seg if hsched.n_heavy > 0 else 0
            assert items.shape[0] - n_seg == int((~heavy[rows_r]).sum())                     # one whole-row record per light rest row
            assert bool((items[n_seg:, 0] >= 0).all()) and int(items[n_seg:, 0].max()) < plan.NR
    allrows = torch.cat(seen)
    assert allrows.numel() == V and torch.equal(torch.sort(allrows).values, torch.arange(V))
    e_total = sum(sum(p.edge_split()) for p in plans.values())
    assert e_total == int(deg.sum())


def test_weight_gradient_tables_of_the_plan(graph_and_plan):
    """DegreePlan.dw_tables (pna_posttrans_dw_grouped_f32): contiguous, equally long tile ranges per workgroup; one workspace entry per
    run of equal degree groups inside a range, numbered in tile order; replaying the kernel's walk reproduces entry_group."""
    g, plan = graph_and_plan
    nt = plan.NV // DG.TILE
    for n_wgs in (1, 7, 64, nt + 5):
        tg, wg_range, wg_entry, entry_group, n_entries = plan.dw_tables(n_wgs)
        assert tg.numel() == nt and wg_range.shape == (n_wgs, 2) and wg_entry.shape == (n_wgs,)
        lo, hi = wg_range[:, 0].long(), wg_range[:, 1].long()
        assert int(lo[0]) == 0 and int(hi[-1]) == nt and torch.equal(lo[1:], hi[:-1])
        assert int((hi - lo).max() - (hi - lo).min()) <= 1
        walked = []
        for w in range(n_wgs):
            a, b = int(lo[w]), int(hi[w])
            if a >= b:
                continue
            e = int(wg_entry[w])
            cur = int(tg[a])
            runs = [cur]
            for t in range(a, b):
                if int(tg[t]) != cur:
                    cur = int(tg[t]); runs.append(cur)
            assert e == len(walked)                                                          # entries are numbered in tile order
            walked += runs
        assert n_entries == len(walked) == entry_group.numel()
        assert walked == entry_group.tolist()
    node_of = plan.node_of_rows()
    assert node_of.numel() == plan.rows and torch.equal(plan.vmap32()[node_of[node_of >= 0].long()].long(), torch.nonzero(node_of >= 0).flatten())


def test_fused_balance_is_a_permutation_of_tiles_round_by_round(graph_and_plan):
    """Round 5: the one-kernel layer's LOAD-BALANCED tile lists (DegreePlan.fused_balance).  Always the same 64-row tiles in another order
    (every 16-row descriptor and its rows move together).  "lpt" (static schedule): the partial last round holds the cheapest tiles, every
    full round holds exactly the tiles the ascending order put there (the device stays degree-synchronous), the heaviest workgroup comes
    closer to the mean.  "dynamic" (tiles claimed from a counter): the G most expensive tiles first, longest first, the G cheapest last,
    ascending cost in between."""
    g, plan = graph_and_plan
    desc, ids, n_rec = plan.fused_tables()
    nt = plan.NV // 64
    cost = desc.view(nt, 4, 4)[:, :, 1].max(dim=1).values.double() + DG.FUSED_TILE_COST
    keep = DG.FUSED_BALANCE
    try:
        for mode in ("lpt", "dynamic"):
            DG.FUSED_BALANCE = mode
            for G in (8, 24, 40):
                d2, p2, src = plan.fused_balance(G)
                assert torch.equal(torch.sort(src).values, torch.arange(nt))
                assert torch.equal(d2.view(nt, 4, 4), desc.view(nt, 4, 4)[src]) and torch.equal(p2.view(nt, 64), plan.perm.view(nt, 64)[src])
                assert plan.fused_balance(G)[2] is src                                           # cached per (grid, mode)
                if mode == "lpt":
                    rem, n_full = nt % G, nt // G
                    if rem:
                        assert torch.equal(torch.sort(src[n_full * G:]).values, torch.arange(rem))    # the partial round: tiles 0 .. rem - 1 (cheapest)
                    for r in range(n_full):
                        assert torch.equal(torch.sort(src[r * G:(r + 1) * G]).values, torch.arange(rem + r * G, rem + (r + 1) * G))
                    wg = torch.arange(nt) % G
                    before = torch.zeros(G, dtype=torch.double).index_add_(0, wg, cost)
                    after = torch.zeros(G, dtype=torch.double).index_add_(0, wg, cost[src])
                    assert after.max() <= before.max() + 1e-9 and abs(after.sum() - before.sum()) < 1e-6
                else:
                    tail = DG.FUSED_DYNAMIC_TAIL * G
                    assert nt > 2 * (G + tail)
                    c = cost[src]
                    assert bool((c[:G - 1] >= c[1:G]).all()) and c[:G].min() >= c[G:].max() - 1e-9          # the G heaviest, longest first
                    assert c[nt - tail:].max() <= c[:nt - tail].min() + 1e-9                                # the cheapest tiles last ...
                    assert bool((c[nt - tail:-1] >= c[nt - tail + 1:]).all())                               # ... the very cheapest at the very end
                    assert bool((c[G:nt - tail - 1] <= c[G + 1:nt - tail]).all())                           # ascending in between
        DG.FUSED_BALANCE = "off"
        assert plan.fused_balance(8) is None
    finally:
        DG.FUSED_BALANCE = keep


def test_tiles_by_group_lists_every_tile_once_with_its_group(graph_and_plan):
    """DegreePlan.tiles_by_group (round 6: what pna_project_grouped_f32 walks): the plan's 128-row tiles re-listed sorted by degree group --
    the same (rows, group) pairs, every tile once, groups non-decreasing; group_scaler_values: one scaler value per group, the value of
    any of the group's rows."""
    g, plan = graph_and_plan
    perm_g, group_g = plan.tiles_by_group()
    nt = plan.NV // DG.TILE
    assert perm_g.numel() == plan.NV and group_g.numel() == nt
    assert bool((group_g[1:] >= group_g[:-1]).all())
    own = sorted((tuple(plan.perm.view(nt, DG.TILE)[t].tolist()), int(plan.tile_image[t])) for t in range(nt))
    listed = sorted((tuple(perm_g.view(nt, DG.TILE)[t].tolist()), int(group_g[t])) for t in range(nt))
    assert own == listed
    deg = g.in_degrees().float()
    amp = torch.log(deg + 1) / 1.7
    sc = plan.group_scaler_values([None, amp])
    assert sc.shape == (plan.G, 2) and bool((sc[:, 0] == 1).all())
    for t in range(nt):                                                                       # every live row of a tile carries its group's value
        rows = perm_g.view(nt, DG.TILE)[t]
        rows = rows[rows >= 0].long()
        assert bool((amp[rows] == sc[int(group_g[t]), 1]).all())


def test_own_buffer_cols_only_pads_the_last_panel_of_an_own_buffer(monkeypatch):
    """functional.own_buffer_cols (pna_fused_degree_args.y_cols_writable): zeros may be written behind column N only by the row's LAST column
    panel, up to the next multiple of 16 columns and never beyond the pitch; an inner panel, or the switch off: 0 (= N)."""
    from pna_amd import functional as PF
    assert PF.own_buffer_cols(96, 0, 75, 75) == 80 and PF.own_buffer_cols(76, 0, 75, 75) == 76 and PF.own_buffer_cols(96, 0, 80, 80) == 80
    assert PF.own_buffer_cols(160, 0, 64, 150) == 0                                          # an inner panel: the next columns are another panel's
    assert PF.own_buffer_cols(160, 128, 150, 150) == 32 and PF.own_buffer_cols(152, 128, 150, 150) == 24
    monkeypatch.setattr(PF, "WRITE_PADDING", False)
    assert PF.own_buffer_cols(96, 0, 75, 75) == 0
