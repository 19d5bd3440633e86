"""Degree-grouped posttrans of PNASimpleLayer (inference, large graphs).

Every PNA scaler is a function of the destination's in-degree alone (models/dgl/scalers.py:7-19: identity, amplification
log(D+1)/delta, attenuation delta/log(D+1), ...).  The posttrans Linear of the simple layer (models/dgl/pna_layer.py:206) sees
[s_0(D) a | s_1(D) a | s_2(D) a] with the SAME aggregate `a` in every block, so for all rows of one in-degree D

    sum_s s_s(D) (W_s a)  =  (sum_s s_s(D) W_s) a  =  W_D a            (K = 4F instead of 3 * 4F multiply-adds per output)

The reference never uses this (it concatenates 12F columns and calls nn.Linear); here the rows are ordered by degree once per
graph, the gather kernel writes the aggregate in that order (its work list carries the output row), and the bf16x3 contraction
runs with ONE scaler block and a per-tile weight image W_D.  On the power-law benchmark graph 63 degree values hold 99.6 % of
the rows; the rest (rare degrees, hub rows) go through the ordinary three-block contraction over a compacted row list.
Both launches scatter their output rows back to node order (pna_posttrans_args.row_perm), so nothing else sees the order.
Output widths up to 128, whole graphs and shards (every rank plans its own rows); DESIGN.md 4.2d has the measurements.
"""
import ctypes

import os

import torch

from . import _lib

TILE = 128                 # rows of a workgroup tile of the one-block grouped kernel (8 wavefronts x 16 rows, two workgroups per CU)
                           # (64 < out_dim <= 80 and 80 < out_dim <= 128 alike: the 128-column block runs one 8-wavefront workgroup per CU)
TILE_REST = 192            # ... of the three-block kernel that takes the rest (12 wavefronts x 16 rows)
MIN_ROWS = 1 << 17         # graphs smaller than this keep the ordinary path (the grouping is worth it when launches are long)
ENABLED = True
# Where the grouping pays (tools/dg_shapes_time.py [removed in round 5: git history], 1 M nodes / 10 M edges, one run, grouped vs ordinary layer, ms): 3 scalers
# F = 20: 0.524 / 0.497, 32: 0.524 / 0.519, 50: 0.973 / 1.017, 64: 0.908 / 1.058, 75: 1.296 / 1.441, 96: 1.426 / 1.998;
# 2 scalers F = 75: 1.297 / 1.288, 128: 1.880 / 2.277.
MIN_OUT = 40               # narrower outputs keep the ordinary path (no gain measured at 20 and 32)
TWO_SCALER_MIN_OUT = 81    # two scaler blocks -> one saves half, not two thirds: a gain only on the 128-column block
TILE_ORDER = os.environ.get("PNA_AMD_TILE_ORDER", "ascending")   # order of the 128-row tiles of the virtual row order: ascending | interleave
AGG_ALIGN = 32             # the aggregate's row pitch is rounded up to this many floats (32 = 128-byte lines; 1 = packed rows), see agg_pitch


_PLAN_SERIAL = __import__("itertools").count()


class DegreePlan:
    """Row order, work list and tile -> group table of one graph (cached on the Graph; independent of weights / scalers)."""

    def __init__(self, graph, row_range=None, with_heavy=True, extra_rows=None, drop_rest=False):
        """row_range = (r0, r1): a plan of the rows [r0, r1) only -- one ROW BLOCK of shard.BlockPipeline, so that every block of a
        pipelined multi-layer run takes the one-kernel layer.  The hub rows (cut into segments by the heavy
        schedule) belong to the block built with_heavy=True wherever they lie, like the blocks' gather work lists; row ids stay
        the graph's own (perm, rest_rows index the full tables).  drop_rest: the rows of the range whose degree fills no tile are
        NOT this plan's (`dropped_rest` lists them); extra_rows: rows from outside the range that this plan takes on top -- the
        pipeline hands every block's leftovers to block 0, so that blocks 1.. are ONE launch each and the chain of small rest-row
        launches runs once per layer."""
        self.serial = next(_PLAN_SERIAL)       # identifies the plan in cache keys (id() of a freed plan can come back for another graph)
        csr, hs = graph.csr, graph.heavy_schedule()
        dev = csr.rowptr.device
        rp = csr.rowptr.long()
        deg = rp[1:] - rp[:-1]
        V = deg.numel()
        hub_limit = hs.threshold if hs.threshold > 0 else (1 << 30)     # (heavy schedule off: no row is cut into segments)
        self.row_range, self.with_heavy = row_range, bool(with_heavy)
        if row_range is None and with_heavy and extra_rows is None and not drop_rest:
            order = torch.sort(deg, stable=True).indices                 # rows by in-degree (ties: ascending row id)
        else:
            r0, r1 = (0, V) if row_range is None else (int(row_range[0]), int(row_range[1]))
            ids = torch.arange(V, device=dev)
            heavy = deg > hub_limit
            sel = ((ids >= r0) & (ids < r1) & ~heavy) | (heavy if with_heavy else torch.zeros_like(heavy))
            if extra_rows is not None and extra_rows.numel():
                sel[extra_rows.long()] = True
            rows_sel = torch.nonzero(sel).flatten()
            order = rows_sel[torch.sort(deg[rows_sel], stable=True).indices]
        n_sel = int(order.numel())
        ud, cnt = torch.unique_consecutive(deg[order], return_counts=True)
        big = (cnt >= TILE) & (ud <= hub_limit)                          # degree values with at least one whole tile of light rows
        start = torch.cumsum(cnt, 0) - cnt
        gid = torch.repeat_interleave(torch.arange(ud.numel(), device=dev), cnt)     # group of every sorted row
        in_big = big[gid]
        padded = (cnt[big] + TILE - 1) // TILE * TILE
        vstart = torch.cumsum(padded, 0) - padded
        big_index = torch.cumsum(big.long(), 0) - 1
        self.NV = int(padded.sum().item()) if padded.numel() else 0
        self.G = int(big.sum().item())
        perm = torch.full((max(self.NV, 1),), -1, dtype=torch.int32, device=dev)
        rank = torch.arange(n_sel, device=dev) - start[gid]
        if self.G:
            vpos = (vstart[big_index[gid].clamp(min=0)] + rank)[in_big]
        else:                                                            # no degree value fills a tile: every row is a rest row
            vpos = torch.zeros(0, dtype=torch.long, device=dev)
        tile_image = torch.repeat_interleave(torch.arange(self.G, device=dev, dtype=torch.int32), padded // TILE)
        if TILE_ORDER in ("interleave", "descending") and self.NV:
            # The tiles in ascending degree make the whole device multiply-bound first (few edges per tile) and gather-bound last.
            # Interleaved -- lowest, highest, second lowest, ... -- every workgroup alternates between the two kinds, the memory
            # system sees an even load from start to end, and two weight images are live at a time instead of one.
            nt = self.NV // TILE
            k = torch.arange(nt, device=dev)
            src_tile = torch.where(k % 2 == 0, k // 2, nt - 1 - k // 2) if TILE_ORDER == "interleave" else nt - 1 - k   # new tile k <- ascending tile src_tile[k]
            new_of = torch.empty(nt, dtype=torch.long, device=dev)
            new_of[src_tile] = k
            vpos = new_of[vpos // TILE] * TILE + vpos % TILE
            tile_image = tile_image[src_tile]
        perm[vpos] = order[in_big].to(torch.int32)
        self.perm = perm[:self.NV].contiguous()
        self.tile_image = tile_image.contiguous()
        self.group_first_row = order[start[big]] if self.G else order[:0]           # a row of each group (its scalers = the group's)
        self.group_degree = ud[big]
        rest = order[~in_big]
        self.dropped_rest = None
        if drop_rest:
            self.dropped_rest, rest = rest, rest[:0]
        self.NR = int(rest.numel())
        self.NRp = (self.NR + TILE_REST - 1) // TILE_REST * TILE_REST
        perm_rest = torch.full((max(self.NRp, 1),), -1, dtype=torch.int32, device=dev)
        perm_rest[:self.NR] = rest.to(torch.int32)
        self.perm_rest = perm_rest[:self.NRp].contiguous()
        self.rest_rows = rest
        # virtual position of every node in the (NV + NRp)-row aggregate buffer (int32: 4 bytes per node resident, not 8 -- VERDICT r5 item 6)
        vmap = torch.full((V,), -1, dtype=torch.int32, device=dev)           # (-1: a row of another block)
        vmap[order[in_big]] = vpos.to(torch.int32)
        vmap[rest] = (self.NV + torch.arange(self.NR, device=dev)).to(torch.int32)
        n_seg = hs.n_seg if hs.n_heavy > 0 else 0
        # (the two-kernel grouped path's work list -- 16 bytes per node -- is built when that path first asks for it: `items`, `heavy_out`)
        self._whole = row_range is None and with_heavy and extra_rows is None and not drop_rest
        self._items, self._graph_ref, self._hs_rows = None, __import__("weakref").ref(graph), (hs.heavy_rows if hs.n_heavy > 0 else None)
        self._vmap, self._n_seg, self._split = vmap, n_seg, None
        self.rows = self.NV + self.NRp
        self._rest_scales = {}
        self._fused, self._rest_items, self._vmap32, self._perm_all, self._rest_items_node, self._ones_rows = None, None, None, None, None, None
        self._edge_split = None
        self._deg, self._csr = deg, csr

    @property
    def items(self):
        """The graph's work list with the plan's output rows (whole-graph plans; None for a row block's: those serve the one-kernel layer
        only -- group rows + rest_items): the gather of the two-kernel grouped path and of the training forward."""
        if not self._whole:
            return None
        if self._items is None:
            items = self._graph_ref().work_items().clone()
            n_seg = self._n_seg
            items[n_seg:, 0] = self._vmap[items[n_seg:, 0].long()]          # whole-row records: `row` = output row (nothing else uses it)
            self._items = (items.contiguous(), self._vmap[self._hs_rows.long()].contiguous() if self._hs_rows is not None else None)
        return self._items[0]

    @property
    def heavy_out(self):
        if not self._whole:
            return None
        self.items
        return self._items[1]

    def fused_tables(self):
        """(tile_desc, tile_ids, n_records) of pna_fused_degree_f32 (include/pna_amd.h), built once per graph on the device:
        tile_desc[w] = {first record, in-degree, weight image, 0} of the 16-row block w of the virtual order; tile_ids = the
        TILE-MAJOR edge list -- record (first + e)[i] = source row of the e-th in-edge of the block's i-th row (a padding row
        repeats the block's first row), max(4, round_up(D, 4)) records per block, the ones past D copies of record D - 1."""
        if self._fused is None:
            dev = self.perm.device
            if self.NV == 0:
                self._fused = (torch.zeros(0, 4, dtype=torch.int32, device=dev), torch.zeros(4, 16, dtype=torch.int32, device=dev), 4)
                return self._fused
            nt = self.NV // 16
            p16 = self.perm.view(nt, 16).long()
            first = p16[:, :1]
            live = first[:, 0] >= 0                                          # (a block of padding rows only: D = 0, records of row 0)
            rows = torch.where(p16 >= 0, p16, first.clamp(min=0))
            D = torch.where(live, self._deg[rows[:, 0]], torch.zeros_like(first[:, 0]))
            nrec = ((D + 3) // 4 * 4).clamp(min=4)
            rec0 = torch.cumsum(nrec, 0) - nrec
            total = int(nrec.sum().item())
            if total * 64 >= (1 << 32):
                self._fused = False                                          # the kernel addresses records with 32-bit byte offsets
                return self._fused
            tile = torch.repeat_interleave(torch.arange(nt, device=dev), nrec)
            e = torch.arange(total, device=dev) - rec0[tile]
            e = torch.minimum(e, (D[tile] - 1).clamp(min=0))
            rp = self._csr.rowptr.long()
            pos = rp[rows[tile]] + e[:, None]                                # (total, 16) positions in col[]
            empty = (D[tile] == 0)[:, None]
            n_col = self._csr.col.numel()
            ids = self._csr.col[pos.clamp(max=max(n_col - 1, 0))] if n_col else torch.zeros_like(pos, dtype=torch.int32)
            ids = torch.where(empty, torch.zeros_like(ids), ids).to(torch.int32).contiguous()
            image = torch.repeat_interleave(self.tile_image.long(), TILE // 16)
            desc = torch.stack([rec0, D, image, torch.zeros_like(D)], dim=1).to(torch.int32).contiguous()
            self._fused = (desc, ids, total)
        return self._fused

    def fused_balance(self, G):
        """(tile_desc, row_perm, src) of pna_fused_degree_f32 for a grid of G workgroups with its 64-row tiles LOAD-BALANCED (round 5), or
        None when the balance is off / does not apply.  The persistent kernel gives workgroup w the tiles at positions w, w + G, w + 2 G, ..;
        the plan lists tiles in ascending degree, and the number of tiles is not a multiple of G: the LAST round is partial and holds
        the most expensive tiles (hub-side degrees: 6-10x a mean tile), handed to the first nt % G workgroups on top of a full share --
        on the benchmark graph the slowest workgroup carries 7.5 % more than the mean (phase timers: 1.88 M cycles against 1.72 M).
        Here (FUSED_BALANCE = "lpt"): the partial round holds the nt % G CHEAPEST tiles instead, and inside every full round the tiles are
        dealt longest-first to the workgroups with the least accumulated cost, rounds processed from the heaviest down -- each workgroup
        still gets one tile per round in ascending degree over time, so the device stays degree-synchronous (one weight image live per
        L2: the id-major order of tools/tile_order_exp.py lost exactly that).  The SAME tiles in another order: no row changes its
        16-row block, the statistics and outputs are bit-identical.  cost(tile) = in-degree + FUSED_TILE_COST (the multiply / epilogue /
        control share in edge units).  One host round trip per (plan, G)."""
        if FUSED_BALANCE not in ("lpt", "cheap_last", "dynamic") or self.NV == 0:
            return None
        tabs = self.fused_tables()
        if tabs is False:
            return None
        key = (int(G), FUSED_BALANCE, float(FUSED_TILE_COST), int(FUSED_DYNAMIC_TAIL))
        hit = self.__dict__.setdefault("_fused_bal", {}).get(key)
        if hit is not None:
            return hit
        import numpy as np
        desc = tabs[0]
        nt = self.NV // 64
        G = int(G)
        D = desc.view(nt, 4, 4)[:, :, 1].max(dim=1).values.cpu().numpy().astype(np.float64)     # (a tile's blocks: one degree, or 0 = padding)
        cost = D + float(FUSED_TILE_COST)
        rem, n_full = nt % G, nt // G
        seq = np.arange(nt)
        if FUSED_BALANCE == "dynamic":
            # the list the DYNAMIC schedule walks (pna_fused_degree_args.tile_counter: every workgroup claims its next tile when it
            # is ready for one): the G most expensive tiles FIRST, one per workgroup while everybody starts anyway (longest first:
            # none of them can become the tail), then ascending degree (the device stays degree-synchronous), and the G cheapest
            # tiles LAST -- whoever is still running at the end runs a tile of a few microseconds.  FUSED_DYNAMIC_TAIL x G of them,
            # cheapest last: a workgroup holds claims four tiles ahead, so the tail it can be left with is its last four tiles
            tail = int(FUSED_DYNAMIC_TAIL) * G
            if nt > 2 * (G + tail):
                order = np.argsort(cost, kind="stable")                                       # ascending cost (= plan order up to padding tiles)
                seq = np.concatenate([order[nt - G:][::-1], order[tail:nt - G], order[:tail][::-1]])
            src = seq.copy()
        elif rem:
            seq = np.concatenate([np.arange(rem, nt), np.arange(rem)])                       # the partial round: the cheapest tiles
        src = seq.copy()
        if FUSED_BALANCE == "lpt" and n_full > 0 and G > 1:
            load = np.zeros(G)
            if rem:
                load[:rem] += cost[seq[n_full * G:]]
            for r in range(n_full - 1, -1, -1):
                tiles = seq[r * G:(r + 1) * G]
                tiles = tiles[np.argsort(-cost[tiles], kind="stable")]
                wgs = np.argsort(load, kind="stable")
                src[r * G + wgs] = tiles
                load[wgs] += cost[tiles]
        src_t = torch.from_numpy(src).to(desc.device)
        out = (desc.view(nt, 4, 4)[src_t].reshape(-1, 4).contiguous(), self.perm.view(nt, 64)[src_t].reshape(-1).contiguous(), src_t)
        cache = self.__dict__["_fused_bal"]
        if len(cache) >= 4:
            cache.clear()
        cache[key] = out
        return out

    def dw_tables(self, n_wgs):
        """Tables of pna_posttrans_dw_grouped_f32 (the weight gradient over the group rows in plan order), once per plan and grid:
        (tile_group [nt], wg_range [n_wgs][2] -- equally long contiguous tile ranges --, wg_entry [n_wgs], entry_group [n_entries]):
        a workgroup writes one partial product per run of equal degree groups inside its range; the entries are numbered in tile
        order."""
        hit = self.__dict__.get("_dw_tables")
        if hit is not None and hit[0] == int(n_wgs):
            return hit[1]
        dev = self.perm.device
        nt = self.NV // TILE
        n_wgs = int(n_wgs)
        tg = self.tile_image.to(torch.int32).contiguous()
        b = (torch.arange(n_wgs + 1, device=dev, dtype=torch.long) * nt) // n_wgs
        lo, hi = b[:-1], b[1:]
        change = torch.zeros(nt, dtype=torch.bool, device=dev)
        if nt > 1:
            change[1:] = tg[1:] != tg[:-1]
        starts = torch.unique(torch.cat([lo[hi > lo], torch.nonzero(change).flatten()]))        # sorted: first tile of every entry
        wg_entry = torch.searchsorted(starts, lo.clamp(max=max(nt - 1, 0))).to(torch.int32)
        out = (tg, torch.stack([lo, hi], dim=1).to(torch.int32).contiguous(), wg_entry.contiguous(),
               tg[starts].contiguous(), int(starts.numel()))
        self.__dict__["_dw_tables"] = (n_wgs, out)
        return out

    def tiles_by_group(self):
        """(row_perm, tile_group) of the plan's tiles listed SORTED BY degree group (stable): int32 [NV], int32 [NV / 128] -- what
        pna_project_grouped_f32 walks, its workgroups refilling their weight image only when the group changes (the plan's own tile order
        may interleave low and high degrees)."""
        hit = self.__dict__.get("_tiles_by_group")
        if hit is None:
            order = torch.sort(self.tile_image.long(), stable=True).indices
            hit = self.__dict__["_tiles_by_group"] = (self.perm.view(-1, TILE)[order].reshape(-1).contiguous(),
                                                      self.tile_image[order].to(torch.int32).contiguous())
        return hit

    def group_scaler_values(self, row_scales):
        """(G, S) fp32: the value of every scaler on the rows of every degree group (None = the identity scaler: 1)."""
        cols = [torch.ones(self.G, dtype=torch.float32, device=self.perm.device) if rs is None else rs[self.group_first_row].to(torch.float32)
                for rs in row_scales]
        return torch.stack(cols, dim=1)

    def vmap32(self):
        """int32 [V]: row of the plan-ordered aggregate buffer that holds node v (pna_segreduce_args.out_row_of)."""
        return self._vmap

    def node_of_rows(self):
        """int32 [rows]: node of every row of the plan-ordered buffer, -1 for padding rows (pna_segreduce_bwd_args.stat_node_of)."""
        hit = self.__dict__.get("_node_of_rows")
        if hit is None:
            hit = self.__dict__["_node_of_rows"] = torch.cat([self.perm, self.perm_rest]).contiguous()
        return hit

    def perm_all(self):
        """int32 [rows]: node of every row of the plan-ordered buffer (degree tiles, then the rest), padding rows -> node 0."""
        if self._perm_all is None:
            self._perm_all = torch.cat([self.perm, self.perm_rest]).clamp(min=0).contiguous()
        return self._perm_all

    def rest_items(self, graph):
        """(work list, heavy_out, heavy schedule) of the rows no degree group holds, output rows counted from the start of the rest
        region: the gather of the one-kernel layer's leftover rows (pna_fused_degree_f32 takes the group rows).  The hub rows are
        cut into REST_SEG_LEN-edge segments here (the graph's own schedule: 128): this launch has only a few thousand items, and
        a lane group walks its segment four edges per memory round trip -- shorter segments, more of them in flight."""
        if self._rest_items is None:
            hs = graph.heavy_schedule(seg_len=REST_SEG_LEN)
            it = graph.work_items(seg_len=REST_SEG_LEN)
            n_seg = hs.n_seg if hs.n_heavy > 0 else 0
            rows = it[n_seg:].clone()
            rows[:, 0] = self._vmap[rows[:, 0].long()]
            light = rows[rows[:, 0] >= self.NV]                    # (rows of other blocks: -1)
            light[:, 0] -= self.NV
            if not self.with_heavy:                                # a block without the hub rows: whole-row records only
                self._rest_items = (light.contiguous(), None, None)
                return self._rest_items
            items = torch.cat([it[:n_seg], light], dim=0).contiguous() if n_seg else light.contiguous()
            hout = (self._vmap[hs.heavy_rows.long()] - self.NV).contiguous() if hs.n_heavy > 0 else None
            self._rest_items = (items, hout, hs)
        return self._rest_items

    def edge_split(self):
        """(edges of the group rows, edges of the rest rows), counted once per plan."""
        if self._edge_split is None:
            live = self.perm[self.perm >= 0].long()
            e_g = int(self._deg[live].sum().item())
            self._edge_split = (e_g, int(self._deg[self.rest_rows].sum().item()))
        return self._edge_split

    def rest_overlap_applies(self, F):
        """Whether the rest-row launches of a layer of F input features should run beside the one-kernel layer instead of behind it
        (see FUSED_SPARE_WGS)."""
        if FUSED_SPARE_WGS <= 0 or not self.NR or self.NV < FUSED_OVERLAP_MIN_ROWS or F < FUSED_OVERLAP_MIN_F:
            return False
        e_g, e_r = self.edge_split()
        return e_r <= FUSED_OVERLAP_MAX_REST_EDGES * e_g

    def ones_rows(self):
        """float32 [NV] of ones: the per-row factor of a layer without graph norm."""
        if self._ones_rows is None:
            self._ones_rows = torch.ones(self.NV, dtype=torch.float32, device=self.perm.device)
        return self._ones_rows

    def rest_items_by_node(self, graph):
        """rest_items() for a gather that needs the NODE of every row (the tower layers' destination term): the work list keeps
        the node in column 0 and the output row comes from a per-node table (pna_segreduce_args.out_row_of), counted from the start
        of the rest region."""
        if self._rest_items_node is None:
            _, hout, hs = self.rest_items(graph)
            it = graph.work_items(seg_len=REST_SEG_LEN)
            n_seg = hs.n_seg if hs.n_heavy > 0 else 0
            rows = it[n_seg:]
            light = rows[self._vmap[rows[:, 0].long()] >= self.NV]
            items = torch.cat([it[:n_seg], light], dim=0).contiguous() if n_seg else light.contiguous()
            self._rest_items_node = (items, (self._vmap - self.NV).contiguous(), hout, hs)
        return self._rest_items_node

    def split_items(self, graph):
        """(interior, boundary) work lists of a sharded graph (HaloGraph.split_work_lists: the rows that read only local
        sources | everything else incl. the hub segments) with the plan's output rows: the halo overlap of
        functional.aggregate, writing in degree order."""
        if self._split is None:
            _, items_in, items_bd = graph.split_work_lists()

            def remap(items, n_seg):
                it = items.clone()
                it[n_seg:, 0] = self._vmap[it[n_seg:, 0].long()]
                return it.contiguous()
            self._split = (remap(items_in, 0), remap(items_bd, self._n_seg))
        return self._split

    def rest_scales(self, key, row_scales):
        """The per-row scalers of the rest rows in their virtual order (padding: 0), cached per scaler set."""
        hit = self._rest_scales.get(key)
        if hit is None:
            out = []
            for rs in row_scales:
                if rs is None:
                    out.append(None)
                else:
                    v = torch.zeros(max(self.NRp, 1), dtype=torch.float32, device=rs.device)
                    v[:self.NR] = rs[self.rest_rows]
                    out.append(v[:self.NRp].contiguous())
            hit = self._rest_scales[key] = out
        return hit


def agg_pitch(K):
    """Row pitch (floats) of the plan-ordered aggregate.  The contraction reads a row as 128-byte strips, one per chunk of 32
    columns; with packed rows of 4F = 300 floats (1200 bytes) 7 strips in 8 straddle two 128-byte lines.  Measured on the
    one-block kernel, 1 M rows (tools/x3_pitch_time.py [removed in round 5: git history], one run, best of 5 x 20 launches): pitch 300 / 304 / 320 / 352 floats
    -> 0.457 / 0.434 / 0.422 / 0.419 ms (the three-block kernel, bound by its matrix work, does not care: 0.75 ms at all four)."""
    a = max(1, int(AGG_ALIGN))
    return (K + a - 1) // a * a


def plan_of(graph):
    plan = graph.__dict__.get("_pna_amd_degree_plan")
    if plan is None:
        plan = graph.__dict__["_pna_amd_degree_plan"] = DegreePlan(graph)
    return plan


def combined_images(weight, K, row_scales, plan):
    """Packed bf16x3 images of W_D = sum_s s_s(D) W_s for every group of `plan` (one buffer, image_stride bytes apart), cached
    on the weight per (version, scaler tensors, plan).  out_dim <= 80: ONE pack call over a (G * 80, K) matrix whose 80-column
    blocks are the images; 80 < out_dim <= 128: one pack call per image (the packer cuts wider matrices into 80-column blocks)."""
    key = (weight._version, weight.data_ptr(), str(weight.device), tuple(weight.shape), K, plan.serial, plan.G,
           tuple(None if rs is None else (rs.data_ptr(), rs._version) for rs in row_scales))
    hit = getattr(weight, "_pna_amd_group_img", None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    N, G = weight.shape[0], plan.G
    with torch.no_grad():
        wc = None
        for s, rs in enumerate(row_scales):
            ws = weight[:, s * K:(s + 1) * K]
            term = ws.unsqueeze(0).expand(G, N, K) if rs is None else rs[plan.group_first_row].view(G, 1, 1) * ws.unsqueeze(0)
            wc = term.clone() if wc is None else wc + term          # scaler order, like the reference's blocks
        BW = 80 if N <= 80 else 128
        w_all = torch.zeros(G, BW, K, dtype=torch.float32, device=weight.device)
        w_all[:, :N] = wc
    L = _lib.lib()
    nh = ctypes.c_int64(0)
    if BW == 80:
        nb = L.pna_posttrans_x3_packed_bytes(K, G * 80, 1, 0, ctypes.byref(nh))
        img = torch.empty(nb // 4, dtype=torch.float32, device=weight.device)
        rc = L.pna_posttrans_x3_pack_f32(_lib.dev_ptr(w_all.view(G * 80, K), torch.float32, "weight"), K, G * 80, K, 1, 0,
                                         _lib.dev_ptr(img, torch.float32, "w_img"), None, _lib.stream_ptr(weight.device))
        _lib.check(rc, "pna_posttrans_x3_pack_f32")
        stride = nb // G
    else:
        stride = L.pna_posttrans_x3_packed_bytes(K, 128, 1, 0, ctypes.byref(nh))
        img = torch.empty(G * stride // 4, dtype=torch.float32, device=weight.device)
        for i in range(G):
            rc = L.pna_posttrans_x3_pack_f32(_lib.dev_ptr(w_all[i], torch.float32, "weight"), K, 128, K, 1, 0,
                                             _lib.dev_ptr(img[i * stride // 4:], torch.float32, "w_img"), None, _lib.stream_ptr(weight.device))
            _lib.check(rc, "pna_posttrans_x3_pack_f32")
    try:
        weight._pna_amd_group_img = (key, img, stride)
    except AttributeError:
        pass
    return img, stride


def fused_tower_images(weight, F, row_scales, plan, x3=False):
    """fused_images() for the tower mode of pna_fused_degree_f32: `weight` (N, S * 5F) in scaler blocks [4F aggregators | F self
    panel (block 0)] (functional._tower_collapsed_weights with one tower); the pack kernel appends the chunks of the two node
    panels."""
    return fused_images(weight, F, row_scales, plan, tower=True, x3=x3)


def fused_arith():
    """The arithmetic of the one-kernel layers' contraction (include/pna_amd.h PNA_FD_ARITH_*): `guarded` (default) = two fp16 terms /
    three partial products with the floor-error guard -- tiles whose outputs it cannot certify are computed again in bf16 x 3 by a second
    launch --; `bf16x3` = three bf16 terms / six products everywhere (rounds 3-4; also what PNA_AMD_POSTTRANS=bf16x3 selects: ADVICE r5);
    `fp16x2` = round 5's unguarded form (opt-in: operands of moderate dynamic range only)."""
    from . import _lib, ops
    name = FUSED_ARITH
    if name == "guarded" and ops.POSTTRANS_ARITH == "bf16x3":
        name = "bf16x3"
    try:
        return {"guarded": _lib.FD_ARITH_GUARDED, "bf16x3": _lib.FD_ARITH_X3, "fp16x2": _lib.FD_ARITH_H2}[name]
    except KeyError:
        raise ValueError(f"PNA_AMD_FUSED_ARITH must be guarded, bf16x3 or fp16x2 (got {name!r})") from None


STANDARD_AGGREGATORS = ("mean", "max", "min", "std")
# the one-kernel layer computes [mean | max | min | std]; a layer's aggregator maps onto one of those slots, `sum` onto the mean's with the
# degree as a factor (sum = D x mean; the degree is a property of the weight image W_D like the scalers are) -- round 6, VERDICT r5 item 4
_AGG_SLOT = {"mean": (0, False), "sum": (0, True), "max": (1, False), "min": (2, False), "std": (3, False)}


def aggregators_fusable(aggregators):
    """Whether the one-kernel layer can serve this aggregator list: distinct names out of mean / sum / max / min / std."""
    a = tuple(aggregators)
    return 0 < len(a) == len(set(a)) and all(x in _AGG_SLOT for x in a)


def _virtual_weight(weight, F, aggregators, scale, plan, rows, feats=None):
    """(w_ref, scale') for pna_fused_pack_f32: the rows `rows` of `weight` (N, S * A * F), layer column order [scaler][aggregator][feature], as a
    weight over the kernel's four statistics of the features `feats` = (f0, f1), (Np, S' * 4 Fp) with Fp = f1 - f0 and S' = S (+ S more
    blocks when `sum` is among the aggregators: the same scaler value times the group's degree, against the mean's columns).  The standard
    list over all rows and features: the weight itself."""
    aggs = tuple(aggregators)
    c0, c1 = rows if rows is not None else (0, weight.shape[0])
    f0, f1 = feats if feats is not None else (0, F)
    if aggs == STANDARD_AGGREGATORS and (f0, f1) == (0, F):
        return weight[c0:c1], scale
    G, S = scale.shape
    A, Fp = len(aggs), f1 - f0
    has_sum = "sum" in aggs
    W = weight[c0:c1].detach()
    w = torch.zeros(c1 - c0, (2 * S if has_sum else S) * 4 * Fp, dtype=torch.float32, device=weight.device)
    for s_ in range(S):
        for j, a in enumerate(aggs):
            slot, times_deg = _AGG_SLOT[a]
            blk = (S + s_) if times_deg else s_
            w[:, (blk * 4 + slot) * Fp:(blk * 4 + slot + 1) * Fp] = W[:, (s_ * A + j) * F + f0:(s_ * A + j) * F + f1]
    if has_sum:
        scale = torch.cat([scale, scale * plan.group_degree.to(torch.float32).view(G, 1)], dim=1)
    return w, scale.contiguous()


def virtual_layer_weight(weight, F, aggregators, S):
    """The layer's posttrans weight (N, S * A * F) over the FOUR standard statistics (+ a fifth block, sum = D x mean, when `sum` is among
    the aggregators), for the contraction of the rest rows -- their gather is the hand-scheduled kernel's: the standard four only --:
    (N, S * Kv), scaler blocks of Kv = 4F or 5F columns [mean | max | min | std (| sum)].  Cached on the weight."""
    aggs = tuple(aggregators)
    key = ("virt", aggs, S, weight._version, weight.data_ptr(), tuple(weight.shape))
    hit = getattr(weight, "_pna_amd_virtual", None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    A = len(aggs)
    Kv = (5 if "sum" in aggs else 4) * F
    with torch.no_grad():
        W = weight.detach()
        w = torch.zeros(W.shape[0], S * Kv, dtype=torch.float32, device=W.device)
        for s_ in range(S):
            for j, a in enumerate(aggs):
                slot = 4 if a == "sum" else _AGG_SLOT[a][0]
                w[:, s_ * Kv + slot * F:s_ * Kv + (slot + 1) * F] = W[:, (s_ * A + j) * F:(s_ * A + j + 1) * F]
    try:
        weight._pna_amd_virtual = (key, w, Kv)
    except AttributeError:
        pass
    return w, Kv


def fused_images(weight, F, row_scales, plan, tower=False, x3=False, rows=None, aggregators=STANDARD_AGGREGATORS, feats=None):
    """Packed images of W_D = sum_s s_s(D) W_s for pna_fused_degree_f32 (K in the kernel's chunk order), one per degree group,
    cached on the weight like combined_images.  The combination and the operand split (x3=False: two fp16 terms behind per-column
    power-of-two scales + the tail of column scales and guard thresholds; x3=True: three bf16 terms) happen in the pack kernel
    (pna_fused_pack_f32) from the (G, S) matrix of the groups' scaler values.  rows = (c0, c1): the images of output columns [c0, c1) only
    (a column panel of a layer wider than one launch takes); feats = (f0, f1): over the statistics of the features [f0, f1) only (a
    feature panel of a layer with more features than one launch gathers); aggregators: the layer's list (see _virtual_weight)."""
    N, G, S = weight.shape[0], plan.G, len(row_scales)
    aggs = tuple(aggregators)
    key = ("fused", tower, x3, rows, feats, aggs, weight._version, weight.data_ptr(), str(weight.device), tuple(weight.shape), F, plan.serial, G,
           tuple(None if rs is None else (rs.data_ptr(), rs._version) for rs in row_scales))
    attr = "_pna_amd_fused_img"
    cache = getattr(weight, attr, None)                     # {plan serial: (key, image, stride)}: the block plans of a pipelined run
    ckey = (tower, x3, rows, feats, aggs, plan.serial)
    hit = cache.get(ckey) if isinstance(cache, dict) else None      # share one weight, each with its own groups
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    L = _lib.lib()
    Np = N if rows is None else rows[1] - rows[0]
    Fp = F if feats is None else feats[1] - feats[0]
    stride = L.pna_fused_image_bytes(Fp, Np, 1 if tower else 0, 1 if x3 else 0)
    if stride <= 0:
        raise RuntimeError(f"pna_fused_degree: unsupported shape F={Fp}, N={Np}")
    with torch.no_grad():
        scale = torch.ones(G, S, dtype=torch.float32, device=weight.device)
        for s, rs in enumerate(row_scales):
            if rs is not None:
                scale[:, s] = rs[plan.group_first_row]
        scale = scale.contiguous()
        if tower:
            w, sc = weight.detach(), scale
        else:
            w, sc = _virtual_weight(weight.detach(), F, aggs, scale, plan, rows, feats)
    img = torch.empty(G * stride // 4, dtype=torch.float32, device=weight.device)
    rc = L.pna_fused_pack_f32(_lib.dev_ptr(w, torch.float32, "weight"), w.stride(0), Np, Fp, sc.shape[1], _lib.dev_ptr(sc, torch.float32, "scale"),
                              G, _lib.dev_ptr(img, torch.float32, "w_img"), 1 if tower else 0, 1 if x3 else 0, _lib.stream_ptr(weight.device))
    _lib.check(rc, "pna_fused_pack_f32")
    try:
        if not isinstance(cache, dict):
            cache = {}
            weight._pna_amd_fused_img = cache
        if len(cache) >= 64:                                 # (plans of dropped graphs: start over rather than grow)
            cache.clear()
        cache[ckey] = (key, img, stride)
    except AttributeError:
        pass
    return img, stride


def _column_panels(F, N):
    L = _lib.lib()
    if N < 4 or L.pna_fused_degree_image_bytes(F, 4) <= 0:  # (no instantiation gathers F features)
        return None
    if L.pna_fused_degree_image_bytes(F, N) > 0:
        return [(0, N)]
    widest = 128 if L.pna_fused_degree_image_bytes(F, 128) > 0 else 80
    n = (N + widest - 1) // widest
    if n > FUSED_MAX_PANELS:
        return None
    step = ((N + n - 1) // n + 3) // 4 * 4
    panels = [(c, min(c + step, N)) for c in range(0, N, step)]
    if any(c1 - c0 < 4 or L.pna_fused_degree_image_bytes(F, c1 - c0) <= 0 for c0, c1 in panels):
        return None
    return panels


def fused_panels(F, N):
    """The launches [(f0, f1, c0, c1), ..] the one-kernel layer runs a layer of F features and N outputs in, or None.  One launch when
    pna_fused_degree_f32 is instantiated for (F, N).  A layer WIDER than an instantiation: up to FUSED_MAX_PANELS output-column panels
    [c0, c1) of the widest instantiation F has (128 columns with exactly two full feature blocks per gather pass, else 80), cut evenly
    on multiples of 4 -- every panel its own gather.  A layer with MORE FEATURES than an instantiation gathers (81 <= F <= 96: the
    statistics of different features never meet before the contraction): feature panels [0, 64) and [64, F), the second launch adding
    its share to the first's partial sums (pna_fused_degree_args.pre_add) and applying the epilogue -- round 6, hidden sizes 90 / 95 of the
    reference's README.  Launch order = list order: within a column panel the feature panels follow one another."""
    cols = _column_panels(F, N)
    if cols is not None:
        return [(0, F, c0, c1) for c0, c1 in cols]
    if not 81 <= F <= 96:
        return None
    out = []
    for f0, f1 in ((0, 64), (64, F)):
        cols = _column_panels(f1 - f0, N)
        if cols is None:
            return None
        out.append([(f0, f1, c0, c1) for c0, c1 in cols])
    if len(out[0]) + len(out[1]) > 2 * FUSED_MAX_PANELS:
        return None
    return out[0] + out[1]            # (all of the first feature panel's launches, then the second's: those read the partial sums)


def bind_fused_arith(a, keep, weight, F, row_scales, plan, tower, device, verification=False, rows=None, aggregators=STANDARD_AGGREGATORS, feats=None):
    """Fill the arithmetic half of a pna_fused_degree_args block: the images of the arithmetic fused_arith() selects and, for the
    guarded form, the plan's hand-over workspace (one per (plan, stream), like the tile counters: launches on one stream are ordered).
    `keep`: a list that keeps the tensors alive.  verification: the agg_out instantiation (fp16 x 2, unguarded)."""
    arith = _lib.FD_ARITH_H2 if verification else fused_arith()
    a.arith = arith
    if arith != _lib.FD_ARITH_X3:
        img, stride = fused_images(weight, F, row_scales, plan, tower=tower, x3=False, rows=rows, aggregators=aggregators, feats=feats)
        a.w_img, a.image_stride = _lib.dev_ptr(img, torch.float32, "w_img"), stride
        keep.append(img)
    if arith != _lib.FD_ARITH_H2:
        img3, stride3 = fused_images(weight, F, row_scales, plan, tower=tower, x3=True, rows=rows, aggregators=aggregators, feats=feats)
        a.w_img_x3, a.image_stride_x3 = _lib.dev_ptr(img3, torch.float32, "w_img_x3"), stride3
        keep.append(img3)
    if arith == _lib.FD_ARITH_GUARDED:
        ws = guard_workspace(plan, device)
        a.guard_ws, a.guard_ws_bytes = _lib.dev_ptr(ws, torch.int32, "guard_ws"), ws.numel() * 4
        keep.append(ws)
    return arith


def guard_workspace(plan, device):
    """int32 tensor of pna_fused_degree_guard_bytes(plan.NV) bytes for the current stream: words [0..1] the running call's, [2] the tiles
    handed over to the bf16 x 3 launch since the caller last zeroed it, [3] the calls (guard_stats)."""
    ckey = (str(device), int(torch.cuda.current_stream(device).cuda_stream))
    wss = plan.__dict__.setdefault("_guard_ws", {})
    ws = wss.get(ckey)
    if ws is None:
        if len(wss) > 16:
            wss.clear()
        nb = _lib.lib().pna_fused_degree_guard_bytes(plan.NV)
        ws = wss[ckey] = torch.zeros((nb + 3) // 4, dtype=torch.int32, device=device)
    return ws


def guard_stats(plan, device, reset=False):
    """(tiles handed over to the bf16 x 3 launch, guarded calls) of this plan on `device` since the last reset, over ALL its workspaces -- one
    per stream a call was bound on (a captured hipGraph's calls own the capture stream's) -- one host read per workspace."""
    t = c = 0
    for (dev, _), ws in plan.__dict__.get("_guard_ws", {}).items():
        if dev == str(device):
            a, b = (int(v) for v in ws[2:4].tolist())
            t, c = t + a, c + b
            if reset:
                ws[2:4].zero_()
    return t, c


# Load balance of the one-kernel layer (DegreePlan.fused_balance): "dynamic" (tiles claimed from a device counter, heaviest first / cheapest
# last) | "lpt" | "cheap_last" (static schedules over a cost-balanced list) | "off" (the plan's ascending order, static)
FUSED_BALANCE = os.environ.get("PNA_AMD_FUSED_BALANCE", "dynamic")
FUSED_MAX_PANELS = 3       # output-column panels of one layer (fused_panels): every panel gathers again -- beyond three the ordinary path's ONE gather wins
FUSED_ARITH = os.environ.get("PNA_AMD_FUSED_ARITH", "guarded")   # guarded | bf16x3 | fp16x2 (fused_arith)
FUSED_DYNAMIC_TAIL = int(os.environ.get("PNA_AMD_FUSED_DYNAMIC_TAIL", "4"))   # "dynamic": this many x G of the cheapest tiles end the list
FUSED_TILE_COST = float(os.environ.get("PNA_AMD_FUSED_TILE_COST", "10"))   # a tile's constant cost in edge units (multiply + epilogue + control)
OUT_PITCH_ALIGN = int(os.environ.get("PNA_AMD_OUT_PITCH_ALIGN", "32"))   # floats: row pitch of the one-kernel layers' own output (functional.out_pitch)
REST_SEG_LEN = 128         # edges per hub-row segment in the rest launch of the one-kernel layer (see DegreePlan.rest_items)
REST_ROWS_PER_GROUP = 1    # work items per lane group in that launch (the full-graph gather: 4): a few thousand items must spread over 256 CUs
TOWERS = True              # the tower layers (PNALayer) through the degree-grouped contraction with collapsed posttrans / mixing weights
FUSED = True               # gather + contraction in ONE kernel (pna_fused_degree_f32) where it applies; False: the two-kernel grouped path
MAX_REST_FRACTION = 0.5    # more rest rows than this: the grouping is overhead, the ordinary path takes the graph
FUSED_HALO_MAX_INTERIOR = 0.5   # shards: below this fraction of rows without remote sources, exchange first and run ONE kernel
# The rest-row launches BESIDE the one-kernel layer (functional.run_fused_call): the persistent kernel leaves this
# many of its 2-per-CU workgroups out and the rest rows' gather / finalize / contraction run on a second stream in the slots left
# free (measured on the benchmark graph: 24 is too few -- the chain of small launches then outlasts the kernel -- 32..40 hide it;
# round 5: the kernel got 6 % faster, the chain did not -- at 32 it outlasts the kernel on some boxes (step 0.745-0.762), at 40 it does not (0.745-0.748)).
FUSED_SPARE_WGS = int(os.environ.get("PNA_AMD_FUSED_SPARE_WGS", "40"))   # (0: the rest rows behind the kernel, as before ABI 17)
FUSED_OVERLAP_MIN_ROWS = 1 << 19   # smaller graphs: the kernel is too short to hide a chain of launches confined to a few CUs
FUSED_OVERLAP_MIN_F = 64           # ... and so it is with few features (the chain is latency, the kernel's time follows F); measured: 75, 128
FUSED_OVERLAP_MAX_REST_EDGES = 1.0 / 12   # ... and so it is when the rest rows hold more than this fraction of the group rows' edges


def out_pitch_floats(N):
    """Row pitch (floats) of the output the degree-grouped layers allocate (functional.out_pitch: whole 128-byte lines), at least the
    80 / 128-column block of the grouped contraction: what the kernels' 32-bit row offsets must cover (the guard used to
    assume 80 floats while the one-kernel layer allocates 96 for 65 <= N <= 80 -- a graph of 11.2-13.4 M nodes passed the guard and was
    then refused by pna_fused_degree_f32)."""
    a = max(4, int(OUT_PITCH_ALIGN))
    return max((N + a - 1) // a * a, 80 if N <= 80 else 128 if N <= 128 else N)


def fused_applies(graph, x, F, N, aggregators=STANDARD_AGGREGATORS):
    """Whether pna_fused_degree_f32 serves this call (whole-graph inference path already chosen by `applies`): a shape it is
    instantiated for -- N possibly in column panels -- and a unit-stride, 4-byte aligned source table.  Since round 4 the rows are read
    through 64-bit lane addresses and no read leaves a row (the last feature block's window slides back to end at F): any pitch >= F -- a
    CONTIGUOUS (V, F) tensor included -- and tables beyond 4 GiB / 2^24 rows (a shard's [local | halo] table at BASELINE configs[4] x 8)."""
    if not fused_shape_ok(x, F, N) or not aggregators_fusable(aggregators):
        return False
    plan = plan_of(graph)
    return plan.G > 0 and plan.fused_tables() is not False


def fused_shape_ok(x, F, N):
    """The graph-independent half of fused_applies (a row block of shard.BlockPipeline brings its own plan)."""
    if not FUSED or fused_panels(F, N) is None:
        return False
    return x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= F and x.data_ptr() % 4 == 0 and x.shape[0] >= 1


def two_kernel_applies(N, n_scaler, aggregators):
    """The operator set / widths the two-kernel grouped path (gather in degree order + grouped contraction) is built for."""
    return (MIN_OUT <= N <= 128 and (n_scaler == 3 or (n_scaler == 2 and N >= TWO_SCALER_MIN_OUT)) and tuple(aggregators) == STANDARD_AGGREGATORS)


def applies(graph, V, N, n_scaler, aggregators, F=None, n_edges=None, x_rows=None):
    """Whether a PNASimpleLayer call over `graph` takes a degree-grouped path at all: the two-kernel one (two_kernel_applies) or -- round 6 --
    the one-kernel layer alone, which also serves one or two scalers, any distinct aggregators out of mean / sum / max / min / std and
    outputs wider than one launch (fused_panels); which of the two runs is functional.simple_layer_degree_grouped's decision."""
    from .graph import Graph
    from .shard import HaloGraph
    if not (ENABLED and type(graph) in (Graph, HaloGraph) and V >= MIN_ROWS and 1 <= n_scaler <= 3):
        return False
    two = two_kernel_applies(N, n_scaler, aggregators)
    one = FUSED and F is not None and aggregators_fusable(aggregators) and N >= MIN_OUT and fused_panels(F, N) is not None
    if not (two or one) or V * out_pitch_floats(N) * 4 >= (1 << 32):
        return False
    # the gather of this path REQUIRES the hand-scheduled kernel (only it writes the plan's row order): its own preconditions
    # (pna_segreduce.hip fast_ok: dwordx4 lanes, 32-bit edge positions; since round 4 source tables beyond 2^24 rows / 4 GiB through
    # its 64-bit-address instantiations) -- ADVICE r2
    if (F is not None and F < 4) or (n_edges is not None and not 0 < n_edges < (1 << 30)) or (x_rows is not None and x_rows >= (1 << 32)):
        return False
    plan = plan_of(graph)
    return plan.G > 0 and plan.NR <= MAX_REST_FRACTION * V
