"""Multi-GPU execution of the PNA layer: destination-range sharding + one halo all-to-all per layer.

The reference has no distributed code; this is the MI355X-native scaling path of
SURVEY.md 8(e).  One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" on
CPU for the tests):

  * nodes are split into `world_size` contiguous destination ranges; rank r owns the feature rows of
    its nodes and the CSR of ALL in-edges of its nodes, so every destination row is reduced entirely
    on one GPU and no reduction crosses GPUs (results are identical to the single-GPU kernel);
  * the only cross-GPU dependency is reading source rows owned by a peer.  Per peer the sorted,
    de-duplicated list of referenced rows (the halo) is computed once per graph; source ids are
    remapped to [local rows | halo rows of peer 0 | halo rows of peer 1 | ...];
  * per layer: pack the rows each peer asked for (pna_pack_rows_f32), ONE all_to_all_single with
    per-peer split sizes (direct point-to-point transfers over xGMI -- every peer link is used
    concurrently; a ring would be bound by a single 153 GB/s link), then the ordinary fused
    segment-reduce over the extended feature table -- the rows that only read local sources are
    aggregated WHILE the exchange is in flight (HaloGraph.split_work_lists, functional.aggregate).
"""
from typing import List, Optional

import torch
import torch.distributed as dist

from .graph import Graph


def partition_bounds(num_nodes: int, world_size: int, dst: Optional[torch.Tensor] = None, balance: str = "nodes") -> List[int]:
    """Contiguous destination ranges: rank r owns [bounds[r], bounds[r+1]).  balance="nodes": equal node counts;
    balance="edges": equal in-edge counts (each rank's gather work; needs the global `dst`) -- on a power-law graph the
    node-balanced ranges differ by the hubs they happen to contain."""
    if balance == "nodes" or dst is None:
        return [(num_nodes * r) // world_size for r in range(world_size + 1)]
    if balance != "edges":
        raise ValueError(f"unknown balance {balance!r} (nodes | edges)")
    deg = torch.bincount(dst.long(), minlength=num_nodes)
    cum = torch.cumsum(deg, 0)
    total = int(cum[-1].item()) if num_nodes else 0
    targets = torch.tensor([(total * r) // world_size for r in range(1, world_size)], device=cum.device, dtype=cum.dtype)
    cuts = torch.searchsorted(cum, targets, right=False) + 1 if world_size > 1 else targets
    b = [0] + [int(v) for v in cuts.tolist()] + [num_nodes]
    for i in range(1, len(b)):                       # monotone, inside [0, V]
        b[i] = min(max(b[i], b[i - 1]), num_nodes)
    return b


def bfs_order(src: torch.Tensor, dst: torch.Tensor, num_nodes: int, start: Optional[int] = None) -> torch.Tensor:
    """A locality-aware renumbering: perm[new_id] = old_id in breadth-first order from `start` (default: the node of largest
    in-degree), unreached nodes appended in id order.  Neighbours get close ids, so contiguous destination ranges cut fewer
    edges on graphs that HAVE locality (meshes, molecules batched in arbitrary order, citation graphs); on the locality-free
    Chung-Lu benchmark graph it cannot help (tools/halo_dryrun.py reports both).  Level-synchronous, torch ops only."""
    dev = src.device
    order = torch.sort(src.long(), stable=True)
    s_sorted, nbr = order.values, dst.long()[order.indices]
    rp = torch.zeros(num_nodes + 1, dtype=torch.long, device=dev)
    rp[1:] = torch.cumsum(torch.bincount(s_sorted, minlength=num_nodes), 0)
    if start is None:
        start = int(torch.argmax(torch.bincount(dst.long(), minlength=num_nodes)).item())
    seen = torch.zeros(num_nodes, dtype=torch.bool, device=dev)
    seen[start] = True
    frontier = torch.tensor([start], device=dev)
    out = [frontier]
    while frontier.numel():
        cnt = rp[frontier + 1] - rp[frontier]
        if int(cnt.sum().item()) == 0:
            break
        base = torch.repeat_interleave(rp[frontier], cnt)
        offs = torch.arange(int(cnt.sum().item()), device=dev) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
        cand = nbr[base + offs]
        cand = cand[~seen[cand]]
        if cand.numel() == 0:
            break
        # first occurrence order keeps the discovery order deterministic
        uniq, inv = torch.unique(cand, return_inverse=True)
        first = torch.full((uniq.numel(),), cand.numel(), dtype=torch.long, device=dev).scatter_reduce_(
            0, inv, torch.arange(cand.numel(), device=dev), reduce="amin")
        frontier = uniq[torch.argsort(first)]
        seen[frontier] = True
        out.append(frontier)
    rest = torch.nonzero(~seen).flatten()
    return torch.cat(out + [rest])


class HaloGraph(Graph):
    """The local shard: a Graph whose destinations are this rank's nodes (renumbered from 0) and whose
    source ids index the extended table [local | halo].  `source_features(h_local)` performs the halo
    exchange and returns that table; the layers call it before gathering.

    Overlap: destination rows whose in-edges ALL come from local sources ("interior" rows) do not depend on the
    exchange.  With `source_features(h, defer=True)` the all-to-all is only STARTED (async); `functional.aggregate` then
    launches the interior rows' work list, waits for the exchange and launches the rest ("boundary" rows and every hub row).
    Every row is still reduced by one lane group in its original edge order: results are bit-identical to the unsharded
    kernel whichever way the launch is split."""

    def __init__(self, src_ext, dst_local, n_local, n_halo, send_idx, send_splits, recv_splits, group, lo, hi,
                 global_num_nodes, batch_num_nodes=None, any_exchange=True, bounds=None, recv_ids=None, rank=None):
        super().__init__(src_ext, dst_local, n_local, batch_num_nodes)
        self.rank = rank                          # this shard's position in `bounds` (= rank in `group`)
        self.bounds = bounds                      # node ranges of all ranks (BlockPipeline needs the peers' row counts)
        self.recv_ids = recv_ids                  # int64 [n_halo]: peer-local id of every halo row, grouped by peer, ascending
        self.n_halo = n_halo
        self.send_idx = send_idx                  # int64 [sum(send_splits)] local rows to pack, grouped by peer
        self.send_splits = send_splits            # rows sent to each peer
        self.recv_splits = recv_splits            # rows received from each peer
        self.group = group
        self.lo, self.hi = lo, hi
        self.global_num_nodes = global_num_nodes
        self.any_exchange = any_exchange          # False only when NO rank of the group has a halo (decided collectively)
        self._ext = None                          # resident [local | halo] table of alloc_features()
        self._send_idx32 = None
        self._pending = None                      # the in-flight exchange started by source_features(defer=True)
        self._split = None                        # (interior mask, interior items, boundary items)
        self._interior_fraction = None

    def to(self, device):
        g = HaloGraph(self.src.to(device), self.dst.to(device), self.num_nodes, self.n_halo, self.send_idx.to(device),
                      self.send_splits, self.recv_splits, self.group, self.lo, self.hi, self.global_num_nodes,
                      self.batch_num_nodes, self.any_exchange, self.bounds,
                      None if self.recv_ids is None else self.recv_ids.to(device), self.rank)
        return g

    # -- row classes -----------------------------------------------------------------------------------
    def interior_mask(self) -> torch.Tensor:
        """bool [n_local]: rows that are not hubs and whose in-edges all have LOCAL sources (ids < n_local)."""
        return self.split_work_lists()[0]

    def interior_fraction(self) -> float:
        """Fraction of the local rows the exchange overlap covers (cached with the row classes)."""
        if self._interior_fraction is None:
            self._interior_fraction = float(self.interior_mask().float().mean().item()) if self.num_nodes else 0.0
        return self._interior_fraction

    def split_work_lists(self):
        """(interior mask, work list of the interior rows, work list of everything else incl. all hub segments)."""
        if self._split is None:
            c = self.csr
            remote = (c.col.long() >= self.num_nodes).to(torch.int32)
            per_row = torch.zeros(self.num_nodes, dtype=torch.int32, device=remote.device)
            if remote.numel():
                per_row.index_add_(0, c.row.long(), remote)
            deg = (c.rowptr[1:] - c.rowptr[:-1])
            hs = self.heavy_schedule()
            interior = (per_row == 0) & (deg <= hs.threshold if hs.threshold > 0 else torch.ones_like(deg, dtype=torch.bool))
            self._split = (interior, self.work_items_subset(interior, include_heavy=False),
                           self.work_items_subset(~interior, include_heavy=True))
        return self._split

    # -- feature tables ---------------------------------------------------------------------------------
    def alloc_features(self, F: int, pitch: Optional[int] = None, device=None) -> torch.Tensor:
        """A resident extended table [local | halo] of row pitch `pitch` floats (default F: dense rows, so that exactly F
        floats per halo row cross xGMI); returns its LOCAL part (n_local, F), a view.  Features kept there are exchanged in
        place by `source_features`: peers' rows land directly behind the local ones, so the step has no concatenation pass
        (which would move 2 x (local + halo) bytes, more than the gather kernel itself at 8 GPUs)."""
        pitch = F if pitch is None else max(int(pitch), F)
        self._ext = torch.zeros(self.num_nodes + self.n_halo, pitch, dtype=torch.float32,
                                device=self.device if device is None else device)
        return self._ext[: self.num_nodes, :F]

    def _resident(self, h_local):
        e = self._ext
        return (e is not None and h_local.dim() == 2 and h_local.untyped_storage().data_ptr() == e.untyped_storage().data_ptr()
                and h_local.storage_offset() == e.storage_offset() and h_local.stride(0) == e.stride(0) and h_local.stride(1) == 1
                and h_local.shape[0] == self.num_nodes and not (torch.is_grad_enabled() and h_local.requires_grad))

    def _pack(self, e):
        """Rows the peers asked for, grouped by peer: pna_pack_rows_f32 on the GPU (whole pitch-wide rows of the resident
        table; pitch = F unless the caller chose otherwise), index_select on the CPU (gloo tests)."""
        if e.is_cuda:
            from . import ops
            if self._send_idx32 is None or self._send_idx32.device != e.device:
                self._send_idx32 = self.send_idx.to(device=e.device, dtype=torch.int32)
            return ops.pack_rows(e[: self.num_nodes], self._send_idx32)
        return e[: self.num_nodes].index_select(0, self.send_idx)

    def finish_exchange(self):
        """Wait for the exchange started by source_features(defer=True) (no-op otherwise)."""
        if self._pending is not None:
            work, keep = self._pending
            work.wait()
            self._pending = None

    def source_features(self, h_local: torch.Tensor, defer: bool = False) -> torch.Tensor:
        """[h_local | halo rows] after one all-to-all; differentiable (the backward is the transposed
        all-to-all followed by a scatter-add into the owners' rows).  defer=True (inference on the resident table only):
        return as soon as the exchange is STARTED; the caller must hand the result to functional.aggregate, which overlaps
        the interior rows with it, or call finish_exchange() before touching the halo part."""
        if h_local.shape[0] != self.num_nodes:
            raise ValueError(f"expected {self.num_nodes} local rows, got {h_local.shape[0]}")
        self.finish_exchange()
        if not self.any_exchange:                 # no rank has a halo: nobody enters the collective
            return h_local
        if self._resident(h_local):               # inference on features living in the resident table: no pack-side
            e = self._ext                         # concatenation, the halo is received in place
            send = self._pack(e)
            recv = e[self.num_nodes:]
            if defer:
                work = dist.all_to_all_single(recv, send, self.recv_splits, self.send_splits, group=self.group, async_op=True)
                self._pending = (work, send)      # `send` must outlive the transfer
            else:
                dist.all_to_all_single(recv, send, self.recv_splits, self.send_splits, group=self.group)
            return e[:, : h_local.shape[1]]
        return _HaloExchange.apply(h_local, self)


class _HaloExchange(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h_local, g: HaloGraph):
        ctx.g = g
        send = h_local.index_select(0, g.send_idx)                                  # pack
        recv = h_local.new_empty((g.n_halo,) + tuple(h_local.shape[1:]))
        dist.all_to_all_single(recv, send, g.recv_splits, g.send_splits, group=g.group)
        return torch.cat([h_local, recv], dim=0)

    @staticmethod
    def backward(ctx, grad_ext):
        g = ctx.g
        n = g.num_nodes
        grad_local = grad_ext[:n].clone()
        grad_halo = grad_ext[n:].contiguous()
        back = grad_halo.new_empty((sum(g.send_splits),) + tuple(grad_halo.shape[1:]))
        dist.all_to_all_single(back, grad_halo, g.send_splits, g.recv_splits, group=g.group)
        grad_local.index_add_(0, g.send_idx, back)
        return grad_local, None


def shard_local(src: torch.Tensor, dst: torch.Tensor, bounds: List[int], rank: int):
    """The part of shard_graph that needs no communication: this rank's in-edges with their sources renumbered into
    [local rows | halo rows], and per peer the (peer-local) ids of the rows wanted from it.  ONE sort of the remote sources: the
    owners are contiguous id ranges, so the sorted unique ids are already grouped by owner, peer by peer (round 2 ran one
    torch.unique per peer over a masked copy: 7 sorts and 14 masked copies at 8 ranks).  At BASELINE configs[4] size (V = 16 M,
    E = 160 M, 8 ranks) on one MI355X: tools/shard_time.py.
    Returns (src_ext int64 [E_local], dst_local int64 [E_local], n_local, [ids wanted from peer p], [their counts])."""
    world_size = len(bounds) - 1
    lo, hi = bounds[rank], bounds[rank + 1]
    n_local = hi - lo
    mine = (dst >= lo) & (dst < hi)
    s, d = src[mine].long(), dst[mine].long() - lo
    del mine
    local = (s >= lo) & (s < hi)
    src_ext = torch.where(local, s - lo, s)                                      # (remote entries overwritten below)
    remote = ~local
    uniq, inv = torch.unique(s[remote], sorted=True, return_inverse=True)        # de-duplicated halo, ascending global id
    src_ext[remote] = n_local + inv
    bt = torch.tensor(bounds, device=s.device, dtype=torch.long)
    cuts = torch.searchsorted(uniq, bt).tolist()                                 # peer p's ids: uniq[cuts[p]:cuts[p+1]]
    recv_lists = [uniq[cuts[p]:cuts[p + 1]] - bounds[p] for p in range(world_size)]
    recv_splits = [cuts[p + 1] - cuts[p] for p in range(world_size)]
    assert recv_splits[rank] == 0
    return src_ext, d, n_local, recv_lists, recv_splits


def shard_graph(src: torch.Tensor, dst: torch.Tensor, num_nodes: int, rank: Optional[int] = None,
                world_size: Optional[int] = None, group=None, balance: str = "nodes") -> HaloGraph:
    """Build this rank's shard from the GLOBAL edge list (every rank passes the same src/dst, e.g. a
    deterministic generator or a replicated file).  Collective: every rank of `group` must call it.
    balance: "nodes" (equal node counts) or "edges" (equal in-edge counts), see partition_bounds.  For a locality-aware
    numbering renumber the graph with bfs_order() first."""
    rank = dist.get_rank(group) if rank is None else rank
    world_size = dist.get_world_size(group) if world_size is None else world_size
    dev = src.device
    bounds = partition_bounds(num_nodes, world_size, dst, balance)
    lo, hi = bounds[rank], bounds[rank + 1]
    src_ext, d, n_local, recv_lists, recv_splits = shard_local(src, dst, bounds, rank)
    s = src_ext
    off = n_local + sum(recv_splits)
    n_halo = off - n_local
    # tell every peer which of its rows we need: counts first, then the id lists
    cnt_out = torch.tensor(recv_splits, dtype=torch.long, device=dev)
    cnt_in = torch.empty_like(cnt_out)
    dist.all_to_all_single(cnt_in, cnt_out, group=group)
    send_splits = [int(v) for v in cnt_in.tolist()]
    want = torch.cat(recv_lists) if n_halo else s.new_empty(0)
    send_idx = s.new_empty(sum(send_splits))
    dist.all_to_all_single(send_idx, want, send_splits, recv_splits, group=group)
    # Does ANY rank exchange anything?  Decided once, collectively: every rank then either always enters the per-layer
    # collective (possibly with empty splits) or never does -- a rank skipping it on its own would hang gloo and
    # desynchronise NCCL's collective sequence.
    tot = torch.tensor([n_halo + sum(send_splits)], dtype=torch.long, device=dev)
    dist.all_reduce(tot, group=group)
    return HaloGraph(src_ext, d, n_local, n_halo, send_idx, send_splits, recv_splits, group, lo, hi, num_nodes,
                     any_exchange=bool(int(tot.item()) > 0), bounds=list(bounds), recv_ids=want, rank=rank)


class BlockPipeline:
    """Multi-layer inference on a sharded graph with the inter-layer halo exchange cut into ROW BLOCKS.

    Per-layer exchange (HaloGraph.source_features) serialises  layer L -> pack -> all-to-all -> layer L+1 ; only the rows without
    remote sources overlap with it, and on a locality-free graph those are 1 % of the rows.  Here every rank cuts its rows into
    `n_blocks` contiguous blocks; as soon as layer L has produced block b, the rows of b that peers need are packed and sent,
    and the peers' block b is received IN PLACE into the next layer's [local | halo] table -- while blocks b+1.. of layer L are
    still being computed.  Layer L+1 starts when all blocks have arrived.  (The send lists are ascending row ids per peer, the
    blocks contiguous id ranges: the rows of one (peer, block) pair are a contiguous piece of the send buffer and of the halo.)

    layer_rows(l, table, r0, r1, out, b): compute rows [r0, r1) of layer l from the complete table `table`
    ((n_local + n_halo, pitch): [local | halo]) into `out` (a view of the next table's rows [r0, r1); the WHOLE next table when
    the callable has `wants_full_out = True`: functional.SimpleLayerRows scatters a block's rows by node id).  Point-to-point
    isend / irecv (RCCL on GPUs, gloo on CPU); every rank must use the same n_blocks."""

    def __init__(self, graph: HaloGraph, n_blocks: int):
        if graph.bounds is None or graph.recv_ids is None or graph.rank is None:
            raise ValueError("BlockPipeline needs a HaloGraph built by shard_graph (bounds / recv_ids / rank)")
        self.g, self.B = graph, int(n_blocks)
        g, B = graph, self.B
        W = len(g.bounds) - 1
        self.rank = g.rank
        n = g.num_nodes
        self.rows = [(n * b) // B for b in range(B + 1)]                          # my block bounds (local row ids)
        dev = g.send_idx.device
        # send side: position of every (block, peer) piece in the peer-major send list, then a block-major copy of the list
        so = [0]
        for v in g.send_splits:
            so.append(so[-1] + v)
        cut_s = []                                                                # cut_s[p][b]: first entry of block b in peer p's segment
        mine = torch.tensor(self.rows, device=dev, dtype=g.send_idx.dtype)
        for p in range(W):
            seg = g.send_idx[so[p]:so[p + 1]]
            cut_s.append((torch.searchsorted(seg, mine) + so[p]).tolist())
        order, self.send_piece = [], []                                           # send_piece[b][p] = (offset, count) in the block-major list
        off = 0
        for b in range(B):
            row = []
            for p in range(W):
                a, z = cut_s[p][b], cut_s[p][b + 1]
                order.append(torch.arange(a, z, device=dev))
                row.append((off, z - a))
                off += z - a
            self.send_piece.append(row)
        perm = torch.cat(order) if order else g.send_idx.new_empty(0)
        self.send_rows = g.send_idx[perm].contiguous()                            # block-major, peer-minor
        self._send_rows32 = None
        # receive side: the halo rows of peer p (ascending peer-local ids) cut by PEER p's block bounds
        ro = [0]
        for v in g.recv_splits:
            ro.append(ro[-1] + v)
        self.recv_piece = []                                                      # recv_piece[b][p] = (first halo row, count)
        cut_r = []
        for p in range(W):
            n_p = g.bounds[p + 1] - g.bounds[p]
            theirs = torch.tensor([(n_p * b) // B for b in range(B + 1)], device=g.recv_ids.device, dtype=g.recv_ids.dtype)
            cut_r.append((torch.searchsorted(g.recv_ids[ro[p]:ro[p + 1]], theirs) + ro[p]).tolist())
        for b in range(B):
            self.recv_piece.append([(cut_r[p][b], cut_r[p][b + 1] - cut_r[p][b]) for p in range(W)])
        self.W = W

    def _pack(self, table, b):
        g = self.g
        o0, o1 = self.send_piece[b][0][0], self.send_piece[b][-1][0] + self.send_piece[b][-1][1]
        if o1 == o0:
            return table.new_empty(0, table.shape[1])
        if table.is_cuda:
            from . import ops
            if self._send_rows32 is None or self._send_rows32.device != table.device:
                self._send_rows32 = self.send_rows.to(device=table.device, dtype=torch.int32)
            return ops.pack_rows(table[: g.num_nodes], self._send_rows32[o0:o1])
        return table[: g.num_nodes].index_select(0, self.send_rows[o0:o1])

    def start_block(self, table, b):
        """Pack and post the exchange of block b of `table`'s local rows; its halo part receives the peers' block b in place.
        Returns the handles to wait for (and the send buffer, which must outlive them)."""
        g = self.g
        if not g.any_exchange:
            return [], None
        send = self._pack(table, b)
        if send.is_cuda and dist.get_backend(g.group) != "nccl":
            # (the one-GPU smoke tests run this over gloo, whose point-to-point calls read device memory from the host without
            # ordering against the stream that is still packing it; RCCL orders against the current stream itself)
            torch.cuda.current_stream(send.device).synchronize()
        base = self.send_piece[b][0][0]
        ops_ = []
        for p in range(self.W):
            if p == self.rank:
                continue
            so, sc = self.send_piece[b][p]
            ro, rc = self.recv_piece[b][p]
            peer = p if g.group is None else dist.get_global_rank(g.group, p)
            if sc:
                ops_.append(dist.P2POp(dist.isend, send[so - base:so - base + sc], peer, group=g.group))
            if rc:
                ops_.append(dist.P2POp(dist.irecv, table[g.num_nodes + ro:g.num_nodes + ro + rc], peer, group=g.group))
        return (dist.batch_isend_irecv(ops_) if ops_ else []), send

    def exchange_all(self, table):
        """The whole halo of `table` (all blocks back to back): the exchange in front of the first layer."""
        pend = [self.start_block(table, b) for b in range(self.B)]
        for works, _ in pend:
            for w in works:
                w.wait()

    def run(self, layer_rows, n_layers, table_a, table_b):
        """n_layers layers from table_a (local rows filled; halo NOT yet exchanged) ping-ponging with table_b; returns the table
        that holds the last layer's local rows (its halo part is stale)."""
        g = self.g
        self.exchange_all(table_a)
        cur, nxt = table_a, table_b
        for l in range(n_layers):
            pend = []
            for b in range(self.B):
                r0, r1 = self.rows[b], self.rows[b + 1]
                layer_rows(l, cur, r0, r1, nxt if getattr(layer_rows, "wants_full_out", False) else nxt[r0:r1], b)
                if l + 1 < n_layers:
                    pend.append(self.start_block(nxt, b))
            for works, _ in pend:
                for w in works:
                    w.wait()
            cur, nxt = nxt, cur
        return cur
