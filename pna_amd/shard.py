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
  * per layer: pack the rows each peer asked for (one index_select), ONE all_to_all_single with
    per-peer split sizes (direct point-to-point transfers over xGMI -- every peer link is used
    concurrently; a ring would be bound by a single 153 GB/s link), then the ordinary fused
    segment-reduce over the extended feature table.
"""
from typing import List, Optional

import torch
import torch.distributed as dist

from .graph import Graph


def partition_bounds(num_nodes: int, world_size: int) -> List[int]:
    """Contiguous destination ranges: rank r owns [bounds[r], bounds[r+1])."""
    return [(num_nodes * r) // world_size for r in range(world_size + 1)]


class HaloGraph(Graph):
    """The local shard: a Graph whose destinations are this rank's nodes (renumbered from 0) and whose
    source ids index the extended table [local | halo].  `source_features(h_local)` performs the halo
    exchange and returns that table; the layers call it before gathering."""

    def __init__(self, src_ext, dst_local, n_local, n_halo, send_idx, send_splits, recv_splits, group, lo, hi,
                 global_num_nodes, batch_num_nodes=None):
        super().__init__(src_ext, dst_local, n_local, batch_num_nodes)
        self.n_halo = n_halo
        self.send_idx = send_idx                  # int64 [sum(send_splits)] local rows to pack, grouped by peer
        self.send_splits = send_splits            # rows sent to each peer
        self.recv_splits = recv_splits            # rows received from each peer
        self.group = group
        self.lo, self.hi = lo, hi
        self.global_num_nodes = global_num_nodes
        self._ext = None                          # resident [local | halo] table of alloc_features()

    def to(self, device):
        g = HaloGraph(self.src.to(device), self.dst.to(device), self.num_nodes, self.n_halo, self.send_idx.to(device),
                      self.send_splits, self.recv_splits, self.group, self.lo, self.hi, self.global_num_nodes,
                      self.batch_num_nodes)
        return g

    def alloc_features(self, F: int, pitch: Optional[int] = None, device=None) -> torch.Tensor:
        """A resident extended table [local | halo] of row pitch `pitch` floats; returns its LOCAL part (n_local, F), a
        view.  Features kept there are exchanged in place by `source_features`: peers' rows land directly behind the
        local ones (whole pitch-sized rows travel), so the step saves the concatenation of the packed halo with the
        local rows -- 2x(local + halo) bytes of HBM traffic, more than the gather kernel itself moves at 8 GPUs."""
        pitch = F if pitch is None else max(int(pitch), F)
        self._ext = torch.zeros(self.num_nodes + self.n_halo, pitch, dtype=torch.float32,
                                device=self.device if device is None else device)
        return self._ext[: self.num_nodes, :F]

    def _resident(self, h_local):
        e = self._ext
        return (e is not None and h_local.dim() == 2 and h_local.untyped_storage().data_ptr() == e.untyped_storage().data_ptr()
                and h_local.storage_offset() == e.storage_offset() and h_local.stride(0) == e.stride(0) and h_local.stride(1) == 1
                and h_local.shape[0] == self.num_nodes and not (torch.is_grad_enabled() and h_local.requires_grad))

    def source_features(self, h_local: torch.Tensor) -> torch.Tensor:
        """[h_local | halo rows] after one all-to-all; differentiable (the backward is the transposed
        all-to-all followed by a scatter-add into the owners' rows)."""
        if h_local.shape[0] != self.num_nodes:
            raise ValueError(f"expected {self.num_nodes} local rows, got {h_local.shape[0]}")
        if self.n_halo == 0 and sum(self.send_splits) == 0:
            return h_local
        if self._resident(h_local):               # inference on features living in the resident table: no pack-side
            e = self._ext                         # concatenation, the halo is received in place
            send = e[: self.num_nodes].index_select(0, self.send_idx)
            dist.all_to_all_single(e[self.num_nodes:], send, self.recv_splits, self.send_splits, group=self.group)
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


def shard_graph(src: torch.Tensor, dst: torch.Tensor, num_nodes: int, rank: Optional[int] = None,
                world_size: Optional[int] = None, group=None) -> HaloGraph:
    """Build this rank's shard from the GLOBAL edge list (every rank passes the same src/dst, e.g. a
    deterministic generator or a replicated file).  Collective: every rank of `group` must call it."""
    rank = dist.get_rank(group) if rank is None else rank
    world_size = dist.get_world_size(group) if world_size is None else world_size
    dev = src.device
    bounds = partition_bounds(num_nodes, world_size)
    lo, hi = bounds[rank], bounds[rank + 1]
    n_local = hi - lo
    mine = (dst >= lo) & (dst < hi)
    s, d = src[mine].long(), dst[mine].long() - lo
    bt = torch.tensor(bounds, device=dev, dtype=torch.long)
    owner = torch.searchsorted(bt, s, right=True) - 1
    src_ext = torch.empty_like(s)
    local = owner == rank
    src_ext[local] = s[local] - lo
    recv_lists, recv_splits, off = [], [], n_local
    for p in range(world_size):
        if p == rank:
            recv_lists.append(s.new_empty(0))
            recv_splits.append(0)
            continue
        m = owner == p
        uniq, inv = torch.unique(s[m], sorted=True, return_inverse=True)            # de-duplicated halo of peer p
        src_ext[m] = off + inv
        recv_lists.append(uniq - bounds[p])                                         # peer-local row ids
        recv_splits.append(int(uniq.numel()))
        off += int(uniq.numel())
    n_halo = off - n_local
    # tell every peer which of its rows we need: counts first, then the id lists
    cnt_out = torch.tensor(recv_splits, dtype=torch.long, device=dev)
    cnt_in = torch.empty_like(cnt_out)
    dist.all_to_all_single(cnt_in, cnt_out, group=group)
    send_splits = [int(v) for v in cnt_in.tolist()]
    want = torch.cat(recv_lists) if n_halo else s.new_empty(0)
    send_idx = s.new_empty(sum(send_splits))
    dist.all_to_all_single(send_idx, want, send_splits, recv_splits, group=group)
    return HaloGraph(src_ext, d, n_local, n_halo, send_idx, send_splits, recv_splits, group, lo, hi, num_nodes)
