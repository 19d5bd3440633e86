"""Graph container consumed by the PNA layers.

The reference's sparse layers take a DGL 0.4.2 `DGLGraph` / `BatchedDGLGraph` and only use
`ndata`, `edata`, `apply_edges`, `update_all` on it (models/dgl/pna_layer.py:56-65,:199-203); the
message passing itself is what libpna_amd.so replaces, so the container here just holds the edge
list and caches the destination-sorted CSR (int32 rowptr/col) the kernels walk, the heavy-row
schedule and the per-row degree scalers.  `Graph.from_dgl` adapts a real DGLGraph when DGL is
installed.  All index preparation is torch tensor ops on whatever device the edge list lives on
(testable on CPU); only `degree_scalers` and the layers themselves need the GPU library.
"""
import ctypes
from typing import NamedTuple, Optional

import torch

from . import _lib

# Rows with more in-edges than this are cut into SEG_LEN-edge segments reduced by separate lane
# groups (see include/pna_amd.h "Heavy rows"); 128/128 measured best on the 10 M-edge power-law graph
# (tools/sweep.py [removed in round 5: git history], profiles/).
HEAVY_THRESHOLD = 128
SEG_LEN = 128


class CSR(NamedTuple):
    rowptr: torch.Tensor   # int32 [V+1]
    col: torch.Tensor      # int32 [E]  source node of each CSR edge
    eid: torch.Tensor      # int64 [E]  original edge id of each CSR edge (stable in dst)
    row: torch.Tensor      # int32 [E]  destination node of each CSR edge
    max_degree: int


class HeavySchedule(NamedTuple):
    threshold: int
    seg_len: int
    n_heavy: int
    n_seg: int
    heavy_rows: Optional[torch.Tensor]    # int32 [n_heavy]
    heavy_segptr: Optional[torch.Tensor]  # int32 [n_heavy+1]
    seg_heavy: Optional[torch.Tensor]     # int32 [n_seg]


def build_csr(src, dst, num_nodes, edge_graph=None, node_offset=None):
    """In-edges grouped by destination; inside a group the original edge order is kept (this is the
    mailbox order DGL's degree-bucketed update_all presents to reduce_func).  On the GPU this is
    pna_collate_csr_i32 (one radix sort over the bits num_nodes needs), which can also apply
    the member-graph offsets of a batch (edge_graph[k] = graph of edge k, node_offset[g] = first node of
    graph g); on the CPU (tests, host-side preparation) the same result from torch sort / bincount."""
    if src.numel() >= 2 ** 31 or num_nodes >= 2 ** 31:
        raise ValueError("graph too large for int32 indices")
    if dst.is_cuda:
        return _build_csr_device(src, dst, num_nodes, edge_graph, node_offset)
    dst = dst.long()
    src = src.long()
    if edge_graph is not None:
        off = node_offset.long()[edge_graph.long()]
        src, dst = src + off, dst + off
    order = torch.sort(dst, stable=True).indices
    deg = torch.bincount(dst, minlength=num_nodes)
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.int32, device=dst.device)
    rowptr[1:] = torch.cumsum(deg, 0).to(torch.int32)
    col = src[order].to(torch.int32)
    row = dst[order].to(torch.int32)
    max_degree = int(deg.max().item()) if num_nodes > 0 and src.numel() > 0 else 0
    return CSR(rowptr, col, order, row, max_degree)


def _build_csr_device(src, dst, num_nodes, edge_graph, node_offset):
    dev = dst.device
    E = int(src.numel())
    i32 = lambda t: None if t is None else t.to(device=dev, dtype=torch.int32).contiguous()   # noqa: E731
    src32, dst32, eg32, no32 = i32(src), i32(dst), i32(edge_graph), i32(node_offset)
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    col = torch.empty(E, dtype=torch.int32, device=dev)
    eid = torch.empty(E, dtype=torch.int32, device=dev)
    row = torch.empty(E, dtype=torch.int32, device=dev)
    L = _lib.lib()
    nbytes = L.pna_collate_workspace_bytes(E, num_nodes)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    ptr = lambda t: None if t is None or t.numel() == 0 else ctypes.c_void_p(t.data_ptr())   # noqa: E731
    rc = L.pna_collate_csr_i32(ptr(src32), ptr(dst32), E, num_nodes, ptr(eg32), ptr(no32), ptr(rowptr), ptr(col), ptr(eid),
                               ptr(row), ptr(ws), nbytes, _lib.stream_ptr(dev))
    _lib.check(rc, "pna_collate_csr_i32")
    ws.record_stream(torch.cuda.current_stream(dev))
    max_degree = int((rowptr[1:] - rowptr[:-1]).max().item()) if num_nodes > 0 and E > 0 else 0
    return CSR(rowptr, col, eid.long(), row, max_degree)


def build_heavy_schedule(rowptr, max_degree, threshold=None, seg_len=None):
    threshold = HEAVY_THRESHOLD if threshold is None else threshold
    seg_len = SEG_LEN if seg_len is None else seg_len
    if threshold <= 0 or max_degree <= threshold:
        return HeavySchedule(threshold, seg_len, 0, 0, None, None, None)
    deg = (rowptr[1:] - rowptr[:-1]).long()
    heavy = torch.nonzero(deg > threshold).flatten()
    nseg = (deg[heavy] + seg_len - 1) // seg_len
    segptr = torch.zeros(heavy.numel() + 1, dtype=torch.int64, device=rowptr.device)
    segptr[1:] = torch.cumsum(nseg, 0)
    seg_heavy = torch.repeat_interleave(torch.arange(heavy.numel(), device=rowptr.device), nseg)
    return HeavySchedule(threshold, seg_len, int(heavy.numel()), int(segptr[-1].item()),
                         heavy.to(torch.int32), segptr.to(torch.int32), seg_heavy.to(torch.int32))


class Graph:
    """Directed multigraph: edge k goes src[k] -> dst[k]; messages flow along edges and are reduced at dst."""

    def __init__(self, src, dst, num_nodes, batch_num_nodes=None):
        src = torch.as_tensor(src)
        dst = torch.as_tensor(dst, device=src.device)
        if src.shape != dst.shape or src.dim() != 1:
            raise ValueError("src and dst must be 1-D tensors of equal length")
        self.src = src.long()
        self.dst = dst.long()
        self.num_nodes = int(num_nodes)
        self.batch_num_nodes = list(batch_num_nodes) if batch_num_nodes is not None else [self.num_nodes]
        self.ndata = {}
        self.edata = {}
        self._csr = None
        self._heavy = {}
        self._scalers = {}
        self._workspace = None
        self._snorm_n = None

    # -- DGLGraph duck-typing ------------------------------------------------------------------
    def number_of_nodes(self):
        return self.num_nodes

    def number_of_edges(self):
        return int(self.src.numel())

    def edges(self):
        return self.src, self.dst

    def in_degrees(self):
        c = self.csr
        return (c.rowptr[1:] - c.rowptr[:-1]).long()

    @property
    def device(self):
        return self.src.device

    def to(self, device):
        g = Graph(self.src.to(device), self.dst.to(device), self.num_nodes, self.batch_num_nodes)
        g.ndata = {k: v.to(device) for k, v in self.ndata.items()}
        g.edata = {k: v.to(device) for k, v in self.edata.items()}
        return g

    @staticmethod
    def from_dgl(g):
        """Adapt a DGL graph (anything with .edges() and .number_of_nodes())."""
        src, dst = g.edges()
        bnn = None
        if hasattr(g, "batch_num_nodes"):
            b = g.batch_num_nodes
            bnn = b() if callable(b) else b
            bnn = [int(v) for v in bnn]
        return Graph(src, dst, g.number_of_nodes(), bnn)

    @staticmethod
    def batch(graphs):
        """Disjoint union with node-id offsets, like dgl.batch (realworld_benchmark/data/molecules.py:163)."""
        offs, srcs, dsts, sizes = 0, [], [], []
        for g in graphs:
            srcs.append(g.src + offs)
            dsts.append(g.dst + offs)
            sizes += g.batch_num_nodes
            offs += g.num_nodes
        return Graph(torch.cat(srcs), torch.cat(dsts), offs, sizes)

    @staticmethod
    def collate(srcs, dsts, num_nodes_list, device=None):
        """dgl.batch of edge lists with LOCAL node ids, on the device: one concatenation, then the member-graph
        offsets are applied inside the CSR build (pna_collate_csr_i32) -- data/molecules.py:153-164 without the
        per-graph host work.  Returns the batched Graph (global ids in .src/.dst, CSR already built)."""
        # one conversion and ONE concatenation per side (on the host when the members live there: a single copy to the device),
        # then the flat path.  What remains is the caller's data layout: concatenating 2 x 2048 small host tensors costs ~3.5 ms of
        # the 4.1 ms this takes for a 2048-molecule batch; a data set stored as one edge array + counts uses collate_flat (0.5 ms)
        ss = [torch.as_tensor(s) for s in srcs]
        dd = [torch.as_tensor(d) for d in dsts]
        counts = torch.tensor([t.numel() for t in ss], dtype=torch.long)
        src = torch.cat(ss) if ss else torch.zeros(0, dtype=torch.long)
        dst = torch.cat(dd) if dd else torch.zeros(0, dtype=torch.long)
        return Graph.collate_flat(src, dst, counts, [int(n) for n in num_nodes_list], device=device if device is not None else src.device)

    @staticmethod
    def collate_flat(src_local, dst_local, edge_counts, num_nodes, device=None):
        """The same batch from FLAT arrays -- what a data set stored as one edge array + per-graph counts hands over after one
        gather of the batch's members: `src_local` / `dst_local` = the members' edge lists with LOCAL node ids, one after the
        other; `edge_counts[i]` / `num_nodes[i]` = edges / nodes of member i (tensors or sequences).  No per-graph Python work
        (collate() concatenates 2 x B small tensors and sizes them one by one: 4.3 ms for 2048 molecules, this path ~0.5 ms):
        offsets, the CSR and the global ids are built by tensor ops and pna_collate_csr_i32 on the device."""
        src = torch.as_tensor(src_local)
        dev = torch.device(device) if device is not None else src.device
        src = src.to(dev)
        dst = torch.as_tensor(dst_local).to(dev)
        sizes_host = [int(n) for n in (num_nodes.tolist() if torch.is_tensor(num_nodes) else num_nodes)]
        sizes = torch.as_tensor(sizes_host, dtype=torch.long, device=dev)
        counts = torch.as_tensor(edge_counts, dtype=torch.long).to(dev)
        if counts.numel() != sizes.numel() or int(counts.sum()) != src.numel() or dst.numel() != src.numel():
            raise ValueError("collate_flat: edge_counts / num_nodes do not describe the edge arrays")
        node_offset = torch.cumsum(sizes, 0) - sizes
        edge_graph = torch.repeat_interleave(torch.arange(sizes.numel(), device=dev), counts)
        V = sum(sizes_host)
        csr = build_csr(src, dst, V, edge_graph, node_offset)
        off = node_offset[edge_graph]
        g = Graph(src.long() + off, dst.long() + off, V, sizes_host)
        g._csr = csr
        return g

    def source_features(self, h, defer=False):
        """Feature table the gather kernel indexes with the CSR's source ids.  Identity for a whole graph;
        pna_amd.shard.HaloGraph overrides it with the halo all-to-all ([local rows | halo rows]); `defer` lets that
        exchange overlap the aggregation of the rows that do not need it."""
        return h

    def finish_exchange(self):
        """Nothing to wait for on a whole graph (see HaloGraph)."""

    # -- cached index structures ---------------------------------------------------------------
    @property
    def csr(self) -> CSR:
        if self._csr is None:
            self._csr = build_csr(self.src, self.dst, self.num_nodes)
        return self._csr

    def heavy_schedule(self, threshold=None, seg_len=None) -> HeavySchedule:
        key = (HEAVY_THRESHOLD if threshold is None else threshold, SEG_LEN if seg_len is None else seg_len)
        if key not in self._heavy:
            c = self.csr
            self._heavy[key] = build_heavy_schedule(c.rowptr, c.max_degree, *key)
        return self._heavy[key]

    def work_items_subset(self, rows_mask, include_heavy, threshold=None, seg_len=None):
        """The work list of work_items() restricted to the light rows selected by `rows_mask` (bool [V]), with or
        without the heavy rows' segments: how the sharded layers split a launch into the rows that only read local sources
        (aggregated while the halo exchange is in flight) and the rest.  Natural order.  Not cached."""
        hs = self.heavy_schedule(threshold, seg_len)
        full = self.work_items(threshold, seg_len)
        n_seg = hs.n_seg if hs.n_heavy > 0 else 0
        light = full[n_seg:]
        keep = rows_mask.to(light.device)[light[:, 0].long()]
        parts = ([full[:n_seg]] if include_heavy and n_seg else []) + [light[keep]]
        return torch.cat(parts, dim=0).contiguous() if len(parts) > 1 else parts[0].contiguous()

    def work_items(self, threshold=None, seg_len=None, order="natural", window=96):
        """int32 (n, 4) work list {row, beg, end, slot} for the hand-scheduled kernel (include/pna_amd.h): first
        the segments of the heavy rows (slot = segment number), then one whole-row record (slot = -1) per other
        row.  order="degree": rows sorted by in-degree, descending and stable, so the lane groups of a wavefront
        walk rows of equal length; order="window": ascending row id, but sorted by in-degree inside consecutive
        windows of `window` rows; order="natural" (default): ascending row id, which keeps whatever locality the
        node numbering has (neighbouring rows tend to share source rows and then hit in L1/L2).  On the 10 M-edge
        power-law benchmark graph natural order is fastest although degree sorting removes a third of the
        instructions -- the gather is bound by cache misses, not by issue (tools/sweep.py [removed in round 5: git history], profiles/)."""
        hs = self.heavy_schedule(threshold, seg_len)
        key = ("items", hs.threshold, hs.seg_len, order, window)
        if key not in self._heavy:
            c = self.csr
            rp = c.rowptr.long()
            deg = rp[1:] - rp[:-1]
            dev = self.device
            if hs.n_heavy > 0:
                rows = torch.nonzero(deg <= hs.threshold).flatten()
                hrow = hs.heavy_rows.long()[hs.seg_heavy.long()]
                slot = torch.arange(hs.n_seg, device=dev)
                sidx = slot - hs.heavy_segptr.long()[hs.seg_heavy.long()]
                hbeg = rp[hrow] + sidx * hs.seg_len
                hend = torch.minimum(hbeg + hs.seg_len, rp[hrow + 1])
                heavy_items = torch.stack([hrow, hbeg, hend, slot], dim=1)
            else:
                rows = torch.arange(self.num_nodes, device=dev)
                heavy_items = torch.zeros(0, 4, dtype=torch.long, device=dev)
            if order == "degree":
                rows = rows[torch.sort(deg[rows], descending=True, stable=True).indices]
            elif order == "window":
                # sort key = (window index ascending, degree descending): one stable sort of a combined key
                w = torch.arange(rows.numel(), device=dev) // window
                key2 = w * (int(deg.max().item()) + 1 if rows.numel() else 1) + (int(deg.max().item()) - deg[rows] if rows.numel() else 0)
                rows = rows[torch.sort(key2, stable=True).indices]
            elif order != "natural":
                raise ValueError(order)
            light = torch.stack([rows, rp[rows], rp[rows + 1], torch.full_like(rows, -1)], dim=1)
            self._heavy[key] = torch.cat([heavy_items, light], dim=0).to(torch.int32).contiguous()
        return self._heavy[key]

    def workspace(self, nbytes):
        """Reusable float32 scratch (heavy-row partials) on the graph's device."""
        n = (nbytes + 3) // 4
        if self._workspace is None or self._workspace.numel() < n:
            self._workspace = torch.empty(max(n, 1), dtype=torch.float32, device=self.device)
        return self._workspace

    def degree_scalers(self, avg_log):
        """(amplification[V], attenuation[V]) per-row multipliers of models/dgl/scalers.py:12-19 for
        this graph's in-degrees, computed on the GPU by pna_degree_scalers_f32 with the reference's
        fp32 rounding sequence.  Cached per avg_log value."""
        key = float(avg_log)
        if key not in self._scalers:
            c = self.csr
            amp = torch.empty(self.num_nodes, dtype=torch.float32, device=self.device)
            att = torch.empty_like(amp)
            L = _lib.lib()
            rc = L.pna_degree_scalers_f32(_lib.dev_ptr(c.rowptr, torch.int32, "rowptr"), self.num_nodes,
                                          ctypes.c_float(key), _lib.dev_ptr(amp, torch.float32, "amp"),
                                          _lib.dev_ptr(att, torch.float32, "att"), _lib.stream_ptr(self.device))
            _lib.check(rc, "pna_degree_scalers_f32")
            self._scalers[key] = (amp, att)
        return self._scalers[key]

    MAX_EDGE_TYPES = 4        # what the hand-scheduled gather keeps in registers (pna_segreduce_args.edge_type, ABI 14)

    def register_edge_types(self, e, types, rows):
        """Tell the graph that the per-edge feature rows `e` ARE `rows[types]` (an embedding lookup: pna_amd.nets.PNANet's
        `e = embedding_e(bond_type)`), so that edge_type_table(e) answers without hashing the rows (one sort of E doubles and two
        host syncs per fresh `e` tensor: ~0.4 ms per molecule batch against a 0.04 ms layer -- ADVICE r3).  types: integer [E] in the
        caller's edge order; rows: (n_types, edge_dim).  More than MAX_EDGE_TYPES rows: nothing is registered."""
        import weakref
        if rows.shape[0] > self.MAX_EDGE_TYPES or e.dim() != 2 or e.shape[0] != self.csr.col.numel() or types.numel() != e.shape[0]:
            return
        with torch.no_grad():
            res = (types.reshape(-1)[self.csr.eid.long()].to(torch.int32).contiguous(), rows.detach())
        key = (e.data_ptr(), e._version, tuple(e.shape), str(e.device))
        self.__dict__["_edge_types"] = (key, res, weakref.ref(e), True)

    def edge_types_registered(self, e):
        """Whether edge_type_table(e) would answer from register_edge_types: types and rows that were COMPUTED on the device from the
        caller's tensors (no values read on the host) -- safe while a hipGraph is being captured, and a replay recomputes them."""
        hit = self.__dict__.get("_edge_types")
        return (hit is not None and len(hit) > 3 and hit[3] and hit[2]() is e
                and hit[0] == (e.data_ptr(), e._version, tuple(e.shape), str(e.device)))

    def edge_type_table(self, e):
        """(types int32 [E] in CSR order, rows (n_types, edge_dim)) when the per-edge feature rows `e` take at most
        MAX_EDGE_TYPES distinct values -- the molecule nets' edge features are an EMBEDDING of the bond type
        (realworld_benchmark/nets/molecules_graph_regression/pna_net.py: `e = self.embedding_e(e)`), so the W_e . ef part of a
        factorised pretrans is a table with one row per type -- else None.  Found by hashing the rows (one sort of E scalars) and
        VERIFIED against e (a hash collision makes it None, never a wrong table); cached per (tensor object, version).  It reads the
        VALUES of e on the host side of the stream (syncs): callers skip it while a stream is capturing (dgl/pna_layer.py)."""
        import weakref
        key = (e.data_ptr(), e._version, tuple(e.shape), str(e.device))
        hit = self.__dict__.get("_edge_types")
        if hit is not None and hit[0] == key and hit[2]() is e:     # (the SAME tensor object: a freed tensor's address can come back)
            return hit[1]
        res = None
        with torch.no_grad():
            if e.dim() == 2 and e.shape[0] == self.csr.col.numel() and e.shape[0] > 0:
                gen = torch.Generator(device="cpu").manual_seed(0x5eed)
                w = torch.randn(e.shape[1], dtype=torch.float64, generator=gen).to(e.device)
                # continuous edge features fail on a few thousand rows already: no sort of E doubles for them
                probe = e[:4096].double() @ w
                if torch.unique(probe).numel() <= self.MAX_EDGE_TYPES:
                    uniq, inv = torch.unique(e.double() @ w, return_inverse=True)
                    if uniq.numel() <= self.MAX_EDGE_TYPES:
                        rep = torch.zeros(uniq.numel(), dtype=torch.long, device=e.device)
                        rep.scatter_(0, inv, torch.arange(e.shape[0], device=e.device))      # any edge of each type
                        rows = e[rep].contiguous()
                        if torch.equal(rows[inv], e):
                            res = (inv[self.csr.eid].to(torch.int32).contiguous(), rows)
        self.__dict__["_edge_types"] = (key, res, weakref.ref(e))     # (a weak reference: the cache does not pin the feature tensor)
        return res

    def snorm_n(self):
        """Graph-size normalisation 1/sqrt(nodes in the node's graph), (V,1) -- data/molecules.py:157-159."""
        if self._snorm_n is None:
            sizes = torch.tensor(self.batch_num_nodes, dtype=torch.float32, device=self.device)
            self._snorm_n = torch.repeat_interleave((1.0 / sizes).sqrt(), sizes.long()).unsqueeze(1)
        return self._snorm_n


def avg_d_from_degrees(D):
    """The `avg_d` dictionary the sparse nets are configured with: statistics of the in-degrees D of ALL training-set
    nodes -- realworld_benchmark/main_molecules.py:368-372, main_HIV.py:240-244.  `D`: a float/int tensor of in-degrees,
    a Graph, or an iterable of Graphs (their degrees are concatenated, like the reference's `torch.cat`).
    Returns 0-dim fp32 tensors: lin = mean(D), exp = mean(exp(1/D) - 1), log = mean(log(D + 1))."""
    if isinstance(D, Graph):
        D = D.in_degrees()
    elif not torch.is_tensor(D):
        D = torch.cat([g.in_degrees() for g in D])
    D = D.to(torch.float32)
    return dict(lin=torch.mean(D), exp=torch.mean(torch.exp(torch.div(1, D)) - 1), log=torch.mean(torch.log(D + 1)))


def avg_d_from_adjacency(adjs):
    """The dense variant's `avg_d`: the mean over batches of per-batch means of the row sums D = adj.sum(-1) --
    multitask_benchmark/util/train.py:90-94.  `adjs`: an iterable of (B, N, N) adjacency tensors (or one tensor)."""
    if torch.is_tensor(adjs):
        adjs = [adjs]
    dlist = [torch.sum(A, dim=-1) for A in adjs]
    n = len(dlist)
    return dict(lin=sum(torch.mean(D) for D in dlist) / n,
                exp=sum(torch.mean(torch.exp(torch.div(1, D)) - 1) for D in dlist) / n,
                log=sum(torch.mean(torch.log(D + 1)) for D in dlist) / n)
