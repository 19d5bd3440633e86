"""Test-only DGL 0.4.2 stand-in so that the reference's `models/dgl/pna_layer.py`
runs UNMODIFIED in the build container (DGL itself is not installed, SURVEY.md
Appendix B).  TEST INFRASTRUCTURE ONLY.

It provides exactly what the reference layer touches:
  * `dgl.function.copy_u`                          (pna_layer.py:202)
  * graph.ndata / graph.edata                       (pna_layer.py:56,58,65,199,203)
  * graph.apply_edges(fn)                           (pna_layer.py:61)
  * graph.update_all(message_fn, reduce_fn)         (pna_layer.py:64,202)
with DGL's degree-bucketing semantics: for each distinct in-degree d > 0 the
reduce function sees a mailbox (n_d, d, F) whose edge order is the graph's edge
order restricted to each destination (stable sort by dst).  Zero in-degree
nodes keep zero rows (DGL's zero initialiser; undefined in the reference,
SURVEY.md A.4).

Nothing here is used on the GPU box: /root/reference does not exist there.
"""
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def install():
    """Register stub `dgl` / `dgl.function` modules and put the reference on sys.path."""
    if "dgl" not in sys.modules:
        dgl = types.ModuleType("dgl")
        fn = types.ModuleType("dgl.function")

        def copy_u(src_field, out_field):
            return ("copy_u", src_field, out_field)

        fn.copy_u = copy_u
        fn.copy_src = copy_u
        dgl.function = fn

        # per-graph readouts of a batched graph (nets/*/pna_net.py:83-90): segments = batch_num_nodes
        def _readout(op):
            def f(g, key):
                parts = torch.split(g.ndata[key], g.batch_num_nodes)
                return torch.stack([op(p) for p in parts])
            return f
        dgl.sum_nodes = _readout(lambda p: p.sum(0))
        dgl.mean_nodes = _readout(lambda p: p.mean(0))
        dgl.max_nodes = _readout(lambda p: p.max(0)[0])
        sys.modules["dgl"] = dgl
        sys.modules["dgl.function"] = fn
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


class _EdgeBatch:
    def __init__(self, src, dst, data):
        self.src, self.dst, self.data = src, dst, data


class _NodeBatch:
    def __init__(self, mailbox, data):
        self.mailbox, self.data = mailbox, data


class StandinGraph:
    """Minimal duck-type of a (batched) DGLGraph: directed multigraph src[k] -> dst[k]."""

    def __init__(self, src, dst, num_nodes, batch_num_nodes=None):
        self.src = torch.as_tensor(src, dtype=torch.int64)
        self.dst = torch.as_tensor(dst, dtype=torch.int64)
        self.N = int(num_nodes)
        self.batch_num_nodes = list(batch_num_nodes) if batch_num_nodes is not None else [self.N]
        self.ndata = {}
        self.edata = {}

    def number_of_nodes(self):
        return self.N

    def number_of_edges(self):
        return int(self.src.numel())

    def edges(self):
        return self.src, self.dst

    def _edge_batch(self):
        return _EdgeBatch({k: v[self.src] for k, v in self.ndata.items()},
                          {k: v[self.dst] for k, v in self.ndata.items()},
                          self.edata)

    def apply_edges(self, f):
        self.edata.update(f(self._edge_batch()))

    def update_all(self, message_func, reduce_func):
        if isinstance(message_func, tuple) and message_func[0] == "copy_u":
            _, sf, of = message_func
            msgs = {of: self.ndata[sf][self.src]}
        else:
            msgs = message_func(self._edge_batch())
        deg = torch.bincount(self.dst, minlength=self.N)
        order = torch.sort(self.dst, stable=True)[1]          # in-edges grouped by dst, edge order kept
        rowptr = torch.zeros(self.N + 1, dtype=torch.int64)
        rowptr[1:] = torch.cumsum(deg, 0)
        out = {}
        for d in torch.unique(deg).tolist():
            if d == 0:
                continue
            nodes = torch.nonzero(deg == d).flatten()
            eidx = order[(rowptr[nodes].unsqueeze(1) + torch.arange(d).unsqueeze(0))]   # (n_d, d)
            mailbox = {k: v[eidx] for k, v in msgs.items()}                               # (n_d, d, F)
            res = reduce_func(_NodeBatch(mailbox, {k: v[nodes] for k, v in self.ndata.items()}))
            for k, v in res.items():
                if k not in out:
                    out[k] = torch.zeros((self.N,) + tuple(v.shape[1:]), dtype=v.dtype)
                out[k][nodes] = v
        self.ndata.update(out)
