"""world_size-2 (and 3) CPU tests of the multi-GPU path over the gloo backend: destination-range
sharding, halo de-duplication, the all-to-all exchange and its backward.  The aggregation kernel itself
needs a GPU, so here the exchange is validated against the global feature table directly: after
`source_features`, gathering with the shard's remapped source ids must equal gathering the global table
with the original ids -- which is exactly the property that makes the sharded kernel results identical
to the single-GPU ones."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

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
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pna_amd.shard import partition_bounds, shard_graph
        from pna_amd.synth import powerlaw_graph
        src, dst = powerlaw_graph(V, E, seed=3)
        x = torch.randn(V, F, generator=torch.Generator().manual_seed(0))          # global table, same on all ranks
        g = shard_graph(src, dst, V)
        b = partition_bounds(V, world)
        lo, hi = b[rank], b[rank + 1]
        assert g.num_nodes == hi - lo and sum(g.recv_splits) == g.n_halo
        x_local = x[lo:hi].clone().requires_grad_(True)
        x_ext = g.source_features(x_local)
        assert x_ext.shape == (g.num_nodes + g.n_halo, F)
        # 1. every local in-edge sees exactly the global source row
        mine = (dst >= lo) & (dst < hi)
        assert torch.equal(g.dst, dst[mine] - lo)
        assert torch.equal(x_ext[g.src], x[src[mine]])
        # 2. halos are de-duplicated: each remote row is received once
        halo_ids = g.src[g.src >= g.num_nodes]
        assert torch.unique(halo_ids).numel() == g.n_halo
        remote = src[mine][(src[mine] < lo) | (src[mine] >= hi)]
        assert torch.unique(remote).numel() == g.n_halo
        # 3. CSR of the shard = rows [lo,hi) of the global CSR
        from pna_amd.graph import build_csr
        cg = build_csr(src, dst, V)
        cl = g.csr
        assert torch.equal(cl.rowptr.long(), cg.rowptr[lo:hi + 1].long() - cg.rowptr[lo].long())
        e0, e1 = int(cg.rowptr[lo]), int(cg.rowptr[hi])
        assert torch.equal(x_ext[cl.col.long()], x[cg.col[e0:e1].long()])
        # 4. backward of the exchange = scatter-add of halo gradients into the owners' rows:
        #    d/dx sum_e w_e . x[src_e]  summed over ALL ranks' edges must match the global computation
        w = torch.randn(E, F, generator=torch.Generator().manual_seed(1))
        loss = (x_ext[g.src] * w[mine]).sum()
        loss.backward()
        xg = x.clone().requires_grad_(True)
        (xg[src] * w).sum().backward()
        torch.testing.assert_close(x_local.grad, xg.grad[lo:hi], rtol=1e-5, atol=1e-5)
        # 5. resident extended table (inference path): features written into alloc_features() are exchanged in place,
        #    with a row pitch wider than F; the same gather identity must hold and the local rows stay untouched
        h_res = g.alloc_features(F, pitch=F + 3)
        h_res.copy_(x[lo:hi])
        with torch.no_grad():
            ext2 = g.source_features(h_res)
        assert ext2.shape == (g.num_nodes + g.n_halo, F) and ext2.stride(0) == F + 3
        assert ext2.untyped_storage().data_ptr() == h_res.untyped_storage().data_ptr()       # no concatenation happened
        assert torch.equal(ext2[g.src], x[src[mine]])
        assert torch.equal(ext2[: g.num_nodes], x[lo:hi])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_halo_exchange_gloo(world):
    mp.spawn(_worker, args=(world, _free_port(), 500, 6000, 5), nprocs=world, join=True)


def test_partition_bounds_cover_all_nodes():
    from pna_amd.shard import partition_bounds
    for V in (0, 1, 7, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            b = partition_bounds(V, w)
            assert b[0] == 0 and b[-1] == V and all(b[i] <= b[i + 1] for i in range(w))
            assert max(b[i + 1] - b[i] for i in range(w)) - min(b[i + 1] - b[i] for i in range(w)) <= 1
