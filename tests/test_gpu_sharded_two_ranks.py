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
        from pna_amd import Graph
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
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_layers_on_one_gpu():
    mp.spawn(_worker, args=(2, _free_port(), 4000, 40000, 20), nprocs=2, join=True)
