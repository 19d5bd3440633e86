"""pna_collate_csr_i32 (device batching + stable destination-sorted CSR) against the torch
sort/bincount construction used on the CPU: every array bit-identical (the sort is stable: DGL's mailbox order)."""
import pytest
import torch

from pna_amd.graph import Graph, build_csr

pytestmark = pytest.mark.gpu


def _same(a, b):
    for x, y, name in zip(a[:4], b[:4], ("rowptr", "col", "eid", "row")):
        assert torch.equal(x.cpu().long(), y.cpu().long()), name
    assert a.max_degree == b.max_degree


@pytest.mark.parametrize("V,E", [(1, 0), (1, 5), (7, 0), (100, 700), (1000, 1), (5000, 60000), (3, 1000), (2 ** 17 + 3, 300000)])
def test_device_csr_equals_host_csr(cuda_device, V, E):
    gen = torch.Generator().manual_seed(V + E)
    src = torch.randint(0, V, (E,), generator=gen)
    dst = torch.randint(0, max(1, V - V // 10), (E,), generator=gen)       # the last rows stay empty
    if E > 10:
        dst[: E // 4] = dst[0]                                              # a hub: long runs of equal keys (stability)
    _same(build_csr(src.to(cuda_device), dst.to(cuda_device), V), build_csr(src, dst, V))


def test_device_collate_of_a_molecule_batch(cuda_device):
    gen = torch.Generator().manual_seed(1)
    sizes = [int(n) for n in torch.randint(9, 38, (128,), generator=gen)]
    srcs = [torch.randint(0, n, (2 * n + 2,), generator=gen) for n in sizes]
    dsts = [torch.randint(0, n, (2 * n + 2,), generator=gen) for n in sizes]
    host = Graph.batch([Graph(s, d, n) for s, d, n in zip(srcs, dsts, sizes)])
    dev = Graph.collate(srcs, dsts, sizes, device=cuda_device)
    assert dev.batch_num_nodes == host.batch_num_nodes and dev.num_nodes == host.num_nodes
    assert torch.equal(dev.src.cpu(), host.src) and torch.equal(dev.dst.cpu(), host.dst)
    _same(dev.csr, host.csr)
    torch.testing.assert_close(dev.snorm_n().cpu(), host.snorm_n())


def test_graph_on_device_uses_the_native_collate_and_layers_still_match(cuda_device):
    """End to end: a layer on a Graph whose CSR came from the device build gives the golden output."""
    from conftest import load_golden
    from pna_amd.dgl.pna_layer import PNASimpleLayer
    meta, a, sd = load_golden("simple_f75")
    layer = PNASimpleLayer(meta["F"], meta["out_dim"], meta["aggregators"], meta["scalers"], {"log": a["avg_log"]}, 0.0,
                           True, meta["residual"], posttrans_layers=meta["posttrans_layers"])
    layer.load_state_dict(sd)
    layer = layer.to(cuda_device).eval()
    g = Graph(a["src"].to(cuda_device), a["dst"].to(cuda_device), meta["N"])
    with torch.no_grad():
        out = layer(g, a["h"].to(cuda_device)).cpu()
    torch.testing.assert_close(out, a["out"], rtol=1e-5, atol=1e-5)


def test_collate_flat_matches_collate(cuda_device):
    """Graph.collate_flat (flat edge arrays + per-graph counts, no per-graph host work) builds the same batch as collate()."""
    import torch
    from pna_amd import Graph
    gen = torch.Generator().manual_seed(11)
    sizes = [int(n) for n in torch.randint(1, 40, (300,), generator=gen)]
    srcs, dsts = [], []
    for n in sizes:
        e = int(torch.randint(0, 4 * n, (1,), generator=gen))
        srcs.append(torch.randint(0, n, (e,), generator=gen))
        dsts.append(torch.randint(0, n, (e,), generator=gen))
    a = Graph.collate(srcs, dsts, sizes, device=cuda_device)
    b = Graph.collate_flat(torch.cat(srcs), torch.cat(dsts), torch.tensor([s.numel() for s in srcs]), torch.tensor(sizes), device=cuda_device)
    assert a.num_nodes == b.num_nodes and a.batch_num_nodes == b.batch_num_nodes
    assert torch.equal(a.src, b.src) and torch.equal(a.dst, b.dst)
    assert torch.equal(a.csr.rowptr, b.csr.rowptr) and torch.equal(a.csr.col, b.csr.col)
    import pytest
    with pytest.raises(ValueError):
        Graph.collate_flat(torch.cat(srcs), torch.cat(dsts), torch.tensor([1, 2]), torch.tensor(sizes), device=cuda_device)
