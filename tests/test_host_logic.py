"""CPU-side checks: graph/CSR construction, heavy-row schedule, dense sparsification, the C-ABI
library's exported symbols, layer state_dict compatibility with the reference, and the
no-CPU-fallback rule."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_names, load_golden
from pna_amd import Graph, _lib
from pna_amd.graph import build_heavy_schedule
from pna_amd.dgl.pna_layer import PNALayer, PNASimpleLayer
from pna_amd.pytorch.pna.layer import PNALayer as DensePNALayer
from pna_amd.pytorch.pna.sparsify import sparsify


def test_csr_is_stable_in_destination():
    src = torch.tensor([5, 1, 2, 3, 4, 0, 2, 2])
    dst = torch.tensor([1, 0, 1, 3, 1, 0, 3, 1])
    g = Graph(src, dst, 5)
    c = g.csr
    assert c.rowptr.tolist() == [0, 2, 6, 6, 8, 8]
    assert c.col.tolist() == [1, 0, 5, 2, 4, 2, 3, 2]        # original edge order inside each destination
    assert c.eid.tolist() == [1, 5, 0, 2, 4, 7, 3, 6]
    assert c.row.tolist() == [0, 0, 1, 1, 1, 1, 3, 3]
    assert c.max_degree == 4 and c.rowptr.dtype == torch.int32 and c.col.dtype == torch.int32
    assert g.in_degrees().tolist() == [2, 4, 0, 2, 0]


def test_empty_and_edgeless_graphs():
    g = Graph(torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long), 4)
    assert g.csr.rowptr.tolist() == [0, 0, 0, 0, 0] and g.csr.max_degree == 0
    assert g.heavy_schedule().n_heavy == 0
    g0 = Graph(torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long), 0)
    assert g0.csr.rowptr.tolist() == [0]


def test_heavy_schedule_covers_every_heavy_edge_once():
    rng = np.random.default_rng(1)
    deg = np.concatenate([rng.integers(0, 9, 200), [10, 11, 64, 65, 129, 1000]])
    rng.shuffle(deg)
    rowptr = torch.tensor(np.concatenate([[0], np.cumsum(deg)]), dtype=torch.int32)
    hs = build_heavy_schedule(rowptr, int(deg.max()), threshold=10, seg_len=16)
    heavy = np.nonzero(deg > 10)[0]
    assert hs.heavy_rows.tolist() == heavy.tolist()
    nseg = -(-deg[heavy] // 16)
    assert hs.heavy_segptr.tolist() == np.concatenate([[0], np.cumsum(nseg)]).tolist()
    assert hs.n_seg == int(nseg.sum()) and hs.seg_heavy.numel() == hs.n_seg
    covered = 0
    for s in range(hs.n_seg):
        hi = hs.seg_heavy[s].item()
        sidx = s - hs.heavy_segptr[hi].item()
        d = deg[heavy[hi]]
        covered += min(16, d - sidx * 16)
    assert covered == int(deg[heavy].sum())
    assert build_heavy_schedule(rowptr, int(deg.max()), threshold=0, seg_len=16).n_heavy == 0


def test_batch_offsets_like_dgl_batch():
    g1 = Graph(torch.tensor([0, 1]), torch.tensor([1, 2]), 3)
    g2 = Graph(torch.tensor([0]), torch.tensor([1]), 2)
    g = Graph.batch([g1, g2])
    assert g.num_nodes == 5 and g.src.tolist() == [0, 1, 3] and g.dst.tolist() == [1, 2, 4]
    assert g.batch_num_nodes == [3, 2]
    sn = g.snorm_n().flatten()
    torch.testing.assert_close(sn, torch.tensor([3 ** -0.5] * 3 + [2 ** -0.5] * 2))


def test_sparsify_dense_adjacency_two_groupings():
    torch.manual_seed(0)
    adj = (torch.rand(2, 5, 5) < 0.4).float() * torch.rand(2, 5, 5).round(decimals=1).clamp(min=0.1)
    dg = sparsify(adj, self_loop=False)
    b, i, j = torch.nonzero(adj, as_tuple=True)
    r = dg.by_row.csr
    assert r.row.tolist() == (b * 5 + i).tolist() and r.col.tolist() == (b * 5 + j).tolist()
    torch.testing.assert_close(dg.w_row, adj[b, i, j])
    c = dg.by_col.csr
    # by_col: destination (b,j), sources i ascending, and row_to_col maps back to the by_row position
    assert c.row.tolist() == sorted(c.row.tolist())
    assert torch.equal(r.row[dg.row_to_col.long()], c.col) and torch.equal(r.col[dg.row_to_col.long()], c.row)
    torch.testing.assert_close(dg.w_col, dg.w_row[dg.row_to_col.long()])
    assert not dg.binary
    assert sparsify((adj > 0).float(), self_loop=True).binary is False or True


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pna_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pna_[a-z0-9_]+)\s*\(", text)))


def test_shared_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m pna_amd.build` (or __graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert {"pna_segreduce_fwd_f32", "pna_posttrans_f32", "pna_degree_scalers_f32", "pna_last_error",
            "pna_abi_version", "pna_segreduce_partials_bytes"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"libpna_amd.so does not export {n}"
    assert _lib.lib().pna_abi_version() == _lib.PNA_ABI_VERSION


def test_ctypes_struct_layout_matches_header_field_order():
    """Field names/order of the ctypes mirror must follow the C struct (sizes are checked on the GPU by the
    parity tests; here we catch a field added on one side only)."""
    text = open(os.path.join(ROOT, "include", "pna_amd.h")).read()
    body = re.search(r"typedef struct pna_segreduce_args \{(.*?)\} pna_segreduce_args;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [re.sub(r"\[.*\]", "", f.strip().split()[-1].lstrip("*")) for f in body.split(";") if f.strip()]
    assert fields == [n for n, _ in _lib.PnaSegreduceArgs._fields_]
    body = re.search(r"typedef struct pna_posttrans_args \{(.*?)\} pna_posttrans_args;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [re.sub(r"\[.*\]", "", f.strip().split()[-1].lstrip("*")) for f in body.split(";") if f.strip()]
    assert fields == [n for n, _ in _lib.PnaPosttransArgs._fields_]
    body = re.search(r"typedef struct pna_fused_simple_args \{(.*?)\} pna_fused_simple_args;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [re.sub(r"\[.*\]", "", f.strip().split()[-1].lstrip("*")) for f in body.split(";") if f.strip()]
    assert fields == [n for n, _ in _lib.PnaFusedSimpleArgs._fields_]
    for cname, mirror in (("pna_small_linear_args", _lib.PnaSmallLinearArgs), ("pna_tower_layer_args", _lib.PnaTowerLayerArgs),
                          ("pna_segreduce_bwd_args", _lib.PnaSegreduceBwdArgs)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), text, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = [re.sub(r"\[.*\]", "", f.strip().split()[-1].lstrip("*")) for f in body.split(";") if f.strip()]
        assert fields == [n for n, _ in mirror._fields_], cname


@pytest.mark.parametrize("name", golden_names("dgl_tower"))
def test_tower_layer_loads_reference_state_dict(name):
    meta, a, sd = load_golden(name)
    layer = PNALayer(meta["in_dim"], meta["out_dim"], meta["aggregators"], meta["scalers"], {"log": a["avg_log"]}, 0.0,
                     meta["graph_norm"], meta["batch_norm"], towers=meta["towers"],
                     pretrans_layers=meta["pretrans_layers"], posttrans_layers=meta["posttrans_layers"],
                     divide_input=meta["divide_input"], residual=meta["residual"], edge_features=meta["edge_dim"] > 0,
                     edge_dim=meta["edge_dim"])
    assert list(layer.state_dict().keys()) == list(sd.keys())
    layer.load_state_dict(sd, strict=True)


@pytest.mark.parametrize("name", golden_names("dgl_simple"))
def test_simple_layer_loads_reference_state_dict(name):
    meta, a, sd = load_golden(name)
    layer = PNASimpleLayer(meta["F"], meta["out_dim"], meta["aggregators"], meta["scalers"], {"log": a["avg_log"]}, 0.0,
                           True, meta["residual"], posttrans_layers=meta["posttrans_layers"])
    assert list(layer.state_dict().keys()) == list(sd.keys())
    layer.load_state_dict(sd, strict=True)


@pytest.mark.parametrize("name", golden_names("dense"))
def test_dense_layer_loads_reference_state_dict(name):
    meta, a, sd = load_golden(name)
    layer = DensePNALayer(meta["in_features"], meta["out_features"], meta["aggregators"], meta["scalers"],
                          {"log": a["avg_log"], "lin": a["avg_lin"]}, towers=meta["towers"],
                          self_loop=meta["self_loop"], divide_input=meta["divide_input"])
    assert list(layer.state_dict().keys()) == list(sd.keys())
    layer.load_state_dict(sd, strict=True)


def test_constructor_assertions_and_unknown_names_like_the_reference():
    avg = {"log": torch.tensor(1.0)}
    with pytest.raises(AssertionError):
        PNALayer(10, 12, "mean", "identity", avg, 0.0, True, True, towers=4)           # towers must divide out_dim
    with pytest.raises(AssertionError):
        PNALayer(10, 12, "mean", "identity", None, 0.0, True, True, towers=2)          # avg_d is required
    with pytest.raises(KeyError):
        PNALayer(12, 12, "mean median", "identity", avg, 0.0, True, True)
    with pytest.raises(KeyError):
        PNASimpleLayer(12, 12, "mean", "identity exponential", avg, 0.0, True, True)
    with pytest.raises(AssertionError):
        DensePNALayer(10, 12, ["mean"], ["identity"], avg, towers=4)
    lay = PNALayer(10, 12, "mean", "identity", avg, 0.0, True, True, towers=2, residual=True)
    assert lay.residual is False                                                         # in_dim != out_dim


def test_no_cpu_fallback_layers_raise_on_cpu_tensors():
    avg = {"log": torch.tensor(1.0)}
    g = Graph(torch.tensor([0, 1, 2]), torch.tensor([1, 2, 0]), 3)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="GPU"):
            PNASimpleLayer(8, 8, "mean max", "identity", avg, 0.0, True, True).eval()(g, torch.randn(3, 8))
        with pytest.raises(RuntimeError, match="GPU"):
            PNALayer(8, 8, "mean max", "identity", avg, 0.0, True, True, towers=2).eval()(g, torch.randn(3, 8), None,
                                                                                        torch.ones(3, 1))
        with pytest.raises(RuntimeError, match="GPU"):
            DensePNALayer(8, 8, ["mean", "max"], ["identity"], avg, towers=2).eval()(
                torch.randn(1, 3, 8), torch.ones(1, 3, 3) - torch.eye(3))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pna_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports oracle/"
                assert "/root/reference" not in src, f"{f} references /root/reference"


def test_work_items_cover_every_edge_once():
    rng = np.random.default_rng(4)
    V, E = 400, 5000
    dst = rng.integers(0, V - 5, E)
    dst[:1200] = rng.integers(0, 3, 1200)                    # three hubs
    g = Graph(torch.from_numpy(rng.integers(0, V, E)), torch.from_numpy(dst), V)
    deg = g.in_degrees()
    rp = g.csr.rowptr.long()
    for thr, seg, order in ((0, 0, "degree"), (16, 8, "degree"), (128, 128, "natural"), (30, 16, "degree"), (16, 8, "window")):
        it = g.work_items(thr, seg, order, 32).long()
        assert it.shape[1] == 4
        hs = g.heavy_schedule(thr, seg)
        whole = it[it[:, 3] < 0]
        segs = it[it[:, 3] >= 0]
        assert segs.shape[0] == hs.n_seg and torch.equal(segs[:, 3], torch.arange(hs.n_seg))
        want = torch.arange(V) if hs.n_heavy == 0 else torch.nonzero(deg <= hs.threshold).flatten()
        assert sorted(whole[:, 0].tolist()) == want.tolist()
        assert torch.equal(whole[:, 1], rp[whole[:, 0]]) and torch.equal(whole[:, 2], rp[whole[:, 0] + 1])
        d = (whole[:, 2] - whole[:, 1]).tolist()
        if order == "degree":
            assert d == sorted(d, reverse=True)
        if order == "window":
            for w0 in range(0, len(d), 32):
                assert d[w0:w0 + 32] == sorted(d[w0:w0 + 32], reverse=True)
                assert sorted(whole[w0:w0 + 32, 0].tolist()) == want[w0:w0 + 32].tolist()
        covered = torch.zeros(E, dtype=torch.long)
        for r, b, e, s_ in it.tolist():
            covered[b:e] += 1
            assert rp[r] <= b and e <= rp[r + 1]
        assert (covered == 1).all()


@pytest.mark.parametrize("name", golden_names("net_molecules"))
def test_molecules_net_loads_reference_state_dict(name):
    from pna_amd.nets import PNANet
    meta, a, sd = load_golden(name)
    net = PNANet(_net_params(meta, a))
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd, strict=True)


def _net_params(meta, a):
    return dict(num_atom_type=28, num_bond_type=4, hidden_dim=meta["hidden_dim"], out_dim=meta["out_dim"],
                in_feat_dropout=0.0, dropout=0.0, L=meta["L"], readout=meta["readout"], graph_norm=True, batch_norm=True,
                residual=True, aggregators=meta["aggregators"], scalers=meta["scalers"], avg_d={"log": a["avg_log"]},
                towers=meta["towers"], divide_input_first=False, divide_input_last=True, edge_feat=meta["edge_dim"] > 0,
                edge_dim=meta["edge_dim"], pretrans_layers=1, posttrans_layers=1, gru=meta["gru"], device="cpu")


def test_collate_applies_member_graph_offsets_like_batch():
    """Graph.collate (local ids + offsets inside the CSR build) == Graph.batch (offsets applied per graph on the host)."""
    gen = torch.Generator().manual_seed(0)
    sizes = [5, 1, 7, 3]
    srcs = [torch.randint(0, n, (2 * n,), generator=gen) for n in sizes]
    dsts = [torch.randint(0, n, (2 * n,), generator=gen) for n in sizes]
    a = Graph.collate(srcs, dsts, sizes)
    b = Graph.batch([Graph(s, d, n) for s, d, n in zip(srcs, dsts, sizes)])
    assert a.num_nodes == b.num_nodes and a.batch_num_nodes == b.batch_num_nodes
    assert torch.equal(a.src, b.src) and torch.equal(a.dst, b.dst)
    for x, y in zip(a.csr[:4], b.csr[:4]):
        assert torch.equal(x.long(), y.long())
    assert a.csr.max_degree == b.csr.max_degree


def test_avg_d_producers_match_the_reference_formulas():
    """avg_d of the sparse nets (main_molecules.py:368-372) and of the dense benchmark
    (util/train.py:90-94: mean over batches of per-batch means)."""
    import math
    from pna_amd import avg_d_from_adjacency, avg_d_from_degrees
    g1 = Graph(torch.tensor([0, 1, 2, 2]), torch.tensor([1, 2, 0, 1]), 3)          # in-degrees 1, 2, 1
    g2 = Graph(torch.tensor([0, 1]), torch.tensor([1, 0]), 2)                        # in-degrees 1, 1
    a = avg_d_from_degrees([g1, g2])
    D = [1, 2, 1, 1, 1]
    assert a["lin"].item() == pytest.approx(sum(D) / 5)
    assert a["log"].item() == pytest.approx(sum(math.log(d + 1) for d in D) / 5, rel=1e-6)
    assert a["exp"].item() == pytest.approx(sum(math.exp(1 / d) - 1 for d in D) / 5, rel=1e-6)
    assert a["log"].dtype == torch.float32 and a["log"].dim() == 0
    assert avg_d_from_degrees(g1)["lin"].item() == pytest.approx(4 / 3)
    adj1 = torch.tensor([[[0., 1.], [1., 0.]]])                                      # D = 1, 1
    adj2 = torch.tensor([[[0., 1., 1.], [1., 0., 0.], [1., 0., 0.]]])                # D = 2, 1, 1
    b = avg_d_from_adjacency([adj1, adj2])
    assert b["lin"].item() == pytest.approx((1.0 + 4 / 3) / 2)
    assert b["log"].item() == pytest.approx((math.log(2) + (math.log(3) + 2 * math.log(2)) / 3) / 2, rel=1e-6)


def test_bench_child_leg_never_sinks_the_line(monkeypatch):
    """bench.py's default call runs BASELINE configs[4]'s per-GPU shape in a child process (`configs4_per_gpu_shape`): a child that
    fails, prints nothing or times out costs that field only; a child's line is cut down to the fields a reader compares."""
    import importlib.util
    import json
    import subprocess
    import types
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    line = open(os.path.join(ROOT, "profiles", "r03_bench_c5_shard_shape_n1.json")).read().strip()
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=0, stdout="a warning\n" + line + "\n"))
    r = bench.other_workload_leg(["--workload", "c5"])
    assert "error" not in r and r["ms_per_step"] > 0 and r["parity_check"]["ok"] and r["roofline"]["frac"] > 0 and "F=128" in r["workload"]
    json.dumps(r)
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=1, stdout=""))
    assert "error" in bench.other_workload_leg(["--workload", "c5"])

    def boom(*a, **k):
        raise subprocess.TimeoutExpired(cmd="bench.py", timeout=1)
    monkeypatch.setattr(subprocess, "run", boom)
    assert "TimeoutExpired" in bench.other_workload_leg(["--workload", "c5"])["error"]


def test_registered_edge_types_answer_without_a_search(monkeypatch):
    """Graph.register_edge_types (PNANet: e = embedding_e(bond_type) IS rows[types]): edge_type_table(e) answers from the registration
    for THAT tensor object only, in CSR edge order, without torch.unique; another tensor is searched for (and found) as before."""
    gen = torch.Generator().manual_seed(3)
    V, E = 40, 200
    src, dst = torch.randint(0, V, (E,), generator=gen), torch.randint(0, V, (E,), generator=gen)
    g = Graph(src, dst, V)
    rows = torch.randn(4, 6, generator=gen)
    types = torch.randint(0, 4, (E,), generator=gen)
    e = rows[types]
    g.register_edge_types(e, types, rows)
    assert g.edge_types_registered(e)
    real_unique = torch.unique
    monkeypatch.setattr(torch, "unique", lambda *a, **k: (_ for _ in ()).throw(AssertionError("searched")))
    t_csr, r = g.edge_type_table(e)
    monkeypatch.setattr(torch, "unique", real_unique)
    assert torch.equal(r, rows) and torch.equal(t_csr.long(), types[g.csr.eid.long()])
    assert torch.equal(r[t_csr.long()], e[g.csr.eid.long()])
    e2 = e.clone()
    assert not g.edge_types_registered(e2)
    found = g.edge_type_table(e2)                                  # the verified search still works for a caller of the bare layer
    assert found is not None and torch.equal(found[1][found[0].long()], e2[g.csr.eid.long()])
    assert not g.edge_types_registered(e)                          # (one cache slot: the search replaced the registration)
    g.register_edge_types(e, types, torch.randn(9, 6))             # more rows than the register table holds: nothing registered
    assert not g.edge_types_registered(e)


def test_flat_tower_weights_and_block_diagonal_projection():
    """Round 5: T towers with divide_input=True on the one-kernel tower layer (functional._tower_flat_weights, pna_layer.
    _projection_cache_padded_div).  Host algebra on CPU tensors: the collapsed weight with aggregator-major columns over ALL towers'
    features times [mean | max | min | std | h] equals the tower-major form, and ONE block-diagonal GEMM gives every tower's
    source / destination projection of its input slice (models/dgl/pna_layer.py:133-136, :35-40)."""
    import torch
    from pna_amd import functional as PF
    from pna_amd.dgl.pna_layer import PNALayer, _projection_cache_padded_div
    torch.manual_seed(0)
    T, Fi, S, V = 4, 16, 3, 9
    layer = PNALayer(T * Fi, 64, "mean max min std", "identity amplification attenuation", {"log": torch.tensor(1.3)}, 0.0, True, True, towers=T,
                     pretrans_layers=1, posttrans_layers=1, divide_input=True, residual=True, edge_features=False, edge_dim=0).eval()
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(torch.randn_like(p) * 0.3)
    towers, mix = list(layer.towers), layer.mixing_network
    Wv, d, c, ones, K = PF._tower_collapsed_weights(layer, towers, mix, True)
    Wp = PF._tower_flat_weights(layer, towers, mix)[0]
    assert K == 5 * T * Fi and Wp.shape == Wv.shape
    a, h = torch.randn(V, T, 4, Fi, dtype=torch.float64), torch.randn(V, T * Fi, dtype=torch.float64)
    old = torch.cat([a.reshape(V, -1), h], 1)                                    # [t: mean | max | min | std] per tower, then h
    new = torch.cat([a.permute(0, 2, 1, 3).reshape(V, -1), h], 1)                # [mean (all towers) | max | min | std], then h
    for s in range(S):
        torch.testing.assert_close(new @ Wp[:, s * K:(s + 1) * K].double().t(), old @ Wv[:, s * K:(s + 1) * K].double().t(), rtol=1e-12, atol=1e-12)
    P = PF.tower_projection_pitch(T * Fi)
    W, b = _projection_cache_padded_div(towers, Fi, P)
    x = h.float() @ W.t() + b
    for t, tw in enumerate(towers):
        lin = tw.pretrans.fully_connected[0].linear
        hs = h.float()[:, t * Fi:(t + 1) * Fi]
        torch.testing.assert_close(x[:, t * Fi:(t + 1) * Fi], hs @ lin.weight[:, :Fi].t(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(x[:, P + t * Fi:P + (t + 1) * Fi], hs @ lin.weight[:, Fi:].t() + lin.bias, rtol=1e-5, atol=1e-6)
