#!/usr/bin/env python
"""Timing of the parity-test configurations of BASELINE.json (configs[0], [1], [3]) on one MI355X: eager launches
vs hipGraph replay, next to the oracle's CPU restatement on this box's host cores.  These are not bench.py
lines (bench.py reports configs[2]); the numbers go to profiles/r0N_configs.json.

    python tools/bench_configs.py > gpurun_out/configs.json
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import torch_oracle as O  # noqa: E402  (reported CPU baseline only)
from pna_amd import Graph, functional as PF  # noqa: E402
from pna_amd.capture import GraphedForward  # noqa: E402
from pna_amd.dgl.pna_layer import PNALayer, PNASimpleLayer  # noqa: E402
from pna_amd.pytorch.pna.layer import PNALayer as DensePNALayer  # noqa: E402
from pna_amd.synth import molecule_batch  # noqa: E402

dev = torch.device("cuda:0")
AGG, SCA = "mean max min std", "identity amplification attenuation"


def gpu_ms(fn, iters=50, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def cpu_ms(fn, iters=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t0) / iters * 1e3


def randomise(layer):
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            p.copy_(torch.randn(p.shape, generator=gen) / (p.shape[-1] ** 0.5 if p.dim() == 2 else 3.0) +
                    (1.0 if "batchnorm" in n and n.endswith("weight") else 0.0))


out = {"cpu_cores": os.cpu_count(), "torch_threads": torch.get_num_threads()}

# ---- configs[1]: ZINC-shaped batch, PNALayer with 5 towers (realworld_benchmark/configs/...ZINC.json) ----
src, dst, sizes = molecule_batch(128, seed=41)
V, E = sum(sizes), src.numel()
g = Graph(src, dst, V, sizes).to(dev)
deg = g.in_degrees()
avg = {"log": torch.log(deg.double() + 1).mean().float().cpu()}
h = torch.randn(V, 75)
layer = PNALayer(75, 75, AGG, SCA, avg, 0.0, True, True, towers=5, divide_input=False, residual=True).eval()
randomise(layer)
sd = {k: v.clone() for k, v in layer.state_dict().items()}
snorm = g.snorm_n()
lay = layer.to(dev)
hd = h.to(dev)
with torch.no_grad():
    eager = gpu_ms(lambda: lay(g, hd, None, snorm))
    gf = GraphedForward(lambda x: lay(g, x, None, snorm), hd)
    graphed = gpu_ms(lambda: gf(hd))
    # the same layer with TWO-layer pretrans / posttrans MLPs (models/dgl/pna_layer.py:28-32, *_layers > 1): no fused path takes it --
    # per-edge messages are materialised, the MLPs run as library GEMMs around the gather kernel (VERDICT r3 missing #7: never
    # timed); parity: the reference-generated golden tower_deep_mlps (tests/test_gpu_layers.py)
    deep = PNALayer(75, 75, AGG, SCA, avg, 0.0, True, True, towers=5, pretrans_layers=2, posttrans_layers=2, divide_input=False, residual=True).eval()
    randomise(deep)
    deep = deep.to(dev)
    deep_eager = gpu_ms(lambda: deep(g, hd, None, snorm))
    gf_d = GraphedForward(lambda x: deep(g, x, None, snorm), hd)
    deep_graphed = gpu_ms(lambda: gf_d(hd))
    out["zinc_tower_layer_deep_mlps"] = dict(graphs=128, V=V, E=E, hidden=75, towers=5, pretrans_layers=2, posttrans_layers=2,
                                             eager_ms=deep_eager, hipgraph_ms=deep_graphed,
                                             note="general route (messages materialised per edge, library GEMMs for the MLPs); parity by the golden tower_deep_mlps")
    ref = O.dgl_layer_forward(sd, src, dst, V, h, None, snorm.cpu(), AGG.split(), SCA.split(), avg["log"], 5, False, True, True,
                              True, False)
    err = (gf(hd).cpu() - ref).abs().max().item()
    cpu = cpu_ms(lambda: O.dgl_layer_forward(sd, src, dst, V, h, None, snorm.cpu(), AGG.split(), SCA.split(), avg["log"], 5,
                                             False, True, True, True, False))


def build_batch(srcs_l, dsts_l, sizes_l, avg_log):
    """What a data loader's collate does per batch: local edge lists (host) -> batched Graph on the device with every index
    structure the first layer needs (CSR, work list, degree scalers, graph-size norm)."""
    gb = Graph.collate(srcs_l, dsts_l, sizes_l, device=dev)
    gb.work_items()
    gb.degree_scalers(avg_log)
    gb.snorm_n()
    return gb


def split_members(src, dst, sizes):
    """Per-graph local edge lists of a batched edge list (members are contiguous in node id and in edge order)."""
    offs = torch.cumsum(torch.tensor([0] + list(sizes)), 0)
    gid = torch.bucketize(dst, offs[1:], right=True)
    cnt = torch.bincount(gid, minlength=len(sizes)).tolist()
    ss, dd = torch.split(src - offs[gid], cnt), torch.split(dst - offs[gid], cnt)
    return list(ss), list(dd)


def build_batch_flat(src_cat, dst_cat, counts, sizes_l, avg_log):
    """The same from flat arrays (a data set stored as one edge array + per-graph counts): Graph.collate_flat."""
    gb = Graph.collate_flat(src_cat, dst_cat, counts, sizes_l, device=dev)
    gb.work_items()
    gb.degree_scalers(avg_log)
    gb.snorm_n()
    return gb


ss, dd = split_members(src, dst, sizes)
flat = (torch.cat(ss), torch.cat(dd), torch.tensor([int(x.numel()) for x in ss]))
t_build_flat = gpu_ms(lambda: build_batch_flat(*flat, sizes, float(avg["log"])), iters=20)
t_build = gpu_ms(lambda: build_batch(ss, dd, sizes, float(avg["log"])), iters=20)
with torch.no_grad():
    t_build_first = gpu_ms(lambda: lay(build_batch(ss, dd, sizes, float(avg["log"])), hd, None, snorm), iters=20)
out["zinc_tower_layer"] = dict(batch_build_ms=t_build, batch_build_flat_ms=t_build_flat, batch_build_plus_first_layer_ms=t_build_first, graphs=128, V=V, E=E, hidden=75, towers=5, eager_ms=eager, hipgraph_ms=graphed,
                               edges_per_s_hipgraph=E / graphed * 1e3, cpu_oracle_ms=cpu, max_abs_err_vs_oracle=err, max_rel_err_vs_oracle=err / ref.abs().max().item())

# ---- the same ZINC-shaped batch with --edge_feat True: edge features = an embedding (edge_dim 50) of 4 bond types
#      (nets/molecules_graph_regression/pna_net.py); the W_e . ef part of the factorised pretrans is a 4-row table the gather
#      indexes by type (ABI 14).  Round 4: eager calls ride the one-call small-batch kernel (pna_tower_layer_f32 edge_type /
#      edge_table); a captured hipGraph keeps the four-launch route (the type table is read from e's VALUES, capture skips it). ----
small_rows = PF.SMALL_TOWER_ROWS
gen_e = torch.Generator().manual_seed(5)
emb = torch.randn(4, 50, generator=gen_e)
btype = torch.randint(0, 4, (E,), generator=gen_e)
e_feat = emb[btype]
layer_e = PNALayer(75, 75, AGG, SCA, avg, 0.0, True, True, towers=5, divide_input=False, residual=True, edge_features=True, edge_dim=50).eval()
randomise(layer_e)
sd_e = {k: v.clone() for k, v in layer_e.state_dict().items()}
lay_e, ed = layer_e.to(dev), e_feat.to(dev)
with torch.no_grad():
    assert g.edge_type_table(ed) is not None
    eager_e = gpu_ms(lambda: lay_e(g, hd, ed, snorm))
    gf_e = GraphedForward(lambda x: lay_e(g, x, ed, snorm), hd)
    graphed_e = gpu_ms(lambda: gf_e(hd))
    PF.SMALL_TOWER_ROWS = 0                                   # the same four-launch path for the layer WITHOUT edge features
    try:                                                      # (a host-bound leg: timed BEFORE the CPU oracle below, whose OpenMP workers
        eager_noe = gpu_ms(lambda: lay(g, hd, None, snorm))   #  keep spinning for a while and slowed it 10x in the first round-4 set)
        eager_e_general = gpu_ms(lambda: lay_e(g, hd, ed, snorm))      # the edge-feature layer on the four-launch route (type table in registers)
    finally:
        PF.SMALL_TOWER_ROWS = small_rows
    ref_e = O.dgl_layer_forward(sd_e, src, dst, V, h, e_feat, snorm.cpu(), AGG.split(), SCA.split(), avg["log"], 5, False, True, True, True, True)
    err_e = (gf_e(hd).cpu() - ref_e).abs().max().item()
    err_e_eager = (lay_e(g, hd, ed, snorm).cpu() - ref_e).abs().max().item()
out["zinc_tower_layer_edge_feat"] = dict(graphs=128, V=V, E=E, hidden=75, towers=5, edge_dim=50, edge_types=4, eager_ms=eager_e, hipgraph_ms=graphed_e,
                                         same_path_without_edge_features_eager_ms=eager_noe, eager_ms_four_launch_route=eager_e_general,
                                         hipgraph_note="a captured hipGraph takes the per-edge route (the type table is read from e's VALUES: skipped while capturing)", max_abs_err_vs_oracle=err_e, max_abs_err_vs_oracle_eager_one_call_kernel=err_e_eager,
                                         max_rel_err_vs_oracle=err_e / ref_e.abs().max().item())

# ---- configs[3]: MolHIV-shaped batch, PNASimpleLayer hidden 80, 2048 graphs ----
src, dst, sizes = molecule_batch(2048, mean_nodes=25.5, sd_nodes=12, lo=6, hi=222, seed=41, lognormal=True)
V, E = sum(sizes), src.numel()
g = Graph(src, dst, V, sizes).to(dev)
deg = g.in_degrees()
avg = {"log": torch.log(deg.double() + 1).mean().float().cpu()}
h = torch.randn(V, 80)
layer = PNASimpleLayer(80, 80, AGG, SCA, avg, 0.0, True, True).eval()
randomise(layer)
sd = {k: v.clone() for k, v in layer.state_dict().items()}
lay = layer.to(dev)
hd = h.to(dev)
with torch.no_grad():
    eager = gpu_ms(lambda: lay(g, hd))
    gf = GraphedForward(lambda x: lay(g, x), hd)
    graphed = gpu_ms(lambda: gf(hd))
    ref = O.simple_layer_forward(sd, src, dst, V, h, AGG.split(), SCA.split(), avg["log"])
    err = (gf(hd).cpu() - ref).abs().max().item()
    cpu = cpu_ms(lambda: O.simple_layer_forward(sd, src, dst, V, h, AGG.split(), SCA.split(), avg["log"]))
ss, dd = split_members(src, dst, sizes)
flat = (torch.cat(ss), torch.cat(dd), torch.tensor([int(x.numel()) for x in ss]))
t_build_flat = gpu_ms(lambda: build_batch_flat(*flat, sizes, float(avg["log"])), iters=10)
t_build = gpu_ms(lambda: build_batch(ss, dd, sizes, float(avg["log"])), iters=10)
with torch.no_grad():
    t_build_first = gpu_ms(lambda: lay(build_batch(ss, dd, sizes, float(avg["log"])), hd), iters=10)
out["molhiv_simple_layer"] = dict(batch_build_ms=t_build, batch_build_flat_ms=t_build_flat, batch_build_plus_first_layer_ms=t_build_first, graphs=2048, V=V, E=E, hidden=80, eager_ms=eager, hipgraph_ms=graphed,
                                  edges_per_s_hipgraph=E / graphed * 1e3, cpu_oracle_ms=cpu, max_abs_err_vs_oracle=err, max_rel_err_vs_oracle=err / ref.abs().max().item())

# ---- configs[3] as BASELINE.json WORDS it ("8 towers"): an EXTENSION -- the reference's HIV net has no towers
# (nets/HIV_graph_classification/pna_net.py:30-38); this is DGL PNALayer(towers=8, divide_input=True) at hidden 80 (10 features per
# tower) on the same 2048-graph batch; golden: tests/golden/tower_hiv_t8_div.npz ----
layer8 = PNALayer(80, 80, AGG, SCA, avg, 0.0, True, True, towers=8, divide_input=True, residual=True).eval()
randomise(layer8)
sd8 = {k: v.clone() for k, v in layer8.state_dict().items()}
lay8 = layer8.to(dev)
snorm8 = g.snorm_n()
with torch.no_grad():
    eager8 = gpu_ms(lambda: lay8(g, hd, None, snorm8))
    gf8 = GraphedForward(lambda x: lay8(g, x, None, snorm8), hd)
    graphed8 = gpu_ms(lambda: gf8(hd))
    ref8 = O.dgl_layer_forward(sd8, src, dst, V, h, None, snorm8.cpu(), AGG.split(), SCA.split(), avg["log"], 8, True, True, True, True, False)
    err8 = (gf8(hd).cpu() - ref8).abs().max().item()
out["molhiv_8_towers_extension"] = dict(note="not a reference configuration: BASELINE.json's wording of configs[3]; PNALayer(towers=8, divide_input=True), 80 -> 80",
                                        graphs=2048, V=V, E=E, hidden=80, towers=8, eager_ms=eager8, hipgraph_ms=graphed8,
                                        edges_per_s_hipgraph=E / graphed8 * 1e3, max_abs_err_vs_oracle=err8, max_rel_err_vs_oracle=err8 / ref8.abs().max().item())

# ---- the whole MolHIV net of the reference's README (PNASimpleLayer x 4, hidden 80, mean readout), same batch ----
from pna_amd.nets import PNANetHIV  # noqa: E402
net = PNANetHIV(dict(hidden_dim=80, out_dim=80, in_feat_dropout=0.0, dropout=0.3, L=4, readout="mean", batch_norm=True,
                     residual=True, aggregators=AGG, scalers=SCA, avg_d=avg, posttrans_layers=1, device=dev)).to(dev)
gen = torch.Generator().manual_seed(0)
atoms = torch.stack([torch.randint(0, d, (V,), generator=gen) for d in (119, 4, 12, 12, 10, 6, 6, 2, 2)], dim=1).to(dev)
labels = torch.randint(0, 2, (len(sizes),), generator=gen)
net.eval()
with torch.no_grad():
    net_eager = gpu_ms(lambda: net(g, atoms))
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
net.train()


def train_step():
    opt.zero_grad()
    loss = net.loss(net(g, atoms), labels)
    loss.backward()
    opt.step()


net_train = gpu_ms(train_step, iters=20)
out["molhiv_net_4_layers"] = dict(graphs=2048, V=V, E=E, hidden=80, L=4, inference_eager_ms=net_eager,
                                  graphs_per_s_inference=2048 / net_eager * 1e3, train_step_ms=net_train,
                                  graphs_per_s_training=2048 / net_train * 1e3)

# ---- the whole ZINC net of the reference's README with --edge_feat True (PNALayer x 4, hidden 75, 5 towers, edge_dim 50, sum
#      readout; realworld_benchmark/README.md:62), a 128-molecule batch: FRESH edge-feature tensors every call (e = embedding_e(bonds)),
#      the bond types handed to the graph by the net (Graph.register_edge_types) ----
from pna_amd.nets import PNANet  # noqa: E402
srcz, dstz, sizesz = molecule_batch(128, seed=41)
Vz, Ez = sum(sizesz), srcz.numel()
gz = Graph(srcz, dstz, Vz, sizesz).to(dev)
avgz = {"log": torch.log(gz.in_degrees().double() + 1).mean().float().cpu()}
znet = PNANet(dict(num_atom_type=28, num_bond_type=4, hidden_dim=75, out_dim=70, in_feat_dropout=0.0, dropout=0.0, L=4, readout="sum",
                   graph_norm=True, batch_norm=True, residual=True, aggregators=AGG, scalers=SCA, avg_d=avgz, towers=5,
                   divide_input_first=False, divide_input_last=True, edge_feat=True, edge_dim=50, pretrans_layers=1, posttrans_layers=1,
                   gru=False, device=dev)).to(dev).eval()
genz = torch.Generator().manual_seed(2)
atomsz = torch.randint(0, 28, (Vz,), generator=genz).to(dev)
bondsz = torch.randint(0, 4, (Ez,), generator=genz).to(dev)
snz = gz.snorm_n()
with torch.no_grad():
    znet_eager = gpu_ms(lambda: znet(gz, atomsz, bondsz, snz, None))
    gfz = GraphedForward(lambda a_, b_: znet(gz, a_, b_, snz, None), atomsz, bondsz)      # atom AND bond types are inputs of the graph
    znet_graphed = gpu_ms(lambda: gfz(atomsz, bondsz))
out["zinc_net_4_layers_edge_feat"] = dict(graphs=128, V=Vz, E=Ez, hidden=75, towers=5, L=4, edge_dim=50, bond_types=4, inference_eager_ms=znet_eager,
                                          graphs_per_s_inference=128 / znet_eager * 1e3, inference_hipgraph_ms=znet_graphed,
                                          graphs_per_s_hipgraph=128 / znet_graphed * 1e3,
                                          note="fresh e = embedding_e(bonds) every call; layers 1-3 on the one-call kernel with the edge-type table, the last "
                                               "layer (divide_input_last) too")

# ---- the superpixels net of the reference's CIFAR10 config (configs/superpixels_graph_classification_pna_CIFAR10.json: batch 128, L 4,
#      hidden 75 -> out 70, 5 towers, divide_input_first=True / _last=False, sum readout, no edge features): 128 graphs x ~117 superpixels x
#      8 nearest neighbours = ~15 k nodes / ~120 k edges -- the MID-SIZE regime between the one-call small-batch kernel
#      (functional.SMALL_TOWER_ROWS) and the degree-planned paths (VERDICT r4 missing #6) ----
import numpy as np  # noqa: E402
from pna_amd.nets import PNANetSuperpixels  # noqa: E402
rng_c = np.random.default_rng(41)
srcs_c, dsts_c, sizes_c, off_c = [], [], [], 0
for _ in range(128):
    n_c = int(np.clip(round(rng_c.normal(117.6, 4.0)), 85, 150))
    pos_c = rng_c.random((n_c, 2))
    d2_c = ((pos_c[:, None, :] - pos_c[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2_c, np.inf)
    nb_c = np.argsort(d2_c, axis=1)[:, :8]
    dsts_c.append(np.repeat(np.arange(n_c), 8) + off_c)
    srcs_c.append(nb_c.reshape(-1) + off_c)
    sizes_c.append(n_c)
    off_c += n_c
src_c, dst_c = torch.from_numpy(np.concatenate(srcs_c)), torch.from_numpy(np.concatenate(dsts_c))
Vs, Es = int(sum(sizes_c)), int(src_c.numel())
gs = Graph(src_c, dst_c, Vs, sizes_c).to(dev)
avgs = {"log": torch.log(gs.in_degrees().double() + 1).mean().float().cpu()}
snet = PNANetSuperpixels(dict(in_dim=5, in_dim_edge=1, hidden_dim=75, out_dim=70, n_classes=10, in_feat_dropout=0.0, dropout=0.0, L=4, readout="sum",
                              graph_norm=True, batch_norm=True, residual=True, aggregators=AGG, scalers=SCA, avg_d=avgs, towers=5,
                              divide_input_first=True, divide_input_last=False, edge_feat=False, edge_dim=0, pretrans_layers=1, posttrans_layers=1,
                              gru=False, device=dev)).to(dev).eval()
xs_c = torch.rand(Vs, 5, device=dev)
sns = gs.snorm_n()
with torch.no_grad():
    snet_eager = gpu_ms(lambda: snet(gs, xs_c, None, sns, None))
    gfs = GraphedForward(lambda x_: snet(gs, x_, None, sns, None), xs_c)
    snet_graphed = gpu_ms(lambda: gfs(xs_c))
    hs_c = torch.randn(Vs, 75, device=dev)
    first_eager = gpu_ms(lambda: snet.layers[0](gs, hs_c, None, sns))
    last_eager = gpu_ms(lambda: snet.layers[-1](gs, hs_c, None, sns))
out["cifar10_superpixels_net_4_layers"] = dict(graphs=128, V=Vs, E=Es, in_dim=5, hidden=75, out_dim=70, towers=5, L=4, divide_input_first=True,
                                               divide_input_last=False, inference_eager_ms=snet_eager, inference_hipgraph_ms=snet_graphed,
                                               graphs_per_s_hipgraph=128 / snet_graphed * 1e3, edges_per_s_per_layer_hipgraph=4 * Es / snet_graphed * 1e3,
                                               first_layer_eager_ms=first_eager, last_layer_eager_ms=last_eager,
                                               small_batch_kernel_rows_limit=PF.SMALL_TOWER_ROWS)

# ---- configs[0]: multitask dense layer, B=128 graphs of N=50 nodes, hidden 16, 4 towers ----
gen = torch.Generator().manual_seed(1234)
B, N = 128, 50
adj = (torch.rand(B, N, N, generator=gen) < 0.1).float()
adj = torch.maximum(adj, adj.transpose(1, 2)) * (1 - torch.eye(N))
ring = torch.zeros(N, N)
i = torch.arange(N)
ring[i, (i + 1) % N] = 1
ring[(i + 1) % N, i] = 1
adj = torch.maximum(adj, ring.unsqueeze(0))
D = adj.sum(-1)
avg_d = dict(lin=D.mean(), log=torch.log(D + 1).mean())
x = torch.randn(B, N, 16, generator=gen)
layer = DensePNALayer(16, 16, AGG.split(), ["identity"], avg_d, towers=4, divide_input=True).eval()
randomise(layer)
sd = {k: v.clone() for k, v in layer.state_dict().items()}
lay = layer.to(dev)
xd, ad = x.to(dev), adj.to(dev)
avg_dev = {k: v.to(dev) for k, v in avg_d.items()}
for t in lay.towers:
    t.avg_d = avg_dev
with torch.no_grad():
    eager = gpu_ms(lambda: lay(xd, ad))
    ref = O.dense_layer_forward(sd, x, adj, AGG.split(), ["identity"], avg_d, 4, True)
    err = (lay(xd, ad).cpu() - ref).abs().max().item()
    cpu = cpu_ms(lambda: O.dense_layer_forward(sd, x, adj, AGG.split(), ["identity"], avg_d, 4, True))
E = int(adj.sum().item())
out["multitask_dense_layer"] = dict(B=B, N=N, hidden=16, towers=4, directed_edges=E, eager_ms=eager,
                                    edges_per_s_eager=E / eager * 1e3, cpu_oracle_ms=cpu, max_abs_err_vs_oracle=err, max_rel_err_vs_oracle=err / ref.abs().max().item())

# ---- the TOWER variant at roofline scale (T = 1, pretrans Linear(150 -> 75) factorised to node level,
# message = W_a h_src + (W_b h_dst + b)): its aggregation runs on the hand-scheduled kernel since round 2 ----
from pna_amd import functional as PF  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402
Vc, Ec, Fc = 1_000_000, 10_000_000, 75
s3, d3 = powerlaw_graph(Vc, Ec, seed=1234, device=dev)
g3 = Graph(s3, d3, Vc)
avg3 = {"log": torch.log(g3.in_degrees().double() + 1).mean().float().cpu()}
h3 = torch.randn(Vc, Fc, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))
sn3 = torch.ones(Vc, 1, device=dev)
tower = PNALayer(Fc, Fc, AGG, SCA, avg3, 0.0, True, True, towers=1, divide_input=False, residual=True).eval()
randomise(tower)
tower = tower.to(dev)
simple = PNASimpleLayer(Fc, Fc, AGG, SCA, avg3, 0.0, True, True).eval()
randomise(simple)
simple = simple.to(dev)
xs = torch.randn(Vc, 80, device=dev)[:, :Fc]
xd3 = torch.randn(Vc, 80, device=dev)[:, :Fc]
with torch.no_grad():
    t_tower = gpu_ms(lambda: tower(g3, h3, None, sn3), iters=20)
    t_simple = gpu_ms(lambda: simple(g3, h3), iters=20)
    t_agg_tower = gpu_ms(lambda: PF.aggregate(g3, xs, Fc, AGG.split(), dst_term=xd3), iters=20)
    t_agg_simple = gpu_ms(lambda: PF.aggregate(g3, xs, Fc, AGG.split()), iters=20)
out["c3_tower_layer"] = dict(V=Vc, E=Ec, F=Fc, towers=1, layer_ms=t_tower, simple_layer_ms=t_simple, aggregate_tower_ms=t_agg_tower,
                             aggregate_simple_ms=t_agg_simple, edges_per_s_layer=Ec / t_tower * 1e3)
# ---- the reference's 5-tower layers at the same scale.  divide_input=True (CIFAR10 / MNIST first layers, ZINC's last:
#      tower t reads the input slice [15 t, 15 t + 15)): all towers' messages are 75 features per edge -- ONE gather, the one-kernel tower
#      layer (round 5).  divide_input=False (ZINC's first layers): five different 150 -> 75 projections = 375 message features per edge,
#      5 x the gather bytes (15.2 GB per layer: >= 3 ms at the 5 TB/s the gather sustains) -- the two-kernel grouped path. ----
from pna_amd import degree_groups as DGc  # noqa: E402
res5 = {}
for div in (True, False):
    t5 = PNALayer(Fc, Fc, AGG, SCA, avg3, 0.0, True, True, towers=5, divide_input=div, residual=True).eval()
    randomise(t5)
    t5 = t5.to(dev)
    with torch.no_grad():
        one = bool(PF.tower_layer_degree_grouped_applies(t5, g3, h3) and PF.tower_layer_degree_fused_applies(t5, g3, h3))
        ms5 = gpu_ms(lambda: t5(g3, h3, None, sn3), iters=10, warmup=3)
        entry = dict(layer_ms=ms5, one_kernel_path=one, message_features_per_edge=75 if div else 375, edges_per_s_layer=Ec / ms5 * 1e3)
        if one:
            keep_f = DGc.FUSED
            DGc.FUSED = False
            entry["two_kernel_grouped_path_ms"] = gpu_ms(lambda: t5(g3, h3, None, sn3), iters=10, warmup=3)
            DGc.FUSED = keep_f
    res5["divide_input" if div else "whole_input"] = entry
    del t5
out["c3_tower_layer_5_towers"] = dict(V=Vc, E=Ec, F=Fc, towers=5, **res5)
print(json.dumps(out, indent=1))
