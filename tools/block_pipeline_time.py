#!/usr/bin/env python
"""Compute side of shard.BlockPipeline on ONE GPU (world size 1: no exchange): a stack of L PNASimpleLayers over the C3 graph cut into
B row blocks, the blocks through the one-kernel layer with their own degree plans (round 4) against the gather + three-block
contraction per block (round 3).  -> gpurun_out/r04_block_pipeline_time.json"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
dist.init_process_group("gloo", rank=0, world_size=1)
from pna_amd import Graph, functional as PF  # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer  # noqa: E402
from pna_amd.shard import BlockPipeline, shard_graph  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

dev = torch.device("cuda:0")
V, E, F, L = 1_000_000, 10_000_000, 75, 4
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
gs, g = shard_graph(src, dst, V), Graph(src, dst, V)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
torch.manual_seed(0)
layers = [PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True).to(dev).eval() for _ in range(L)]
with torch.no_grad():
    for lay in layers:
        for p in lay.parameters():
            p.copy_(torch.randn_like(p) / (p.shape[-1] ** 0.5 if p.dim() == 2 else 3.0))
h = torch.randn(V, F, device=dev)
out = {"V": V, "E": E, "F": F, "layers": L}
with torch.no_grad():
    x = torch.zeros(V, 80, device=dev)[:, :F]
    x.copy_(h)
    want = x
    for lay in layers:
        want = lay(g, want)
    for B in (1, 4, 8):
        for fused in (False, True):
            pipe = BlockPipeline(gs, B)
            ta = torch.zeros(gs.num_nodes + gs.n_halo, 80, device=dev)
            tb = torch.zeros_like(ta)
            ta[:V, :F] = h
            rows = PF.SimpleLayerRows(layers, gs, B, fused=fused)
            for _ in range(3):
                ta[:V, :F] = h
                res = pipe.run(rows, L, ta, tb)
            torch.cuda.synchronize()
            err = (res[:V, :F] - want).abs().max().item() / want.abs().max().item()
            n = 10
            t0 = time.perf_counter()
            for _ in range(n):
                pipe.run(rows, L, ta, tb)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            out[f"blocks{B}_{'one_kernel_per_block' if fused else 'gather_plus_three_block'}"] = {"ms_per_layer": ms / L, "max_err_rel_to_max_vs_unsharded": err}
            print(B, fused, ms / L, err, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_block_pipeline_time.json"), "w"), indent=1)
dist.destroy_process_group()
