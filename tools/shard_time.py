#!/usr/bin/env python
"""Set-up cost of one rank's shard at BASELINE configs[4] size: V = 16 M, E = 160 M, 8 ranks, on ONE
MI355X without a process group -- shard_local (filter this rank's in-edges out of the global list, renumber the sources into
[local | halo], de-duplicate the halo per peer), then everything a layer needs from the shard (CSR, work lists of the halo
overlap, degree plan): seconds and peak device bytes per rank.  The global edge list itself (2 x int64 x E = 2.56 GB) is the
caller's, as in shard_graph's contract.

    python tools/shard_time.py [out.json] [--scale 1.0]
"""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd.shard import HaloGraph, partition_bounds, shard_local
from pna_amd.synth import powerlaw_graph
from pna_amd import degree_groups as DG

scale = float(sys.argv[sys.argv.index("--scale") + 1]) if "--scale" in sys.argv else 1.0
dev = torch.device("cuda:0")
V, E, W = int(16_000_000 * scale), int(160_000_000 * scale), 8
t0 = time.time()
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
torch.cuda.synchronize()
res = {"V": V, "E": E, "world": W, "generate_global_edge_list_s": time.time() - t0, "global_edge_list_bytes": src.numel() * 16}
for balance in ("nodes", "edges"):
    bounds = partition_bounds(V, W, dst, balance)
    per = []
    for rank in (0, W - 1):
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        torch.cuda.synchronize(); t = time.time()
        src_ext, d, n_local, recv_lists, recv_splits = shard_local(src, dst, bounds, rank)
        torch.cuda.synchronize(); t_local = time.time() - t
        peak_local = torch.cuda.max_memory_allocated() - base
        n_halo = sum(recv_splits)
        t = time.time()
        g = HaloGraph(src_ext, d, n_local, n_halo, src_ext.new_empty(0), [0] * W, recv_splits, None, bounds[rank], bounds[rank + 1], V, any_exchange=False)
        g.csr; g.work_items(); g.heavy_schedule(); g.split_work_lists()
        torch.cuda.synchronize(); t_graph = time.time() - t
        t = time.time()
        plan = DG.plan_of(g); plan.split_items(g)
        torch.cuda.synchronize(); t_plan = time.time() - t
        per.append(dict(rank=rank, local_rows=n_local, in_edges=int(d.numel()), halo_rows=n_halo, shard_local_s=t_local, shard_local_peak_bytes=peak_local,
                        csr_and_work_lists_s=t_graph, degree_plan_s=t_plan, peak_bytes_total=torch.cuda.max_memory_allocated() - base))
        print(balance, per[-1], flush=True)
        del g, plan, src_ext, d
    res[balance] = per
print(json.dumps(res, indent=1))
if len(sys.argv) > 1 and not sys.argv[1].startswith("--"):
    json.dump(res, open(sys.argv[1], "w"), indent=1)
