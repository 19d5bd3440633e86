#!/usr/bin/env python
"""CPU dry run of the destination-range sharding on the BASELINE multi-GPU workloads (no GPU, no process group): per rank,
the number of de-duplicated halo rows, interior rows (all in-edges local) and in-edges, for

    C3 x 8  : V = 8 M, E = 80 M (bench.py --gpus 8, weak scaling of configs[2]),  F = 75
    C5      : V = 16 M, E = 160 M (configs[4]),                                    F = 128

under (a) contiguous ranges of equal node count, (b) equal in-edge count, (c) ranges over the BFS renumbering (bfs_order).
The halo byte volume per layer = halo_rows * F * 4.  Writes profiles/r02_halo_dryrun.json.

    python tools/halo_dryrun.py [--scale 1.0]        (--scale 0.125 for a quick look)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd.shard import bfs_order, partition_bounds  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402


def shard_stats(src, dst, V, world, balance):
    bounds = partition_bounds(V, world, dst, balance)
    bt = torch.tensor(bounds)
    owner_s = torch.searchsorted(bt, src, right=True) - 1
    owner_d = torch.searchsorted(bt, dst, right=True) - 1
    out = []
    remote_per_row = torch.zeros(V, dtype=torch.int32).index_add_(0, dst, (owner_s != owner_d).to(torch.int32))
    for r in range(world):
        mine = owner_d == r
        rem = src[mine & (owner_s != r)]
        halo = int(torch.unique(rem).numel())
        lo, hi = bounds[r], bounds[r + 1]
        out.append(dict(rank=r, nodes=hi - lo, in_edges=int(mine.sum()), remote_in_edges=int(rem.numel()), halo_rows=halo,
                        interior_rows=int((remote_per_row[lo:hi] == 0).sum())))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    args = ap.parse_args()
    res = {"scale": args.scale}
    for name, V, E, F in (("c3x8", 8_000_000, 80_000_000, 75), ("c5", 16_000_000, 160_000_000, 128)):
        V, E = int(V * args.scale), int(E * args.scale)
        E -= (E - 2 * V) % 2
        t0 = time.time()
        src, dst = powerlaw_graph(V, E, seed=1234)
        entry = {"V": V, "E": E, "F": F, "world": 8}
        for label, balance, reorder in (("nodes", "nodes", False), ("edges", "edges", False), ("bfs+nodes", "nodes", True)):
            s, d = src, dst
            if reorder:
                order = bfs_order(src, dst, V)
                new_id = torch.empty(V, dtype=torch.long)
                new_id[order] = torch.arange(V)
                s, d = new_id[src], new_id[dst]
            st = shard_stats(s, d, V, 8, balance)
            halo = [x["halo_rows"] for x in st]
            entry[label] = dict(per_rank=st, halo_rows_max=max(halo), halo_rows_mean=sum(halo) / len(halo),
                                halo_bytes_per_layer_max=max(halo) * F * 4, in_edges_max=max(x["in_edges"] for x in st),
                                in_edges_min=min(x["in_edges"] for x in st),
                                interior_fraction_mean=sum(x["interior_rows"] for x in st) / V)
            print(name, label, {k: v for k, v in entry[label].items() if k != "per_rank"}, f"({time.time() - t0:.0f} s)", flush=True)
        res[name] = entry
    with open(os.path.join(ROOT, "profiles", "r02_halo_dryrun.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
