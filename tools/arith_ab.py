#!/usr/bin/env python
"""Same-box A/B of the one-kernel layer's arithmetics (degree_groups.FUSED_ARITH: guarded | fp16x2 | bf16x3) on the C3 layer (V = 1 M,
E = 10 M, 75 -> 75) and on BASELINE configs[4]'s per-GPU shape (V = 2 M, E = 20 M, 128 -> 128): group-rows launch(es) alone, round robin,
best of `reps` x 20 launches per round.  Boxes of the pool differ by 4-5 %: only figures of ONE run compare.

    python tools/arith_ab.py [rounds]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, degree_groups as DG, functional as PF   # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer                   # noqa: E402
from pna_amd.synth import powerlaw_graph                           # noqa: E402

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def ev(fn, n=20, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best


out = {}
for name, V, E, F, pitch in (("c3", 1_000_000, 10_000_000, 75, 80), ("c5_shard_shape", 2_000_000, 20_000_000, 128, 128)):
    src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
    g = Graph(src, dst, V)
    avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
    torch.manual_seed(0)
    layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True).to(dev).eval()
    h = torch.randn(V, pitch, device=dev)[:, :F]
    calls = {}
    with torch.no_grad():
        for ar in ("fp16x2", "guarded", "bf16x3"):
            DG.FUSED_ARITH = ar
            c = calls[ar] = PF.FusedDegreeCall(layer, g, h, x=h)
            c.set_spare(False)
        DG.FUSED_ARITH = "guarded"
        res = {ar: [] for ar in calls}
        for r in range(rounds):
            for ar, c in calls.items():
                res[ar].append(ev(c.group_rows))
        plan = DG.plan_of(g)
        DG.guard_stats(plan, dev, reset=True)
        calls["guarded"].group_rows()
        torch.cuda.synchronize()
        handed = DG.guard_stats(plan, dev)[0]
    tiles = plan.NV // 64
    out[name] = {ar: round(min(v), 4) for ar, v in res.items()}
    out[name]["tiles_handed_over"] = [handed, tiles]
    print(name, out[name], flush=True)
    del g, h, calls, layer
    torch.cuda.empty_cache()
print("RESULT " + json.dumps(out))
