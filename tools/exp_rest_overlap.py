#!/usr/bin/env python
"""Can the rest-row launches (hub rows / rare degrees: gather + finalize + three-block contraction, 0.087 ms of dependent small
kernels) hide beside the one-kernel layer?  The persistent kernel books every register of every CU, so launches on a second stream
wait for its workgroups to retire.  Here the kernel leaves SPARE of its 512 workgroups out
(pna_fused_degree_args.spare_workgroups) and the rest launches go to a second stream, before or after the kernel's launch.

    SPARE=32 python tools/exp_rest_overlap.py          (profiles/r03_exp_rest_overlap.log: 0, 8, 16, 24, 32, 40, 48, 64)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph, degree_groups as DG, functional as PF  # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

dev = torch.device("cuda:0")
V, E, F = 1_000_000, 10_000_000, 75
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
torch.manual_seed(0)
layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True).to(dev).eval()
h = torch.randn(V, (F + 7) // 8 * 8, device=dev)[:, :F]


def ev(fn, n=20, reps=5):
    for _ in range(5):
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


with torch.no_grad():
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    call = PF.FusedDegreeCall(layer, g, h)
    call.args.spare_workgroups = int(os.environ.get("SPARE", "32"))
    spare_product, DG.FUSED_SPARE_WGS = DG.FUSED_SPARE_WGS, 0
    y_ref = PF.simple_layer_degree_fused(layer, g, h).clone()                 # (the serial order)
    DG.FUSED_SPARE_WGS = spare_product
    with torch.cuda.stream(side):
        call_side = PF.FusedDegreeCall(layer, g, h, out=call.y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(), torch.cuda.Event()

    def serial():
        call.group_rows()
        call.rest_rows()

    def overlapped():
        e0.record(main)
        side.wait_event(e0)
        with torch.cuda.stream(side):
            call_side.rest_rows()
            e1.record(side)
        call.group_rows()
        main.wait_event(e1)

    def overlapped_fused_first():
        e0.record(main)
        call.group_rows()
        side.wait_event(e0)
        with torch.cuda.stream(side):
            call_side.rest_rows()
            e1.record(side)
        main.wait_event(e1)

    call.y.zero_()
    overlapped()
    torch.cuda.synchronize()
    print("overlapped result identical to the serial one:", bool(torch.equal(call.y, y_ref)))
    print(f"the product path (functional.run_fused_call, {DG.FUSED_SPARE_WGS} spare): {ev(lambda: PF.simple_layer_degree_fused(layer, g, h, out=call.y)):.4f} ms")
    for rep in range(2):
        print(f"spare workgroups {call.args.spare_workgroups}: serial {ev(serial):.4f} ms, overlapped {ev(overlapped):.4f} ms (kernel launched first: {ev(overlapped_fused_first):.4f} ms), group rows alone {ev(call.group_rows):.4f} ms", flush=True)
