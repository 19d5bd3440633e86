#!/usr/bin/env python
"""Forward + backward of one PNASimpleLayer on the roofline workload (C3) -- not a bench.py line (the metric is the
forward layer); documents what the training path costs.  Prints one JSON object."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import Graph  # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

dev = torch.device("cuda:0")
V, E, F = 1_000_000, 10_000_000, 75
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
avg_log = float(torch.log(g.in_degrees().float() + 1).mean())
layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", {"log": avg_log}, 0.0, True, True).to(dev).train()
h = torch.randn(V, F, device=dev, requires_grad=True)


def step():
    # like the reference's loop (train_molecules_graph_regression.py:29-32: optimizer.zero_grad() every iteration): gradients start
    # empty, nothing is ACCUMULATED into last step's tensors (until round 4 this tool left them in place: one extra add of (V, F) per step)
    h.grad = None
    layer.zero_grad(set_to_none=True)
    out = layer(g, h)
    out.sum().backward()


def timed(fn, n=20, warmup=12, repeats=2):
    """best of `repeats` runs of n steps after `warmup` steps (the first steps of a process allocate: 27 device allocations in the
    first dozen training steps at C3; with 3 warm-up steps the same code measured anything from 7.7 to 10.8 ms)"""
    for _ in range(warmup):
        fn()
    best = 1e9
    for _ in range(repeats):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / n * 1e3)
    return best


gy_next = torch.randn(V, F, device=dev)


def step_given_gy():
    """The same step with the output's gradient GIVEN (what the next layer's backward hands this one), instead of made by a loss over
    the layer's own (V, F) output: `out.sum().backward()` costs a (V, F) reduction and a (V, F) fill of ones that no layer inside a
    network pays."""
    h.grad = None
    layer.zero_grad(set_to_none=True)
    layer(g, h).backward(gy_next)


def fwd_only():
    with torch.no_grad():
        layer(g, h)


res = {"workload": "C3 PNASimpleLayer, train mode (batch-stat BatchNorm), V=1M E=10M F=75",
       "fwd_bwd_ms": timed(step), "fwd_bwd_given_output_gradient_ms": timed(step_given_gy), "fwd_train_mode_nograd_ms": timed(fwd_only)}
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda r: -r.device_time_total)[:14]
res["top_kernels_us"] = {r.key[:70]: round(r.device_time_total, 1) for r in rows}
# a roofline block per kernel of the step: ALGORITHMIC bytes of the kernel's launches in one step / their summed
# device time (torch profiler, one traced step), against 8 TB/s.  N = F here; the arg ranks are 16-bit, the arg indices 32-bit.
N = F
HBM = 8.0e12
alg = {
    "k_segreduce_fast": ("forward gather with arg tracking: E (4F + 4) read; V (16F aggregate + 8F arg indices) written", E * (4 * F + 4) + V * 24 * F),
    "k_posttrans_x3": ("forward contraction (V (16F + 4F) read, 4N written) + d agg = gy W^T (V 4N read, 16F written): two launches", V * (20 * F + 4 * N) + V * (4 * N + 16 * F)),
    "k_bwd_rowprep": ("rowprep + ranks, packed rows: V (16F d agg + 8F mean / std + 8F arg indices) read; V (16F R1 | R2 | G_max | G_min + 4F ranks) written", V * 52 * F),
    "k_bwd_pull<": ("pull over the transposed graph (rounds 3-5): per out-edge R1 | R2 (8F), G_max | G_min (8F), two 16-bit rank rows (4F), the id and its rank (8); V (4F x + 4F grad) per row", E * (20 * F + 8) + V * 8 * F),
    "k_bwd_edge_rows": ("round 6, per-edge rows: V (16F d agg + 8F mean / std + 8F arg indices) read; E x 4F edge rows + V x 4F R2 written", V * 36 * F + E * 4 * F),
    "k_bwd_pull_rows": ("round 6, pull over the edge rows: per out-edge the edge's row (4F) + R2 of its destination (4F) + the id and the position (8); V (4F x + 4F grad) per row", E * (8 * F + 8) + V * 8 * F),
    "k_posttrans_dw_grouped(": ("weight gradient in degree-plan order: V (4N gy + 16F aggregate + 4F h) read once", V * (4 * N + 20 * F)),
    "k_posttrans_dw(": ("weight gradient with per-row scalers: gy read by three column thirds", V * (12 * N + 20 * F)),
    "k_bn_apply": ("BatchNorm tail forward + backward element-wise passes", V * 4 * N * 3 + V * 4 * N * 3),
    "k_bn_colsums": ("BatchNorm tail column sums, forward + backward", V * 4 * N + V * 8 * N),
}
roof = []
for key, (what, nbytes) in alg.items():
    hits = [r for r in prof.key_averages() if key in r.key]
    if hits:
        us = sum(r.device_time_total for r in hits)
        roof.append({"kernel": key.rstrip("("), "what": what, "launches": int(sum(r.count for r in hits)), "us": round(us, 1),
                     "algorithmic_bytes": int(nbytes), "achieved_GB_per_s": nbytes / us / 1e3, "frac_of_8_TB_per_s": nbytes / (us * 1e-6) / HBM})
res["roofline_per_kernel"] = roof
lib = [r.key[:60] for r in prof.key_averages() if r.key.startswith("Cijk_") and r.device_time_total > 50]
res["vendor_gemm_kernels_over_50us"] = lib
print(json.dumps(res, indent=1))
