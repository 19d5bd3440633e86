#!/usr/bin/env python
"""Variants of the slab products of autograd._DwPlan at the C3 shape (1 M rows, K = 300 aggregate columns, N = 75 outputs):
operand order, slab length, and how the slabs' scaler values are applied.  HIP events.
    python tools/exp_dw_grouped.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dev = torch.device("cuda:0")
M, K, N, S = 1_000_000, 300, 75, 3


def ev(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best


a = torch.randn(M, K, device=dev)
g = torch.randn(M, N, device=dev)
G3 = torch.randn(M, S * N, device=dev)
Mc = M // 256 * 256
print(f"scaled blocks, 256 slabs (today): {ev(lambda: torch.bmm(G3[:Mc].view(256, -1, S * N).transpose(1, 2), a[:Mc].view(256, -1, K)).sum(0)):.3f} ms")
for SL in (1024, 2048, 4096, 8192):
    nb = M // SL
    av, gv = a[:nb * SL].view(nb, SL, K), g[:nb * SL].view(nb, SL, N)
    t_gn = ev(lambda: torch.bmm(gv.transpose(1, 2), av))                  # (N, K) per slab
    t_kn = ev(lambda: torch.bmm(av.transpose(1, 2), gv))                  # (K, N) per slab
    P = torch.bmm(gv.transpose(1, 2), av)
    w = torch.rand(S, nb, device=dev)
    Pf = P.view(nb, N * K)
    t_w = ev(lambda: w @ Pf)
    w16 = torch.zeros(16, nb, device=dev); w16[:S] = w
    t_w16 = ev(lambda: w16 @ Pf)
    t_wt = ev(lambda: (Pf.t() @ w.t()))
    t_el = ev(lambda: torch.stack([(Pf * w[s][:, None]).sum(0) for s in range(S)]))
    print(f"slab {SL}: bmm gy^T a {t_gn:.3f} ms, a^T gy {t_kn:.3f} ms; slab weights: w @ P {t_w:.3f}, 16-row w {t_w16:.3f}, P^T w^T {t_wt:.3f}, elementwise {t_el:.3f} ms", flush=True)
# the three scaled blocks as ONE batched product over slabs of equal scalers would be (S N, K) per slab -- the same flops as today;
# a 2-slab-wide product (two slabs side by side in the N dimension, block-masked) doubles the tile instead:
for SL in (1024, 2048):
    nb = M // (2 * SL)
    av = a[:nb * 2 * SL].view(nb, 2 * SL, K)
    gv2 = torch.zeros(nb, 2 * SL, 2 * N, device=dev)
    t = ev(lambda: torch.bmm(gv2.transpose(1, 2), av))
    print(f"two slabs of {SL} side by side ((2N, K) per pair, half of gy zeros): {t:.3f} ms")
