#!/usr/bin/env python
"""posttrans: the bf16x3 path vs the f32-MFMA path vs float64 (development tool)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def ref64(a, K, W, scales, b, h):
    M, S = a.shape[0], len(scales)
    Kh = 0 if h is None else h.shape[1]
    y = b.double()[None, :].repeat(M, 1)
    if h is not None:
        y = y + h.double() @ W[:, :Kh].double().t()
    mass = b.abs().double()[None, :].repeat(M, 1)
    if h is not None:
        mass = mass + h.abs().double() @ W[:, :Kh].abs().double().t()
    for s in range(S):
        sc = torch.ones(M, device=dev, dtype=torch.float64) if scales[s] is None else scales[s].double()
        Ws = W[:, Kh + s * K:Kh + (s + 1) * K].double()
        y = y + sc[:, None] * (a[:, :K].double() @ Ws.t())
        mass = mass + sc.abs()[:, None] * (a[:, :K].abs().double() @ Ws.abs().t())
    return y, mass


def case(M, K, N, S, Kh, seed=0, scale=1.0):
    gen = torch.Generator().manual_seed(seed)
    a = (torch.randn(M, K, generator=gen) * scale).to(dev)
    h = torch.randn(M, Kh, generator=gen).to(dev) if Kh else None
    W = (torch.randn(N, Kh + S * K, generator=gen) / (S * K) ** 0.5).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    scales = [None] + [(torch.rand(M, generator=gen) + 0.5).to(dev) for _ in range(S - 1)]
    y64, mass = ref64(a, K, W, scales, b, h)
    out = {}
    for arith in ("f32", "bf16x3"):
        y = ops.posttrans(a, K, W, scales, b, h, arith=arith)
        out[arith] = ((y.double() - y64).abs() / mass).max().item()
    ok = out["bf16x3"] <= max(4 * out["f32"], 2e-7)
    print(f"M={M} K={K} N={N} S={S} Kh={Kh}: err/mass f32={out['f32']:.2e} bf16x3={out['bf16x3']:.2e} {'ok' if ok else 'BAD'}", flush=True)
    return ok


ok = True
for args in [(1000, 300, 75, 3, 0), (257, 300, 75, 3, 0), (64, 32, 16, 1, 0), (100, 12, 5, 2, 0), (300, 280, 70, 3, 70), (129, 33, 40, 2, 7),
             (50, 4, 80, 3, 4), (513, 900, 150, 1, 0), (1000, 300, 75, 3, 0, 1, 1e4), (777, 64, 48, 3, 16)]:
    ok &= case(*args)
print("ACCURACY", "PASS" if ok else "FAIL", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "time":
    M, K, N, S = 1_000_000, 300, 75, 3
    pitch = int(os.environ.get("A_PITCH", K))
    a = torch.randn(M, pitch, device=dev)[:, :K]
    W = torch.randn(N, S * K, device=dev) / 30
    b = torch.randn(N, device=dev)
    scales = [None, torch.rand(M, device=dev), torch.rand(M, device=dev)]
    res = torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev)
    for arith in ("f32", "bf16x3"):
        fn = lambda: ops.posttrans(a, K, W, scales, b, out=y, relu=True, residual=res, arith=arith)  # noqa: E731
        for _ in range(3):
            fn()
        ms = 1e9
        for _ in range(5):                       # best of 5 batches of 20 launches
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            ms = min(ms, (time.perf_counter() - t) / 20 * 1e3)
        print(f"{arith}: {ms:.3f} ms  ({2 * M * K * N * S / ms / 1e9:.1f} TF/s fp32-equivalent)", flush=True)
