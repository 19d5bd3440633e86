#!/usr/bin/env python
"""posttrans: the 32x32-tile bf16x3 kernel (pna_posttrans_x3w_f32) vs the 16x16 one vs float64 (development tool).

    python tools/x3w_check.py [time]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import ops  # noqa: E402

ops.X3_WIDE = True
dev = torch.device("cuda:0")


def ref64(a, K, W, scales, b):
    M, S = a.shape[0], len(scales)
    y = b.double()[None, :].repeat(M, 1)
    mass = b.abs().double()[None, :].repeat(M, 1)
    for s in range(S):
        sc = torch.ones(M, device=dev, dtype=torch.float64) if scales[s] is None else scales[s].double()
        Ws = W[:, s * K:(s + 1) * K].double()
        y = y + sc[:, None] * (a[:, :K].double() @ Ws.t())
        mass = mass + sc.abs()[:, None] * (a[:, :K].abs().double() @ Ws.abs().t())
    return y, mass


def run(a, K, W, scales, b, wide, **kw):
    _keep = ops.X3_WIDE
    ops.X3_WIDE = wide
    try:
        return ops.posttrans(a, K, W, scales, b, arith="bf16x3", **kw)
    finally:
        ops.X3_WIDE = _keep


def case(M, K, N, S=3, seed=0, tail=False, ldy=None, lda=None):
    gen = torch.Generator().manual_seed(seed)
    a = torch.randn(M, lda or K, generator=gen).to(dev)[:, :K]
    W = (torch.randn(N, S * K, generator=gen) / (S * K) ** 0.5).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    scales = [None] + [(torch.rand(M, generator=gen) + 0.5).to(dev) for _ in range(S - 1)]
    assert ops.x3w_supported(K, N, S, 0), (K, N, S)
    y64, mass = ref64(a, K, W, scales, b)
    kw = {}
    if tail:
        kw = dict(row_post=(torch.rand(M, generator=gen) + 0.5).to(dev), col_scale=(torch.rand(N, generator=gen) + 0.5).to(dev),
                  col_shift=torch.randn(N, generator=gen).to(dev), relu=True, residual=torch.randn(M, N, generator=gen).to(dev))
        z = (y64 * kw["row_post"].double()[:, None]) * kw["col_scale"].double() + kw["col_shift"].double()
        y64 = kw["residual"].double() + torch.relu(z)
        mass = mass * kw["row_post"].double()[:, None] * kw["col_scale"].double() + kw["col_shift"].abs().double() + kw["residual"].abs().double()
    errs = {}
    for wide in (False, True):
        out = None
        if ldy:
            out = torch.full((M, ldy), float("nan"), device=dev)[:, :N]
        y = run(a, K, W, scales, b, wide, out=out, **kw)
        errs[wide] = ((y.double() - y64).abs() / mass).max().item()
        if ldy:
            assert torch.isnan(out.as_strided((M, ldy - N), (ldy, 1), N)).all(), "wrote into the padding"
    ok = errs[True] <= max(4 * errs[False], 3e-7)
    print(f"M={M} K={K} N={N} tail={tail} ldy={ldy}: err/mass 16x16={errs[False]:.2e} 32x32={errs[True]:.2e} {'ok' if ok else 'BAD'}", flush=True)
    return ok


ok = True
for kw in [dict(M=1000, K=300, N=75), dict(M=257, K=300, N=75, tail=True), dict(M=31, K=300, N=75), dict(M=4096, K=300, N=75, tail=True, ldy=80),
           dict(M=1000, K=320, N=80, tail=True), dict(M=777, K=300, N=70, tail=True), dict(M=513, K=256, N=64), dict(M=2000, K=512, N=128, tail=True),
           dict(M=300, K=44, N=75), dict(M=300, K=20, N=66, tail=True), dict(M=70000, K=300, N=75, tail=True, ldy=80, lda=304),
           dict(M=1000, K=300, N=75, lda=301, ldy=77, tail=True), dict(M=1, K=4, N=64)]:
    try:
        ok &= case(**kw)
    except Exception as e:  # noqa: BLE001
        ok = False
        print("EXC", kw, repr(e), flush=True)
print("ACCURACY", "PASS" if ok else "FAIL", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "time":
    for (M, F, N) in [(1_000_000, 75, 75), (2_000_000, 128, 128)]:
        K, S = 4 * F, 3
        a = torch.randn(M, K, device=dev)
        W = torch.randn(N, S * K, device=dev) / 30
        b = torch.randn(N, device=dev)
        scales = [None, torch.rand(M, device=dev), torch.rand(M, device=dev)]
        ldy = (N + 3) // 4 * 4 if N % 4 else N
        res = torch.randn(M, ldy + (4 if N == 75 else 0), device=dev)[:, :N]
        y = torch.empty(M, ldy + (4 if N == 75 else 0), device=dev)[:, :N]
        cs, ct = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
        for name, wide, kw in [("16x16 plain", False, {}), ("32x32 plain", True, {}),
                               ("16x16 tail", False, dict(relu=True, residual=res, col_scale=cs, col_shift=ct)),
                               ("32x32 tail", True, dict(relu=True, residual=res, col_scale=cs, col_shift=ct))]:
            fn = lambda: run(a, K, W, scales, b, wide, out=y, **kw)  # noqa: E731
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
            print(f"M={M} N={N} {name}: {ms:.3f} ms  ({2 * M * K * N * S / ms / 1e9:.1f} TF/s fp32-equivalent)", flush=True)
