#!/usr/bin/env python
"""Random shapes through pna_project_f32 / pna_project_scaled_f32 / pna_project_grouped_f32 against float64 (1e-6 of sum |x||w|, the bar of
tests/test_gpu_project.py): row counts around the 128-row workgroup tile, every K in 4..128, ragged N, pitched inputs and outputs, -1 padding rows,
groups in any order.   python tools/fuzz_project.py [seconds] [seed]"""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pna_amd import ops  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
g = torch.Generator().manual_seed(rnd.randrange(1 << 30))
dev = torch.device("cuda:0")
t0, n, worst = time.time(), 0, 0.0


def err(y, ref, floor):
    return ((y.cpu().double() - ref).abs() / floor.clamp(min=1e-300)).max().item() if ref.numel() else 0.0


while time.time() - t0 < secs:
    kind = rnd.choice(["plain", "scaled", "grouped"])
    M = rnd.choice([1, 15, 16, 17, 127, 128, 129, 255, 1000, rnd.randrange(1, 20000)])
    K = rnd.randrange(4, 129)
    ldx = K + rnd.choice([0, 0, 1, 3, 5])
    xb = torch.randn(M, ldx, generator=g) * torch.exp(torch.randn(M, 1, generator=g))
    x = xb.to(dev)[:, :K]
    if kind == "plain":
        N = rnd.randrange(1, 513)
        if not ops.project_applies(x, K, N):
            continue
        w = torch.randn(N, K, generator=g) * 0.3
        ldy = N + rnd.choice([0, 0, 2, 7])
        yb = torch.full((M, ldy), 7.0, device=dev)
        y = ops.project(x, K, w.to(dev), out=yb[:, :N])
        ref, floor = xb[:, :K].double() @ w.double().t(), xb[:, :K].abs().double() @ w.abs().double().t()
        e = err(y, ref, floor)
        ok = e <= 1e-6 and bool((yb[:, N:] == 7.0).all())
    elif kind == "scaled":
        N, S, self_block = rnd.randrange(1, 81), rnd.randrange(0, 4), rnd.random() < 0.6
        B = S + int(self_block)
        if B == 0 or not ops.project_scaled_applies(x, K, N, B):
            continue
        w = torch.randn(N, B * K, generator=g) * 0.3
        sc = torch.rand(M, S, generator=g) * 3 if S else None
        beta = torch.randn(S, N, generator=g) if (S and rnd.random() < 0.7) else None
        ldy = N + rnd.choice([0, 0, 5])
        yb = torch.full((M, ldy), 7.0, device=dev)
        y = ops.project_scaled(x, K, w.to(dev), None if sc is None else sc.to(dev), None if beta is None else beta.to(dev), self_block, out=yb[:, :N])
        xd, wd = xb[:, :K].double(), w.double()
        ref = torch.zeros(M, N, dtype=torch.float64); floor = torch.zeros(M, N, dtype=torch.float64)
        b0 = 0
        if self_block:
            ref += xd @ wd[:, :K].t(); floor += xd.abs() @ wd[:, :K].abs().t(); b0 = 1
        for s in range(S):
            blk = wd[:, (b0 + s) * K:(b0 + s + 1) * K]
            bs = beta[s].double() if beta is not None else torch.zeros(N, dtype=torch.float64)
            ref += sc[:, s:s + 1].double() * (xd @ blk.t() + bs)
            floor += sc[:, s:s + 1].double() * (xd.abs() @ blk.abs().t() + bs.abs())
        e = err(y, ref, floor)
        ok = e <= 1e-6 and bool((yb[:, N:] == 7.0).all())
    else:
        N, G = rnd.randrange(1, 400), rnd.randrange(1, 20)
        if not ops.project_applies(x, K, N):
            continue
        nt = (M + 127) // 128 + rnd.randrange(0, 3)
        tg = torch.randint(0, G, (nt,), generator=g, dtype=torch.int32)
        if rnd.random() < 0.5:
            tg = torch.sort(tg).values
        perm = torch.full((nt * 128,), -1, dtype=torch.int32)
        named = torch.randperm(M, generator=g)[:max(1, M - M // 7)]
        slots = torch.randperm(nt * 128, generator=g)[:named.numel()]
        perm[slots] = named.to(torch.int32)
        w = torch.randn(G, N, K, generator=g) * 0.3
        ldy = N + rnd.choice([0, 3])
        yb = torch.full((M, ldy), 7.0, device=dev)
        ops.project_grouped(x, K, w.to(dev), perm.to(dev), tg.to(dev), out=yb[:, :N])
        grp = tg[(slots // 128)].long()
        ref = torch.einsum("vk,vnk->vn", xb[named, :K].double(), w[grp].double())
        floor = torch.einsum("vk,vnk->vn", xb[named, :K].abs().double(), w[grp].abs().double())
        e = err(yb[named.to(dev)][:, :N], ref, floor)
        untouched = torch.ones(M, dtype=torch.bool); untouched[named] = False
        ok = e <= 1e-6 and bool((yb.cpu()[untouched] == 7.0).all()) and bool((yb[:, N:] == 7.0).all())
    n += 1
    worst = max(worst, e)
    if not ok:
        print(f"FAIL {kind} M={M} K={K} ldx={ldx} N={N} err={e:.2e}")
        sys.exit(1)
print(f"SUMMARY {n} cases passed, worst error {worst:.2e} of sum |x||w|, {secs:.0f} s")
