#!/usr/bin/env python
"""bf16x3 contraction vs the pitch of the aggregate rows (development tool): 1200-byte rows straddle 128-byte lines for 7 rows
in 8 -- does a 1280-byte pitch (line-aligned 128-byte chunk strips) read faster?

    python tools/x3_pitch_time.py
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, F, N = 1_000_000 // 192 * 192, 75, 75
K = 4 * F
for S in (1, 3):
    W = torch.randn(N, S * K, device=dev) / 30
    b = torch.randn(N, device=dev)
    scales = [None] + [torch.rand(M, device=dev) for _ in range(S - 1)]
    y = torch.empty(M, 80, device=dev)[:, :N]
    res = torch.randn(M, 80, device=dev)[:, :N]
    for lda in (300, 304, 320, 352):
        a = torch.randn(M, lda, device=dev)[:, :K]
        fn = lambda: ops.posttrans(a, K, W, scales, b, arith="bf16x3", out=y, relu=True, residual=res)  # noqa: E731
        for _ in range(3):
            fn()
        ms = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            ms = min(ms, (time.perf_counter() - t) / 20 * 1e3)
        print(f"S={S} lda={lda}: {ms:.3f} ms", flush=True)
