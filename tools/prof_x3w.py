#!/usr/bin/env python
"""Driver for rocprofv3 passes over the two bf16x3 contraction kernels (16x16 and 32x32 tiles) on the C3 shape."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
M, F = 1_000_000, 75
K, N = 4 * F, F
a = torch.randn(M, K, device=dev)
W = torch.randn(N, 3 * K, device=dev) / 30
b = torch.randn(N, device=dev)
sc = [None, torch.rand(M, device=dev) + 0.5, torch.rand(M, device=dev) + 0.5]
res = torch.randn(M, 80, device=dev)[:, :N]
y = torch.empty(M, 80, device=dev)[:, :N]
cs, ct = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
for wide in (False, True):  # both kernels, same inputs
    ops.X3_WIDE = wide
    for _ in range(n):
        ops.posttrans(a, K, W, sc, b, arith="bf16x3", out=y, col_scale=cs, col_shift=ct, relu=True, residual=res)
torch.cuda.synchronize()
print("ok")
