#!/usr/bin/env python
"""pna_project_f32 against the library GEMM at the multi-tower layer's shape (1 M rows, K = 75, 5 x 80 columns)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pna_amd import ops

M, K, N = int(os.environ.get("M", 1000000)), 75, 400
x, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
wt = w.t().contiguous()
y = torch.empty(M, N, device="cuda")

def t(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

tm = t(lambda: torch.mm(x, wt, out=y))
tp = t(lambda: ops.project(x, K, w, out=y))
gb = (M * K + M * N) * 4 / 1e9
print(f"torch.mm {tm:.4f} ms   pna_project_f32 {tp:.4f} ms  ({gb / tp * 1e3:.0f} GB/s of x + y, {2e-9 * M * 80 * N / tp:.1f} TFLOP/s fp32 MFMA)")
