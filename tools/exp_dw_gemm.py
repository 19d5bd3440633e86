#!/usr/bin/env python
"""Weight-gradient product of the posttrans backward at C3 size: (S N x M)(M x K), M = 1e6 -- variants of the library call."""
import torch, sys, os, json
dev = torch.device("cuda:0")
M, N3, K = 1_000_000, 225, 300
G = torch.randn(M, N3, device=dev); a = torch.randn(M, K, device=dev)
ref = None


def ev(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        r = fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n, r


out = {}
t, ref = ev(lambda: G.t() @ a); out["mm"] = t
for C in (16, 64, 256, 1024):
    Mc = M // C * C
    def f(C=C, Mc=Mc):
        r = torch.bmm(G[:Mc].view(C, Mc // C, N3).transpose(1, 2), a[:Mc].view(C, Mc // C, K)).sum(0)
        return r + G[Mc:].t() @ a[Mc:] if Mc < M else r
    t, r = ev(f); out[f"bmm{C}"] = (t, float((r - ref).abs().max() / ref.abs().max()))
try:
    torch.backends.cuda.preferred_blas_library("hipblaslt")
    t, r = ev(lambda: G.t() @ a); out["mm_hipblaslt"] = (t, float((r - ref).abs().max() / ref.abs().max()))
    def f():
        C = 64; Mc = M // C * C
        return torch.bmm(G[:Mc].view(C, Mc // C, N3).transpose(1, 2), a[:Mc].view(C, Mc // C, K)).sum(0)
    t, r = ev(f); out["bmm64_hipblaslt"] = t
except Exception as ex:
    out["hipblaslt"] = repr(ex)
# the transposed formulation (K x M)(M x SN)
t, r = ev(lambda: a.t() @ G); out["mm_T"] = t
print(json.dumps(out, indent=1))
