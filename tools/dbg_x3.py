import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import ops
dev = torch.device("cuda:0")

def poison():
    bufs = [torch.full((64 * 1024 * 1024 // 4,), float("nan"), device=dev) for _ in range(8)]
    del bufs

def case(M, K, N, S, Kh, post, bn, relu, resid, seed):
    gen = torch.Generator().manual_seed(seed)
    a = torch.randn(M, K, generator=gen).to(dev)
    h = torch.randn(M, Kh, generator=gen).to(dev) if Kh else None
    W = (torch.randn(N, Kh + S * K, generator=gen) / 14).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    scales = [None] + [(torch.rand(M, generator=gen) + 0.5).to(dev) for _ in range(S - 1)]
    cs = (torch.rand(N, generator=gen) + 0.5).to(dev) if bn else None
    ct = torch.randn(N, generator=gen).to(dev) if bn else None
    rp = (torch.rand(M, generator=gen) + 0.5).to(dev) if post else None
    res = torch.randn(M, N, generator=gen).to(dev) if resid else None
    ref = ops.posttrans(a, K, W, scales, b, h, row_post=rp, col_scale=cs, col_shift=ct, relu=relu, residual=res, arith="f32")
    for it in range(30):
        if it % 10 == 0:
            poison()
        for pl in (2, 3):
            y = ops.posttrans(a, K, W, scales, b, h, row_post=rp, col_scale=cs, col_shift=ct, relu=relu, residual=res, arith="bf16x3", pipeline=pl)
            bad = (y - ref).abs() > 1e-3 * (1 + ref.abs())
            if bad.any():
                rows = torch.nonzero(bad.any(1)).flatten().tolist()
                cols = torch.nonzero(bad.any(0)).flatten().tolist()
                print(f"MISMATCH M={M} K={K} N={N} S={S} Kh={Kh} it={it} pipeline={pl}: {int(bad.sum())} elems, rows {rows[:12]}..{rows[-3:]} ({len(rows)}), cols {cols[:20]}", flush=True)
                return False
    print(f"ok M={M} K={K} N={N} S={S} Kh={Kh}", flush=True)
    return True

case(50, 64, 16, 3, 0, False, True, True, True, 1)
case(777, 64, 15, 3, 15, True, True, False, False, 2)
case(48, 64, 16, 3, 0, False, True, True, True, 3)
case(200, 64, 16, 3, 0, False, False, False, False, 4)
case(200, 32, 16, 3, 0, False, False, False, False, 5)
case(1000, 300, 75, 3, 0, False, True, True, True, 6)
case(3000, 60, 15, 3, 15, True, True, False, False, 7)
case(100000, 300, 75, 3, 0, False, True, True, True, 8)
