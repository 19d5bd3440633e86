import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, K, N = 192, 64, 16
W = torch.zeros(N, K, device=dev)
W[:, :32] = 1.0          # chunk 0 weights = 1, chunk 1 weights = 100
W[:, 32:] = 100.0
for name, a in (("chunk0", torch.cat([torch.ones(M, 32), torch.zeros(M, 32)], 1)), ("chunk1", torch.cat([torch.zeros(M, 32), torch.ones(M, 32)], 1)),
                ("both", torch.ones(M, 64)), ("rowid", torch.arange(M).float()[:, None].repeat(1, 64))):
    a = a.to(dev)
    ref = ops.posttrans(a, K, W, [None], arith="f32")
    for pl in (2, 3):
        for it in range(6):
            y = ops.posttrans(a, K, W, [None], arith="bf16x3", pipeline=pl)
            if not torch.allclose(y, ref, rtol=1e-4, atol=1e-3):
                d = (y - ref)
                rows = torch.nonzero((d.abs() > 1e-3).any(1)).flatten().tolist()
                print(name, "pipeline", pl, "it", it, "bad rows", rows[:8], "...", rows[-4:], len(rows), "y[row0]", y[rows[0], :4].tolist(), "ref", ref[rows[0], :4].tolist(), flush=True)
                break
        else:
            print(name, "pipeline", pl, "ok", flush=True)
