import os, sys, time, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/pna_amd") else os.environ["GRAFT_REPO_ROOT"])
from pna_amd import _lib
if os.environ.get("PNA_AMD_LIB"):
    _lib.LIB_PATH = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), os.environ["PNA_AMD_LIB"])
from pna_amd import ops
dev = torch.device("cuda:0")
M, K, N = 1_000_000, 300, 75
a = torch.randn(M, K, device=dev); W = torch.randn(N, 3 * K, device=dev) / 30; b = torch.randn(N, device=dev)
sc = [None, torch.rand(M, device=dev) + 0.5, torch.rand(M, device=dev) + 0.5]
res = torch.randn(M, 80, device=dev)[:, :N]; y = torch.empty(M, 80, device=dev)[:, :N]
cs, ct = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
for tail in (False, True):
    kw = dict(col_scale=cs, col_shift=ct, relu=True, residual=res) if tail else {}
    fn = lambda: ops.posttrans(a, K, W, sc, b, arith="bf16x3", out=y, **kw)
    for _ in range(3): fn()
    ms = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); ms = min(ms, (time.perf_counter() - t) / 10 * 1e3)
    print(sys.argv[1], "tail" if tail else "plain", f"{ms:.3f} ms", flush=True)
