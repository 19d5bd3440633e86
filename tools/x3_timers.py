"""Phase timers of the bf16x3 contraction (experiments build): per wavefront, cycles spent (0) from the top of a step to the
mid-chunk wait, (1) in that counted vmcnt wait, (2) in s_barrier, (3) from the barrier to the last MFMA issue, (4) in take()
(anchor + split of the next A fragment), (5) in the epilogue; (6) steps, (7) whole loop."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "pna_amd", "lib", "libpna_amd_exp.so")
from pna_amd import ops
dev = torch.device("cuda:0")
M, K, N = 1_000_000, 300, 75
a = torch.randn(M, K, device=dev)
W = (torch.randn(N, 3 * K, device=dev) / 30)
b = torch.randn(N, device=dev)
sc = [None, torch.rand(M, device=dev) + 0.5, torch.rand(M, device=dev) + 0.5]
res = torch.randn(M, 80, device=dev)[:, :N]
cs, ct = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
dbg = torch.zeros(256 * 12 * 8, dtype=torch.int64, device=dev)
for tail in (False, True):
    kw = dict(col_scale=cs, col_shift=ct, relu=True, residual=res) if tail else {}
    for _ in range(3):
        ops.posttrans(a, K, W, sc, b, arith="bf16x3", pipeline=3, **kw)
    torch.cuda.synchronize()
    os.environ["PNA_X3_DBG_PTR"] = hex(dbg.data_ptr())
    dbg.zero_()
    ops.posttrans(a, K, W, sc, b, arith="bf16x3", pipeline=3, **kw)
    torch.cuda.synchronize()
    del os.environ["PNA_X3_DBG_PTR"]
    d = dbg.view(256, 12, 8).double()
    steps = d[:, :, 6].mean().item()
    names = ["pre-barrier MFMA half", "vmcnt wait", "s_barrier", "post-barrier MFMA half", "take/split", "epilogue(+loop tail)"]
    tot = d[:, :, 7].mean().item()
    print(f"tail={tail}: steps/wave {steps:.1f}, loop cycles/wave {tot:.0f}, per step {tot / steps:.0f}")
    for i, n in enumerate(names):
        v = d[:, :, i].mean().item()
        print(f"   {n:28s} {v / steps:8.0f} cycles/step  {100 * v / tot:5.1f} %   (min wave {d[:, :, i].min().item() / steps:.0f}, max wave {d[:, :, i].max().item() / steps:.0f})")
    # by wave index within the workgroup (SIMD = wave % 4 presumably)
    print("   s_barrier wait by wave:", [round(d[:, w, 2].mean().item() / steps) for w in range(12)])
    print("   pre half by wave     :", [round(d[:, w, 0].mean().item() / steps) for w in range(12)])
    print("   post half by wave    :", [round(d[:, w, 3].mean().item() / steps) for w in range(12)])
