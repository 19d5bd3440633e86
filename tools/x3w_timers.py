"""Raw time stamps of the first 64 steps of every wavefront of k_posttrans_x3w (library built with -DPNA_X3W_TIMERS):
stamp 0 top of step, 1 before the counted vmcnt wait, 2 after it, 3 after s_barrier, 4 last MFMA issued, 5 end of step."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "pna_amd", "lib", "libpna_amd_timers.so")
from pna_amd import ops
ops.X3_WIDE = True
dev = torch.device("cuda:0")
M, K, N = 1_000_000, 300, int(os.environ.get("N", 75))
a = torch.randn(M, K, device=dev)
W = (torch.randn(N, 3 * K, device=dev) / 30)
b = torch.randn(N, device=dev)
sc = [None, torch.rand(M, device=dev) + 0.5, torch.rand(M, device=dev) + 0.5]
res = torch.randn(M, 80, device=dev)[:, :N]
y = torch.empty(M, 80, device=dev)[:, :N]
cs, ct = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
dbg = torch.zeros(256 * 8 * 64 * 8, dtype=torch.int64, device=dev)
kw = dict(col_scale=cs, col_shift=ct, relu=True, residual=res)
fn = lambda: ops.posttrans(a, K, W, sc, b, arith="bf16x3", out=y, **kw)  # noqa: E731
for _ in range(3):
    fn()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    fn()
torch.cuda.synchronize()
print(f"kernel with stamps off: {(time.perf_counter() - t) / 10 * 1e3:.3f} ms")
os.environ["PNA_X3W_DBG_PTR"] = hex(dbg.data_ptr())
fn()
torch.cuda.synchronize()
d = dbg.view(256, 8, 64, 8)[:, :, :, :6].double().cpu()
names = ["top->pre-wait (pre-barrier half)", "counted vmcnt wait", "s_barrier", "barrier->last MFMA issued", "last MFMA->end of step (slow take, copy, epilogue)"]
dur = d[:, :, :, 1:] - d[:, :, :, :-1]                     # [wg][wave][step][5]
nxt = d[:, :, 1:, 0] - d[:, :, :-1, 5]                    # end of step -> top of next
steps = list(range(3, 57))
NS = 19
reg = [j for j in steps if j % NS != NS - 1]
tail = [j for j in steps if j % NS == NS - 1]
print("cycles of the constant-rate counter (100 MHz => x ~20 for shader cycles?) -- compare ratios")
per = (d[:, :, 57, 0] - d[:, :, 3, 0]).mean().item() / 54
print(f"mean step period {per:.1f}")
for i, n in enumerate(names):
    print(f"  {n:52s} regular steps {dur[:, :, reg, i].mean().item():8.1f}   tile-end steps {dur[:, :, tail, i].mean().item():8.1f}")
print(f"  {'end of step -> top of next':52s} {nxt[:, :, 3:56].mean().item():8.1f}")
for w in range(8):
    print(f"  wave {w}: pre {dur[:, w, reg, 0].mean().item():7.1f} wait {dur[:, w, reg, 1].mean().item():7.1f} barrier {dur[:, w, reg, 2].mean().item():7.1f} post {dur[:, w, reg, 3].mean().item():7.1f} end {dur[:, w, reg, 4].mean().item():7.1f} | tile-end: end {dur[:, w, tail, 4].mean().item():7.1f} barrier-next {dur[:, w, [j + 1 for j in tail], 2].mean().item():7.1f}")
te = d.new_tensor(dbg.view(256, 8, 64, 8).double().cpu())
for w in (0, 4):
    print(f"  tile-end step wave {w}: MFMA end -> deposit0 done {(te[:, w, tail, 6] - te[:, w, tail, 4]).mean().item():8.0f}   residual wait {(te[:, w, tail, 7] - te[:, w, tail, 6]).mean().item():8.0f}   rest (readback0, deposit1, wait, readback1) {(te[:, w, tail, 5] - te[:, w, tail, 7]).mean().item():8.0f}")
# one workgroup's timeline, steps 17..21
wg = 5
t0 = d[wg, :, 17, 0].min().item()
for j in range(17, 22):
    for w in range(8):
        print(f"   wg{wg} step {j} wave {w}: " + " ".join(f"{d[wg, w, j, i].item() - t0:8.0f}" for i in range(6)))
