"""Phase timers of the molecule-batch tower layer (experiments build, tools/build_experiments.sh): per wavefront, shader
clock at the marks of pna_tower_fused.hip, printed as mean / max over wavefronts of the time since the kernel's first mark."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "pna_amd", "lib", "libpna_amd_exp.so")
from pna_amd import Graph
from pna_amd.dgl.pna_layer import PNALayer
from pna_amd.synth import molecule_batch
dev = torch.device("cuda:0")
src, dst, sizes = molecule_batch(128, seed=41)
V = sum(sizes)
g = Graph(src, dst, V, sizes).to(dev)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
sn = g.snorm_n()
NWG = (V + 15) // 16
for name, (fi, fo, div) in (("zinc_mid", (75, 75, False)), ("zinc_last", (75, 70, True))):
    layer = PNALayer(fi, fo, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True, towers=5, divide_input=div, residual=True).to(dev).eval()
    h = torch.randn(V, fi, device=dev)
    with torch.no_grad():
        for _ in range(5):
            layer(g, h, None, sn)
        torch.cuda.synchronize()
        d1 = torch.zeros(NWG * 8 * 16, dtype=torch.int64, device=dev)
        d2 = torch.zeros(NWG * 8 * 16, dtype=torch.int64, device=dev)
        os.environ["PNA_TF_DBG_LINEAR"] = hex(d1.data_ptr())
        os.environ["PNA_TF_DBG_ROWS"] = hex(d2.data_ptr())
        layer(g, h, None, sn)
        torch.cuda.synchronize()
        del os.environ["PNA_TF_DBG_LINEAR"], os.environ["PNA_TF_DBG_ROWS"]
    for kname, d, marks in (("k_small_linear", d1, ["start", "x tile in LDS", "end"]),
                            ("k_tower_rows", d2, ["start", "tile operands requested", "(group loop top)", "gather done", "sync", "items done", "sync (K parts)",
                                                  "reduce+epilogue", "sync", "mixing + store"])):
        t = d.view(NWG, 8, 16).double()
        wall = (t[:, :, 15].max() - t[:, :, 14].min()).item() * 10.0       # wall_clock64 ticks of 10 ns
        cyc = (t[:, :, len(marks) - 1].max() - t[:, :, 0].min()).item()
        print(f"{name} {kname}: first mark -> last mark {wall:.0f} ns wall, {cyc:.0f} shader cycles ({cyc / max(wall, 1):.2f} GHz); start skew {(t[:, :, 0].max() - t[:, :, 0].min()).item():.0f} cycles")
        for k in range(1, len(marks)):
            dt = t[:, :, k] - t[:, :, 0]
            valid = t[:, :, k] > 0
            dk = (t[:, :, k] - t[:, :, k - 1])[valid]
            print(f"   {marks[k]:24s} +{dk.mean().item():8.0f} cycles (max {dk.max().item():8.0f})   cumulative mean {dt[valid].mean().item():8.0f}  max {dt[valid].max().item():8.0f}")
