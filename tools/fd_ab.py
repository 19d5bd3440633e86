#!/usr/bin/env python
"""A/B timing of pna_fused_degree_f32 builds on ONE box (boxes of the pool differ by 4-5 %): every library named in FD_LIBS
(comma-separated paths under the repo; tools/build_variant.sh) times the C3 group-rows kernel in its own process, round robin.

    FD_LIBS=pna_amd/lib/libpna_amd_r3.so,pna_amd/lib/libpna_amd.so python tools/fd_ab.py [rounds]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, ROOT)
    from pna_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["PNA_AMD_LIB"])
    _lib.PNA_ABI_VERSION = None
    import ctypes
    L = ctypes.CDLL(_lib.LIB_PATH)
    _lib.PNA_ABI_VERSION = L.pna_abi_version()          # (a variant may carry an older ABI number with the same structs)
    from pna_amd import Graph, degree_groups as DG, functional as PF
    from pna_amd.dgl.pna_layer import PNASimpleLayer
    from pna_amd.synth import powerlaw_graph
    dev = torch.device("cuda:0")
    V, E, F = 1_000_000, 10_000_000, 75
    src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
    g = Graph(src, dst, V)
    avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
    torch.manual_seed(0)
    layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True).to(dev).eval()
    h = torch.randn(V, 80, device=dev)[:, :F]

    def ev(fn, n=20, reps=5):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / n)
        return best
    with torch.no_grad():
        call = PF.FusedDegreeCall(layer, g, h, x=h)
        call.set_spare(False)
        out = {"group_rows_ms": ev(call.group_rows)}
        call.set_spare(True)
        out["group_rows_spare32_ms"] = ev(call.group_rows)
        try:
            hc = torch.cat([h.contiguous().reshape(-1), torch.zeros(8, device=dev)])[:V * F].view(V, F)
            c2 = PF.FusedDegreeCall(layer, g, hc, x=hc)
            c2.set_spare(False)
            out["contiguous_ms"] = ev(c2.group_rows)
        except Exception as ex:    # noqa: BLE001
            out["contiguous_ms"] = None
    print("RESULT " + json.dumps(out))
    sys.exit(0)

libs = os.environ["FD_LIBS"].split(",")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, PNA_AMD_LIB=l)
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=300)
        line = [x for x in o.stdout.splitlines() if x.startswith("RESULT ")]
        if line:
            res[l].append(json.loads(line[0][7:]))
            print(r, os.path.basename(l), line[0][7:], flush=True)
        else:
            print(r, os.path.basename(l), "FAILED", o.stderr[-400:], flush=True)
for l in libs:
    if res[l]:
        print(f"{os.path.basename(l):28s} group rows best {min(x['group_rows_ms'] for x in res[l]):.4f} ms, 32 workgroups left out {min(x['group_rows_spare32_ms'] for x in res[l]):.4f},"
              f" contiguous {min((x['contiguous_ms'] or 9) for x in res[l]):.4f}")
