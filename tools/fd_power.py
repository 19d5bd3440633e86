#!/usr/bin/env python
"""Is the one-kernel layer power-bound?  Socket power and shader clock (rocm-smi, sampled every 0.25 s) while ONE kernel variant runs
in a loop on the C3 graph, and the shader clock the kernel itself sees (s_memtime ticks of a wavefront over the launch / wall time).
With the experiments build (tools/build_experiments.sh; PNA_AMD_LIB=pna_amd/lib/libpna_amd_exp.so) the ablation knobs of
pna_fused_degree.hip (PNA_FD_ABL) are swept too: what each part of the kernel costs in time AND in watts.

    python tools/fd_power.py [json-out]
"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import _lib  # noqa: E402
if os.environ.get("PNA_AMD_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["PNA_AMD_LIB"])
from pna_amd import Graph, degree_groups as DG, functional as PF  # noqa: E402
from pna_amd.dgl.pna_layer import PNASimpleLayer  # noqa: E402
from pna_amd.synth import powerlaw_graph  # noqa: E402

dev = torch.device("cuda:0")
V, E, F = 1_000_000, 10_000_000, 75
src, dst = powerlaw_graph(V, E, seed=1234, device=dev)
g = Graph(src, dst, V)
avg = {"log": torch.log(g.in_degrees().double() + 1).mean().float().cpu()}
torch.manual_seed(0)
layer = PNASimpleLayer(F, F, "mean max min std", "identity amplification attenuation", avg, 0.0, True, True)
with torch.no_grad():
    for p in layer.parameters():
        p.copy_(torch.randn_like(p) / (p.shape[-1] ** 0.5 if p.dim() == 2 else 3.0))
layer = layer.to(dev).eval()
h = torch.randn(V, 80, device=dev)[:, :F]
SECONDS = float(os.environ.get("FD_POWER_SECONDS", "2.5"))


def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
        return o.strip().splitlines()
    except Exception as e:    # noqa: BLE001
        return [repr(e)]


def probe(name, fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], [False]

    def poll():
        while not stop[0]:
            samples.append(smi())
            time.sleep(0.25)
    th = threading.Thread(target=poll)
    th.start()
    t0, n = time.perf_counter(), 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    while time.perf_counter() - t0 < SECONDS:
        a.record()
        for _ in range(50):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 50)
        n += 50
    dt = time.perf_counter() - t0
    stop[0] = True
    th.join()
    rows = [s[-1] for s in samples[2:] if len(s) >= 2]
    hdr = samples[-1][0] if samples and len(samples[-1]) >= 2 else ""
    out = {"ms_wall": dt / max(n, 1) * 1e3, "ms_best_of_50": best, "smi_header": hdr, "smi": rows[:6]}
    print(f"{name}: {out['ms_wall']:.4f} ms/launch (best 50-launch mean {best:.4f})", flush=True)
    for r in rows[:4]:
        print("    ", r[:260], flush=True)
    return out


res = {"idle_smi": smi()}
print("idle:", res["idle_smi"], flush=True)
with torch.no_grad():
    call = PF.FusedDegreeCall(layer, g, h)
    call.args.spare_workgroups = 0
    res["fused_group_rows"] = probe("one-kernel layer, group rows, whole device", call.group_rows)
    call.args.spare_workgroups = 32
    res["fused_group_rows_spare32"] = probe("one-kernel layer, group rows, 32 workgroups left out", call.group_rows)
    call.args.spare_workgroups = 0
    res["gather_only"] = probe("standalone gather (pna_segreduce_fwd_f32, 4F aggregate written)",
                               lambda: PF.aggregate(g, h, F, ["mean", "max", "min", "std"]))
    if "exp" in os.path.basename(_lib.LIB_PATH):
        props = torch.cuda.get_device_properties(0)
        nw = props.multi_processor_count * 2 * 4
        for abl, what in [(0, "nothing skipped"), (1, "no MFMAs"), (3, "no MFMAs, no fragment maths"), (4, "no fold"), (8, "no y stores"),
                          (19, "no MFMA / fragment maths / B reads"), (31, "skeleton: loads, waits, barriers, weight copies only"), (0, "nothing skipped (again)")]:
            os.environ["PNA_FD_ABL"] = str(abl)
            r = probe(f"ablation {abl} ({what})", call.group_rows)
            dbg = torch.zeros(nw * 4, dtype=torch.int64, device=dev)
            os.environ["PNA_FD_DBG_PTR"] = hex(dbg.data_ptr())
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            call.group_rows()
            b.record()
            torch.cuda.synchronize()
            del os.environ["PNA_FD_DBG_PTR"]
            d = dbg.view(nw, 4).double()
            d = d[d[:, 3] > 0]
            r["one_launch_ms"] = a.elapsed_time(b)
            r["wave_ticks_max"] = d[:, 3].max().item()
            r["wave_ticks_mean"] = d[:, 3].mean().item()
            r["ticks_per_us_of_the_longest_wave"] = r["wave_ticks_max"] / (r["one_launch_ms"] * 1e3)
            print(f"     s_memtime: longest wavefront {r['wave_ticks_max']:.0f} ticks over {r['one_launch_ms']:.4f} ms = {r['ticks_per_us_of_the_longest_wave']:.0f} ticks/us", flush=True)
            res[f"ablation_{abl}" + ("_again" if f"ablation_{abl}" in res else "")] = r
        del os.environ["PNA_FD_ABL"]
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
