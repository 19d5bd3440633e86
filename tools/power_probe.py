#!/usr/bin/env python
"""Socket power and clocks (rocm-smi) while one kernel runs in a loop: is the contraction power-bound?
    python tools/power_probe.py x3|segreduce|idle"""
import os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pna_amd import ops, Graph, functional as PF
from pna_amd.synth import powerlaw_graph
which = sys.argv[1] if len(sys.argv) > 1 else "x3"
dev = torch.device("cuda:0")
M, K, N = 1_000_000, 300, 75
a = torch.randn(M, K, device=dev); W = torch.randn(N, 3 * K, device=dev) / 30; b = torch.randn(N, device=dev)
sc = [None, torch.rand(M, device=dev) + 0.5, torch.rand(M, device=dev) + 0.5]
y = torch.empty(M, 80, device=dev)[:, :N]
if which == "segreduce":
    src, dst = powerlaw_graph(M, 10 * M, seed=1234, device=dev)
    g = Graph(src, dst, M)
    x = torch.randn(M, 80, device=dev)[:, :75]
    fn = lambda: PF.aggregate(g, x, 75, ["mean", "max", "min", "std"])
elif which == "f32":
    fn = lambda: ops.posttrans(a, K, W, sc, b, arith="f32", out=y)
else:
    fn = lambda: ops.posttrans(a, K, W, sc, b, arith="bf16x3", out=y)
samples, stop = [], False
def poll():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(o.strip().splitlines()[-1])
        except Exception as e:
            samples.append(repr(e))
        time.sleep(0.3)
if which != "idle":
    for _ in range(5): fn()
torch.cuda.synchronize()
th = threading.Thread(target=poll); th.start()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 6.0:
    if which == "idle":
        time.sleep(0.05)
    else:
        for _ in range(50): fn()
        torch.cuda.synchronize(); n += 50
dt = time.perf_counter() - t0
stop = True; th.join()
print(which, f"{dt / max(n, 1) * 1e3:.3f} ms per launch over {n} launches")
for s in samples[2:8]:
    print("  ", s[:300])
