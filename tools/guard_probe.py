import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_fused_degree as T
dev = torch.device("cuda:0")
for F, N in [(75, 75), (128, 128), (40, 72), (64, 96)]:
    for ratio, wb in [(1.0, 1.0), (1e-6, 1.0), (1e4, 0.0), (1e7, 0.0), (1e7, 1e-8), (1e12, 0.0), (1e12, 1e-8)]:
        out = {}
        for ar in ("guarded", "bf16x3", "fp16x2"):
            w, h, t = T._adversarial_case(dev, 140_000, 1_100_000, F, N, ratio, wb, ar)
            out[ar] = (round(w, 4), h, t)
        print(F, N, ratio, wb, out, flush=True)
