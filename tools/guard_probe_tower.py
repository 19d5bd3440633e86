import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_fused_degree as T
dev = torch.device("cuda:0")
for towers, div, mut in [(5, True, T._towers_apart), (1, False, T._huge_own_feature), (1, False, lambda l, h: None), (5, True, lambda l, h: None)]:
    print(getattr(mut, "__name__", "benign"), {ar: T._tower_case(dev, towers, div, mut, ar) for ar in ("bf16x3", "guarded", "fp16x2")}, flush=True)
