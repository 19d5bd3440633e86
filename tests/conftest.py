import glob
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def load_golden(name):
    """-> (meta dict, arrays dict of torch tensors, state_dict)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    arrays, sd = {}, {}
    for k in z.files:
        if k == "meta":
            continue
        t = torch.from_numpy(z[k])
        if k.startswith("sd/"):
            sd[k[3:]] = t
        else:
            arrays[k] = t
    return meta, arrays, sd


def golden_names(kind):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        z = np.load(p, allow_pickle=False)
        if json.loads(str(z["meta"]))["kind"] == kind:
            out.append(os.path.splitext(os.path.basename(p))[0])
    return out


@pytest.fixture(scope="session")
def cuda_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
