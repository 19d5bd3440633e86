"""GPU parity of the PyG front end against golden vectors produced by the reference's own
models/pytorch_geometric/pna.py (oracle/make_golden_pyg.py; graphs include nodes without in-edges).
Tolerances as in test_gpu_layers.py."""
import pytest
import torch

from conftest import golden_names, load_golden
from pna_amd.pytorch_geometric import AGGREGATORS, SCALERS, PNAConv, PNAConvSimple

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", golden_names("pyg_simple"))
def test_simple_conv_golden(cuda_device, name):
    meta, a, sd = load_golden(name)
    layer = PNAConvSimple(meta["F"], meta["out"], meta["aggregators"], meta["scalers"], a["deg_hist"], post_layers=meta["post_layers"])
    layer.load_state_dict(sd)
    layer = layer.to(cuda_device).eval()
    x, ei = a["x"].to(cuda_device), a["edge_index"].to(cuda_device)
    with torch.no_grad():
        out = layer(x, ei).cpu()
        agg = layer.aggregate(x, ei).cpu()
    # the scaled (V, S*A*F) tensor: max/min under the identity scaler bit-exact, the rest to tolerance
    A, S, F = len(meta["aggregators"]), len(meta["scalers"]), meta["F"]
    got, want = agg.view(-1, S, A, F), a["agg"].view(-1, S, A, F)
    for s, sn in enumerate(meta["scalers"]):
        for k, an in enumerate(meta["aggregators"]):
            if sn == "identity" and an in ("max", "min"):
                assert torch.equal(got[:, s, k], want[:, s, k]), (sn, an)
            else:
                torch.testing.assert_close(got[:, s, k], want[:, s, k], rtol=1e-5, atol=2e-5, msg=lambda m: f"{sn}/{an}: {m}")
    torch.testing.assert_close(out, a["out"], **TOL)


@pytest.mark.parametrize("name", golden_names("pyg_conv"))
def test_conv_golden(cuda_device, name):
    meta, a, sd = load_golden(name)
    layer = PNAConv(meta["in_c"], meta["out_c"], meta["aggregators"], meta["scalers"], a["deg_hist"],
                    edge_dim=meta["edge_dim"] or None, towers=meta["towers"], pre_layers=meta["pre_layers"],
                    post_layers=meta["post_layers"], divide_input=meta["divide_input"])
    layer.load_state_dict(sd)
    layer = layer.to(cuda_device).eval()
    ea = a["edge_attr"].to(cuda_device) if meta["edge_dim"] else None
    with torch.no_grad():
        out = layer(a["x"].to(cuda_device), a["edge_index"].to(cuda_device), ea).cpu()
    torch.testing.assert_close(out, a["out"], **TOL)


def test_registry_functions_on_the_gpu(cuda_device):
    """The registry entries called directly, against their definitions (torch_scatter semantics: empty segments -> 0,
    std of an empty segment -> sqrt(1e-5), var not clamped)."""
    gen = torch.Generator().manual_seed(5)
    src = torch.randn(50, 3, 4, generator=gen).to(cuda_device)
    index = torch.randint(0, 8, (50,), generator=gen).to(cuda_device)      # segments 8, 9 stay empty
    n = 10
    one = torch.zeros(n, device=cuda_device).index_add_(0, index, torch.ones(50, device=cuda_device))
    s = torch.zeros(n, 3, 4, device=cuda_device).index_add_(0, index, src)
    q = torch.zeros(n, 3, 4, device=cuda_device).index_add_(0, index, src * src)
    cnt = one.clamp(min=1).view(-1, 1, 1)
    mean = s / cnt
    var = q / cnt - mean * mean
    torch.testing.assert_close(AGGREGATORS["sum"](src, index, n), s, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(AGGREGATORS["mean"](src, index, n), mean, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(AGGREGATORS["var"](src, index, n), var, rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(AGGREGATORS["std"](src, index, n), torch.sqrt(torch.relu(var) + 1e-5), rtol=1e-5, atol=2e-5)
    mx = AGGREGATORS["max"](src, index, n)
    assert torch.equal(mx[8:], torch.zeros(2, 3, 4, device=cuda_device))
    for v in range(8):
        assert torch.equal(mx[v], src[index == v].max(dim=0).values)
    deg = one.view(-1, 1, 1)
    att = SCALERS["attenuation"](torch.ones(n, 1, 1, device=cuda_device), deg, {"log": 1.5, "lin": 3.0})
    assert att[8].item() == 1.0 and att[0].item() == pytest.approx(1.5 / torch.log(deg[0] + 1).item(), rel=1e-6)
