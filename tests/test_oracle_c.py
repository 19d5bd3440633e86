"""The C restatement (oracle/pna_oracle.c) against the reference-generated goldens: max/min bit-exact,
mean/sum/std/var to fp32 summation-order tolerance, degree scalers bit-exact."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from oracle import c_oracle
from oracle import torch_oracle as O


def _csr(src, dst, N):
    rowptr, order, deg = O.csr_by_dst(src, dst, N)
    return rowptr.numpy().astype(np.int32), torch.as_tensor(src)[order].numpy().astype(np.int32)


@pytest.mark.parametrize("name", golden_names("dgl_simple"))
def test_c_oracle_matches_reference_aggregate(name):
    meta, a, _ = load_golden(name)
    aggs, scalers = meta["aggregators"].split(), meta["scalers"].split()
    rowptr, col = _csr(a["src"], a["dst"], meta["N"])
    amp, att = c_oracle.degree_scalers(rowptr, float(a["avg_log"]))
    scales = [{"identity": None, "amplification": amp, "attenuation": att}[s] for s in scalers]
    got = c_oracle.segreduce(rowptr, col, a["h"].numpy(), meta["F"], aggs, scales)
    ref = a["agg"].numpy()
    F, A = meta["F"], len(aggs)
    for s in range(len(scalers)):
        for i, ag in enumerate(aggs):
            blk = slice((s * A + i) * F, (s * A + i + 1) * F)
            if ag in ("max", "min"):
                # identity-scaled blocks are bit-exact; scaled ones too (same fp32 factor, one multiply)
                assert np.array_equal(got[:, blk], ref[:, blk]), (ag, scalers[s])
            else:
                np.testing.assert_allclose(got[:, blk], ref[:, blk], rtol=1e-5, atol=2e-6, err_msg=f"{ag} {scalers[s]}")


def test_c_oracle_degree_scalers_bit_exact_for_every_degree():
    avg = torch.tensor(1.5207983)
    degs = np.arange(0, 5000)
    rowptr = np.concatenate([[0], np.cumsum(degs)]).astype(np.int32)
    amp, att = c_oracle.degree_scalers(rowptr, float(avg))
    for D in list(range(1, 300)) + [999, 2048, 4999]:
        assert amp[D] == (np.log(D + 1) / avg).item()          # models/dgl/scalers.py:14
        assert att[D] == (avg / np.log(D + 1)).item()          # models/dgl/scalers.py:19
    assert amp[0] == 0 and att[0] == 0


def test_c_oracle_double_accumulation_close_to_fp32():
    rng = np.random.default_rng(0)
    V, E, F = 300, 4000, 7
    dst = np.sort(rng.integers(0, V, E))
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=V))]).astype(np.int32)
    col = rng.integers(0, V, E).astype(np.int32)
    x = rng.standard_normal((V, F)).astype(np.float32)
    a32 = c_oracle.segreduce(rowptr, col, x, F, ["mean", "std", "max", "sum"])
    a64 = c_oracle.segreduce(rowptr, col, x, F, ["mean", "std", "max", "sum"], acc_double=True)
    np.testing.assert_allclose(a32, a64, rtol=1e-5, atol=1e-5)
    assert np.array_equal(a32[:, 2 * F:3 * F], a64[:, 2 * F:3 * F])
