"""bench.py's N > 1 path end to end on ONE GPU: two ranks over gloo on cuda:0 (PNA_BENCH_ONE_DEVICE=1
PNA_BENCH_BACKEND=gloo: shard by destination range, halo all-to-all into the resident table, max-over-ranks timing), launched the
way the driver launches it.  Timings are meaningless here; the CONTRACT is checked: rank 0 prints one parsable JSON line with the
fields the scaling run is judged on -- value over both ranks' edges, scaling "weak", the halo exchange priced in GB/s, the
partition, every rank's sampled parity verdict."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload,port", [("c3", "29547"), ("c5", "29548")])
def test_bench_two_ranks_on_one_gpu_prints_the_contract_line(cuda_device, workload, port):
    """(c5: BASELINE configs[4]'s layer, 128 -> 128, the wide instantiation of the one-kernel layer over a shard's [local | halo] table --
    VERDICT r5 item 8.)"""
    env = dict(os.environ, PNA_BENCH_ONE_DEVICE="1", PNA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    V, E = 150_000, 1_500_000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--nodes-per-gpu", str(V), "--edges-per-gpu", str(E),
           "--no-cpu-baseline", "--kernel-iters", "2", "--workload", workload]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-500:], out.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "edges/s"
    assert abs(d["value"] - 2 * E / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]          # whole-job edges over the max-over-ranks time
    assert d["config"]["partition_balance"] == "nodes" and d["config"]["halo_rows_rank0"] > 0 and "halo_exchange_in_step" in d["config"]
    hx = d["halo_exchange"]
    assert hx["recv_bytes_rank0"] > 0 and hx["recv_GB_per_s_rank0"] > 0 and hx["recv_GB_per_s_per_peer_link"] > 0
    # the link arithmetic a hardware run is judged against rides in the line
    assert hx["halo_bytes_per_rank"] == hx["recv_bytes_rank0"] and hx["exchange_ms_at_link_peak"] > 0
    sp = hx["max_speedup_by_link_bound"]
    assert 0 < sp["exchange_exposed"] <= sp["exchange_hidden"] <= 2.0 + 1e-9
    assert f"F={75 if workload == 'c3' else 128}" in d["metric"]
    ranks = d["parity_check_all_ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and all(r["ok"] for r in ranks), ranks
    assert d["parity_check"]["ok"] and d["diagnostics_error"] is None
    assert d["roofline"]["frac"] is not None and d["roofline"]["bound"] == "hbm"
