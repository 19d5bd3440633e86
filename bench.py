#!/usr/bin/env python
"""bench.py -- PNA-layer forward edges/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], "roofline config"): synthetic power-law graph,
|V| = 1M and |E| = 10M directed edges PER GPU, F = 75, fp32; one step = one PNASimpleLayer forward
(models/dgl/pna_layer.py:197-216: gather + mean/max/min/std + identity/amplification/attenuation +
posttrans Linear(12F->F) + BatchNorm(eval) + ReLU + residual), eval mode, inputs resident in HBM.
With N > 1 the graph has N x the nodes/edges (weak scaling), is sharded by destination range, and each
step includes the RCCL halo all-to-all of source rows.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the step's dominant kernel vs the 8 TB/s HBM roofline on ALGORITHMIC bytes / its measured mean
                  duration (HIP events): the one-kernel layer pna_fused_degree_f32 (gather + aggregators + scalers + posttrans; the
                  4F aggregate is never written: E_g (4F+4) + V_g (8F+4) bytes) where it runs, else the standalone segment-reduce
                  (E (4F+4) + 4 (V+1) + V 16F); `read_only_frac` = the gather's reads alone (the north star's 60 % target)
  cpu_baseline -- the oracle's C port (oracle/pna_oracle.c, OpenMP) + torch CPU Linear/BN of the same
                  layer, timed on this box's host cores on a bounded sample (rank 0, N=1 only)
"""
import argparse
import gc
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

V_PER_GPU = 1_000_000
E_PER_GPU = 10_000_000
F = 75
AGGREGATORS = "mean max min std"
SCALERS = "identity amplification attenuation"
HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md
MFMA_F32_PEAK = 157.3e12   # FLOP/s, f32-input MFMA (= the fp32 vector rate)
ARITH_TEXT = {
    "fp16x2_guarded": "fp16x2_guarded: fp32 in / out; every operand as two fp16 terms behind a power-of-two row / column scale, three partial products "
                      "per multiply, fp32 accumulation -- with the floor-error GUARD: tiles whose outputs the bound does not certify are computed again "
                      "in bf16x3 (componentwise fp32-accurate on every input; include/pna_amd.h ARITHMETIC, DESIGN.md 4.8.17)",
    "fp16x2": "fp16x2 (unguarded, opt-in PNA_AMD_FUSED_ARITH=fp16x2): two fp16 terms, three partial products; normwise- but not componentwise-accurate "
              "when a row's statistics span more than ~2^17",
    "bf16x3": "bf16x3: fp32 in / out; each fp32 operand cut exactly into 3 bf16 terms, 6 partial products per multiply on the bf16 MFMA pipe, fp32 "
              "accumulate; error vs float64 at the exact-f32 kernel's level (tests/test_gpu_posttrans_x3.py)",
}
MFMA_BF16_PEAK = 2.5e15    # FLOP/s, dense bf16 MFMA; the bf16x3 contraction spends 6 bf16 products per fp32 multiply


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--nodes-per-gpu", type=int, default=V_PER_GPU)
    p.add_argument("--edges-per-gpu", type=int, default=E_PER_GPU)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample-rows", type=int, default=250_000)
    p.add_argument("--x-pitch", type=int, default=80, help="row pitch (floats) of the resident feature matrix")
    p.add_argument("--kernel-iters", type=int, default=20, help="launches used for the per-kernel HIP-event timing")
    p.add_argument("--workload", choices=("c3", "c5"), default="c3",
                   help="c3 = BASELINE.json configs[2] (1M nodes / 10M edges per GPU, F=75: the configuration the metric is quoted on); "
                        "c5 = configs[4] (2M nodes / 20M edges per GPU, F=128: V=16M E=160M over 8 GPUs)")
    p.add_argument("--balance", choices=("nodes", "edges"), default="nodes", help="N > 1: destination ranges of equal node or in-edge counts")
    p.add_argument("--no-power-probe", action="store_true", help="skip the rocm-smi power / clock samples (2 x ~1.5 s)")
    p.add_argument("--no-cold", action="store_true", help="skip the cold-cache leg (3 rotating copies of the inputs)")
    p.add_argument("--check-rows", type=int, default=96, help="rows re-computed on the host after the timed loop")
    p.add_argument("--no-c5-leg", action="store_true", help="N = 1, default workload: skip the extra leg that runs BASELINE configs[4]'s per-GPU "
                                                            "shape (--workload c5) in a child process and adds its line as `configs4_per_gpu_shape`")
    p.add_argument("--no-prewarm", action="store_true", help="skip the ~150 ms of untimed steps in front of the W warm-up steps")
    p.add_argument("--layers", type=int, default=1, help="> 1: a step = this many stacked layers (N > 1: inter-layer halo exchange cut into row "
                   "blocks, pna_amd.shard.BlockPipeline); a reduced JSON line, the default line describes ONE layer")
    p.add_argument("--blocks", type=int, default=4, help="--layers > 1, N > 1: row blocks per layer of the pipelined exchange")
    return p.parse_args()


def power_probe(fn, seconds=1.5):
    """{"socket_w", "sclk_mhz"}: medians of rocm-smi samples taken while `fn` is launched back to back for `seconds`; None
    when rocm-smi is missing or its output is not understood (reported, never fatal)."""
    import shutil
    import subprocess
    import threading
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    rows, stop = [], [False]

    def poll():
        while not stop[0]:
            try:
                out = subprocess.run([exe, "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
                lines = [ln for ln in out.strip().splitlines() if ln.startswith(("device", "card"))]
                if len(lines) >= 2:
                    rows.append(dict(zip(lines[0].split(","), lines[1].split(","))))
            except Exception:                                         # noqa: BLE001
                pass
            time.sleep(0.2)

    th = threading.Thread(target=poll, daemon=True)
    try:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        th.start()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
    finally:
        stop[0] = True
        if th.is_alive():
            th.join(timeout=6)
    watts, mhz = [], []
    for r in rows[1:] or rows:                                         # (the first sample may predate the load)
        for k, v in r.items():
            try:
                if "Power (W)" in k:
                    watts.append(float(v))
                elif k.startswith("sclk clock speed"):
                    mhz.append(float(v.strip("()").lower().replace("mhz", "")))
            except ValueError:
                pass
    if not watts or not mhz:
        return None
    med = lambda xs: sorted(xs)[len(xs) // 2]   # noqa: E731
    return {"socket_w": med(watts), "sclk_mhz": med(mhz), "samples": len(watts)}


def event_time_ms(fn, iters, warmup=3):
    """Mean duration of fn() over `iters` back-to-back launches, HIP events on torch's current stream
    (the stream every pna_amd kernel is launched on)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def cpu_baseline(src, dst, V, h, layer_sd, avg_log, sample_rows):
    """Reference-formulation layer forward on the host: C port of reduce_func (materialises the (V,12F)
    aggregate like the reference) + torch CPU Linear / BatchNorm / ReLU / residual, on rows [0, sample_rows)."""
    import numpy as np
    from oracle import c_oracle
    from pna_amd.graph import build_csr
    torch.set_num_threads(os.cpu_count())
    csr = build_csr(src.cpu(), dst.cpu(), V)
    n = min(sample_rows, V)
    rp = csr.rowptr[:n + 1].numpy().copy()
    e_n = int(rp[-1])
    col = csr.col[:e_n].numpy().copy()
    x = h.cpu().numpy()
    amp, att = c_oracle.degree_scalers(rp, float(avg_log))
    sd = {k: v.cpu() for k, v in layer_sd.items()}
    W, b = sd["posttrans.fully_connected.0.linear.weight"], sd["posttrans.fully_connected.0.linear.bias"]
    out = np.empty((n, 12 * F), np.float32)
    out[:] = 0                                                   # fault the pages in outside the clock
    ncpu = os.cpu_count()
    best = None
    for threads in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4)}, reverse=True):   # SMT / NUMA: take the fastest
        c_oracle.set_threads(threads)
        torch.set_num_threads(threads)
        for _ in range(2):
            t0 = time.perf_counter()
            c_oracle.segreduce(rp, col, x, F, AGGREGATORS.split(), [None, amp, att], out=out)
            y = torch.nn.functional.linear(torch.from_numpy(out), W, b)
            y = torch.nn.functional.batch_norm(y, sd["batchnorm_h.running_mean"], sd["batchnorm_h.running_var"],
                                               sd["batchnorm_h.weight"], sd["batchnorm_h.bias"], False)
            y = torch.from_numpy(x[:n]) + torch.relu(y)
            t = time.perf_counter() - t0
            if best is None or t < best[0]:
                best = (t, threads)
    t, threads = best
    return {"value": e_n / t, "unit": "edges/s", "cores": threads, "kind": "port",
            "host_logical_cpus": ncpu,
            "sample": f"destination rows [0,{n}) of the same graph = {e_n} edges, 1 layer forward, best of 2 at "
                      f"{threads} threads (fastest of {ncpu}, {ncpu // 2}, {ncpu // 4}; {t:.2f} s); C/OpenMP port of "
                      f"reduce_func + torch CPU Linear/BN/ReLU"}


def other_workload_leg(extra, timeout_s=150):
    """One more bench.py call in a CHILD process (its own device context; a failure or a timeout costs this field, not the line):
    -> the child's JSON line cut down to what a reader compares, or {"error": ...}."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + list(extra)
    try:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout_s, text=True, env=env, cwd=ROOT)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not lines:
            return {"error": f"child exited {out.returncode} without a line", "command": " ".join(cmd[1:])}
        d = json.loads(lines[-1])
        r = d.get("roofline") or {}
        return {"command": "python " + " ".join(os.path.relpath(c, ROOT) if os.path.isabs(c) else c for c in cmd[1:]),
                "workload": (d.get("config") or {}).get("workload"), "ms_per_step": d.get("ms_per_step"), "value": d.get("value"), "unit": d.get("unit"),
                "steps": d.get("steps"), "warmup": d.get("warmup"), "prewarm_steps_untimed": d.get("prewarm_steps_untimed"),
                "roofline": {k: r.get(k) for k in ("kernel", "frac", "ms_per_launch", "achieved", "unit", "rest_rows_beside_kernel", "full_grid", "traffic",
                                                     "traffic_source", "traffic_over_algorithmic", "traffic_GB_per_s", "algorithmic_bytes_per_launch", "read_only_frac")},
                "roofline_layer_frac": (d.get("roofline_layer") or {}).get("frac"), "parity_check": d.get("parity_check")}
    except Exception as ex:                                         # (timeout, JSON, OS): never the parent's problem
        return {"error": repr(ex), "command": " ".join(cmd[1:])}


def sampled_check(g, h_ext, y, layer_sd, avg_log, n_rows, lo=0):
    """Rows of the timed step's OUTPUT against the oracle, after the timed loop: the C restatement of reduce_func on the
    sampled rows' in-edges (oracle/pna_oracle.c), then Linear / BatchNorm / ReLU / residual in float64.  Proves the timed
    kernels did the work; the full-size bit-level properties live in tests/test_gpu_fullsize.py."""
    import numpy as np
    from oracle import c_oracle
    csr = g.csr
    V = g.num_nodes
    deg = (csr.rowptr[1:] - csr.rowptr[:-1]).long()
    gen = torch.Generator().manual_seed(7)
    rows = torch.randint(0, V, (n_rows,), generator=gen)
    rows[0] = int(torch.argmax(deg))                                   # a hub row (heavy-segment path)
    rows[1] = int(torch.argmin(deg))
    rows = torch.unique(rows).to(csr.rowptr.device)
    beg, end = csr.rowptr[rows].long(), csr.rowptr[rows + 1].long()
    cnt = end - beg
    rp = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=rows.device)
    rp[1:] = torch.cumsum(cnt, 0)
    pos = torch.arange(int(rp[-1]), device=rows.device) - torch.repeat_interleave(rp[:-1], cnt) + torch.repeat_interleave(beg, cnt)
    srcs = csr.col[pos].long()
    uniq, inv = torch.unique(srcs, return_inverse=True)
    Fw = y.shape[1]
    x_sub = h_ext[uniq][:, :F_of(layer_sd)].float().cpu().numpy()
    amp, att = c_oracle.degree_scalers(rp.cpu().numpy().astype(np.int32), float(avg_log))
    agg = c_oracle.segreduce(rp.cpu().numpy().astype(np.int32), inv.cpu().numpy().astype(np.int32), x_sub, x_sub.shape[1],
                             AGGREGATORS.split(), [None, amp, att])
    sd = {k: v.double().cpu() for k, v in layer_sd.items()}
    W, b = sd["posttrans.fully_connected.0.linear.weight"], sd["posttrans.fully_connected.0.linear.bias"]
    a64 = torch.from_numpy(agg).double()
    z = a64 @ W.t() + b
    bn_scale = sd["batchnorm_h.weight"] / torch.sqrt(sd["batchnorm_h.running_var"] + 1e-5)
    z = (z - sd["batchnorm_h.running_mean"]) * bn_scale + sd["batchnorm_h.bias"]
    ref = h_ext[rows][:, :Fw].double().cpu() + torch.relu(z)
    got = y[rows].double().cpu()
    # the north star's bar, per element: 1e-5 relative + the fp32 rounding floor of a K = 12F sum in another order,
    # C_EPS x sum_k |w_k a_k| (tests/conftest.py check_blocks' floor model, C_EPS = 2e-6), carried through BatchNorm's scale
    mass = (a64.abs() @ W.abs().t() + b.abs()) * bn_scale.abs()
    tol = 1e-5 * ref.abs() + 2e-6 * mass
    errs = (got - ref).abs()
    err = errs.max().item()
    scale = ref.abs().max().item()
    worst = (errs / tol.clamp(min=1e-30)).max().item()
    return {"rows": int(rows.numel()), "edges": int(rp[-1]), "max_abs_err": err, "max_abs_ref": scale, "rel_to_max": err / max(scale, 1e-30),
            "worst_err_over_tolerance": worst, "ok": bool(worst <= 1.0),
            "tolerance": "per element 1e-5 |ref| + 2e-6 sum_k |w_k a_k| |bn scale| (fp32 layer vs float64 contraction of the fp32 oracle aggregate)"}


def _finite(x):
    """The record with every non-finite float replaced by None (a failed diagnostic leaves inf / nan behind: not JSON)."""
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    if isinstance(x, float) and not (x == x and abs(x) != float("inf")):
        return None
    return x


def F_of(layer_sd):
    return layer_sd["posttrans.fully_connected.0.linear.weight"].shape[1] // 12


def main():
    args = parse()
    global F
    if args.workload == "c5":
        F = 128
        if args.nodes_per_gpu == V_PER_GPU and args.edges_per_gpu == E_PER_GPU:
            args.nodes_per_gpu, args.edges_per_gpu = 2_000_000, 20_000_000
        if args.x_pitch == 80:
            args.x_pitch = 128
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path)"
    # PNA_BENCH_ONE_DEVICE=1 PNA_BENCH_BACKEND=gloo: smoke-test the N > 1 code path on a box with ONE GPU (all ranks on
    # cuda:0, exchange staged through the host; timings are meaningless) -- tools/gpu_check.sh does this with N=2
    one_dev = os.environ.get("PNA_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("PNA_BENCH_BACKEND", "nccl")
    local_dev = 0 if one_dev else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from pna_amd import Graph
    from pna_amd.dgl.pna_layer import PNASimpleLayer
    from pna_amd.shard import shard_graph
    from pna_amd.synth import powerlaw_graph
    from pna_amd import ops, functional as PF

    V, E = args.nodes_per_gpu * world, args.edges_per_gpu * world
    src, dst = powerlaw_graph(V, E, seed=1234, device=dev)          # identical on every rank
    deg = torch.bincount(dst, minlength=V)
    avg_log = torch.log(deg.double() + 1).mean().float()            # avg_d['log'] of this graph (main_HIV.py:240-244)
    if world > 1:
        g = shard_graph(src, dst, V, balance=args.balance)
        lo, hi = g.lo, g.hi
    else:
        g = Graph(src, dst, V)
        lo, hi = 0, V
    n_local = hi - lo
    torch.cuda.synchronize()
    t_csr0 = time.perf_counter()
    e_local = int(g.csr.rowptr[-1].item())                           # first use builds the CSR (pna_collate_csr_i32); once per graph
    torch.cuda.synchronize()
    csr_build_ms = (time.perf_counter() - t_csr0) * 1e3               # reported separately, not part of a step
    hs = g.heavy_schedule()
    # x ~ N(0,1), seed 1234.  N=1: generated on the host so the CPU baseline sees the same values; N>1:
    # every rank draws only its own rows (per-rank seed) -- an 8M x 75 host tensor per rank would only slow start-up
    h_all = torch.randn(V, F, generator=torch.Generator().manual_seed(1234)) if world == 1 else None
    # node features live in a 16-byte aligned row pitch (80 floats for F=75), the layout a multi-layer net keeps its activations
    # in (this library's layers write their outputs at such a pitch).  Since round 4 the one-kernel layer takes ANY pitch: a
    # contiguous (V, 75) tensor -- what a drop-in caller of PNASimpleLayer.forward passes -- lands on the same kernel; the step
    # over such a table is timed too (`ms_per_step_contiguous_input`, 3-7 % slower: 300-byte rows straddle more 64-byte sectors)
    if world > 1:
        # multi-GPU: the features live in the shard's resident [local | halo] table, so the halo exchange of the
        # timed step receives the peers' rows in place (no concatenation pass)
        # round 3: rows at the smallest 16-byte aligned pitch (F = 75 -> 76 floats: 1.3 % padding on the wire) -- what the
        # one-kernel layer reads; `--x-pitch 75` restores the dense rows
        h = g.alloc_features(F, pitch=(F + 3) // 4 * 4 if args.x_pitch == 80 else max(args.x_pitch, F), device=dev)
    else:
        h_buf = torch.zeros(hi - lo, max(args.x_pitch, F), device=dev)
        h = h_buf[:, :F]
    if world == 1:
        h.copy_(h_all)
    else:
        h.copy_(torch.randn(hi - lo, F, device=dev, generator=torch.Generator(device=dev).manual_seed(1234 + rank)))

    torch.manual_seed(0)
    layer = PNASimpleLayer(F, F, AGGREGATORS, SCALERS, {"log": avg_log}, 0.0, True, True)
    with torch.no_grad():                                            # random-init weights of the architecture, O(1) activations
        for p in layer.parameters():
            p.copy_(torch.randn_like(p) / (p.shape[-1] ** 0.5 if p.dim() == 2 else 3.0))
    layer_sd = {k: v.clone() for k, v in layer.state_dict().items()}
    layer = layer.to(dev).eval()
    g.degree_scalers(float(avg_log))

    # ---- per-graph / per-weight set-up of the one-kernel layer, timed on its own BEFORE any step (set-up reported
    #      separately).  Once per graph: the degree plan (row order, tile-major id records, descriptors) and the
    #      rest rows' work list; once per (weights, graph): the packed weight images W_D (fp16 x 2 and bf16 x 3).  Neither is part of a step.
    setup = None
    if world == 1 and hasattr(layer, "_degree_grouped_path"):
        from pna_amd import degree_groups as _DGs
        with torch.no_grad():
            # (round 6, VERDICT r5 item 6: the FIRST plan of a process used to be what this clock saw -- 440 ms, of which 436 were torch loading
            # the code objects of sort / unique / repeat_interleave / index kernels on their first use (tools/plan_build_time.py: a second,
            # fresh Graph of the same size builds its plan in 3.5 ms).  A small throw-away graph goes first; its time is reported as
            # `first_plan_in_process_ms`, and `degree_plan_build_ms` is what every further graph of this size costs.)
            from pna_amd.synth import powerlaw_graph as _plg
            torch.cuda.synchronize()
            t_s = time.perf_counter()
            _ws, _wd = _plg(200_000, 2_000_000, seed=7, device=dev)
            _wg = Graph(_ws, _wd, 200_000)
            _wp = _DGs.plan_of(_wg)
            _wp.fused_tables()
            if _wp.NR:
                _wp.rest_items(_wg)
            _wp.fused_balance(PF._fused_grid(dev, 0, max(_wp.NV // 64, 1)))
            torch.cuda.synchronize()
            t_first_plan = (time.perf_counter() - t_s) * 1e3
            del _ws, _wd, _wg, _wp
            torch.cuda.synchronize()
            t_s = time.perf_counter()                         # (the path check below is what builds the plan: inside the clock)
            if layer._degree_grouped_path(g, h) and _DGs.fused_applies(g, h, F, F):
                from pna_amd.dgl.pna_layer import _row_scales
                plan_s = _DGs.plan_of(g)
                tabs = plan_s.fused_tables()
                if plan_s.NR:
                    plan_s.rest_items(g)
                for sp in (0, _DGs.FUSED_SPARE_WGS):            # the tile lists of the two grids the step and the diagnostics launch
                    plan_s.fused_balance(PF._fused_grid(dev, sp, plan_s.NV // 64))
                torch.cuda.synchronize()
                t_plan = (time.perf_counter() - t_s) * 1e3
                t_s = time.perf_counter()
                img_s, _ = _DGs.fused_images(layer.posttrans.fully_connected[0].linear.weight, F, _row_scales(g, layer.scalers, layer.avg_d, dev), plan_s)
                torch.cuda.synchronize()
                t_img = (time.perf_counter() - t_s) * 1e3
                nbytes = lambda t: 0 if t is None else int(t.numel() * t.element_size())   # noqa: E731
                rest = plan_s.rest_items(g) if plan_s.NR else (None, None, None)
                setup = {"degree_plan_build_ms": t_plan, "first_plan_in_process_ms": t_first_plan, "weight_image_pack_ms": t_img, "csr_build_ms": csr_build_ms,
                         "plan_device_bytes": {"row_perm": nbytes(plan_s.perm) + nbytes(plan_s.perm_rest), "tile_desc": nbytes(tabs[0]),
                                               "tile_ids": nbytes(tabs[1]), "rest_work_list": nbytes(rest[0]) + nbytes(rest[1]),
                                               "node_to_plan_row": nbytes(plan_s._vmap),
                                               "two_kernel_work_list": nbytes(plan_s._items[0]) if plan_s._items else 0,   # (built when the two-kernel grouped path / training asks)
                                               "balanced_tile_lists": sum(nbytes(x) for v_ in plan_s.__dict__.get("_fused_bal", {}).values() for x in v_)},
                         "weight_images_bytes": nbytes(img_s), "degree_groups": plan_s.G,
                         "note": "host wall-clock around the first build, device idle before and synchronised after; once per graph (plan) / "
                                 "once per (weights, graph) (images), amortised over layers, steps and epochs; a one-shot forward on a fresh "
                                 "graph pays csr + plan + images + one step"}
                setup["plan_device_bytes"]["total"] = sum(setup["plan_device_bytes"].values())

    def step():
        with torch.no_grad():
            return layer(g, h)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.layers > 1:
        # ---- a stack of layers per step.  N = 1: the layers back to back on the one-kernel path, activations
        #      kept at the 16-byte pitch; N > 1: BlockPipeline -- block b of layer L is packed and sent while blocks b+1.. are still
        #      being computed; per-layer kernels = the one-kernel layer over the block's own degree plan (round 4: DegreePlan(row_range);
        #      block 0 also takes the hub rows and every block's leftover rows), else gather + three-block contraction per block
        import copy
        from pna_amd.shard import BlockPipeline
        L = args.layers
        layers = [layer] + [copy.deepcopy(layer) for _ in range(L - 1)]
        P = max(args.x_pitch, (F + 3) // 4 * 4)
        if world > 1:
            pipe = BlockPipeline(g, args.blocks)
            ta = torch.zeros(n_local + g.n_halo, P, device=dev)
            tb = torch.zeros_like(ta)
            ta[:n_local, :F] = h
            rows_fn = PF.SimpleLayerRows(layers, g, args.blocks)

            def step_l():
                with torch.no_grad():
                    return pipe.run(rows_fn, L, ta, tb)
        else:
            def step_l():
                with torch.no_grad():
                    x = h
                    for i, lay in enumerate(layers):
                        x = lay(g, x)                    # (the one-kernel path writes rows at an aligned pitch: the next layer stays on it)
                    return x
        # the same untimed pre-conditioning as the one-layer line below (first calls: plans, weight images; then the device's
        # sustained clock state): without it 3 + 10 steps of 4 layers measured 0.80 / 0.98 / 1.28 ms per layer on three boxes
        prewarm_l = 0
        gc.collect()
        gc.disable()                                           # (see the one-layer loop below)
        if not args.no_prewarm:
            for _ in range(2):
                step_l()
            prewarm_l = 2 + (16 if world > 1 else 48)         # (N > 1: every rank the same, fixed number of collectives)
            for _ in range(prewarm_l - 2):
                step_l()
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            step_l()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_l()
        sync()
        dt = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        if rank == 0:
            print(json.dumps({"metric": "PNA-layer fwd edges/sec (F=%d, 4 aggr x 3 scalers), %d stacked layers per step" % (F, L),
                              "value": E * L / (dt / args.steps), "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": dt / args.steps * 1e3, "ms_per_layer": dt / args.steps * 1e3 / L, "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "prewarm_steps_untimed": prewarm_l,
                              "config": {"workload": "%d x PNASimpleLayer(F=%d) stacked, |V|=%d |E|=%d per GPU" % (L, F, args.nodes_per_gpu, args.edges_per_gpu),
                                         "layers": L, "row_blocks": args.blocks if world > 1 else None,
                                         "exchange": "block-pipelined point-to-point (pna_amd.shard.BlockPipeline)" if world > 1 else None,
                                         "halo_rows_rank0": int(getattr(g, "n_halo", 0))}}))
        if world > 1:
            dist.destroy_process_group()
        return

    # Pre-conditioning, untimed, BEFORE the W warm-up steps: ~150 ms of the same step.  A 0.9 ms step needs more than W = 5 of them
    # for the device to reach its sustained clock state: measured on one box, W = 5 / K = 20 gave 0.899 ms/step, W = 50 0.838, K = 200
    # 0.841 -- the first ~50 steps after an idle phase run 7 % slower.  The timed region below is still exactly K steps after W
    # warm-up steps, bracketed by barrier + synchronize.
    # The interpreter's cyclic garbage collector stays out of the pre-conditioning, W and K steps (collected in front of them, re-enabled after -- what `timeit` does):
    # a full collection of this process' ~1 M objects takes ~47 ms, and WHICH step it lands on depends on the number of objects allocated
    # so far -- `python bench.py` without arguments had it inside the 20 timed steps (2.5-3.9 ms per step reported for 0.72 ms steps), any
    # argument moved it out (found in round 6 with PNA_BENCH_STEP_TIMES=1: one step of 47.65 ms, the other 59 of 0.84-0.91).  Collected IN FRONT of
    # the pre-conditioning steps: the collection itself leaves the device idle for those 47 ms, and the first steps after an idle phase run slower.
    gc.collect()
    gc.disable()
    prewarm = 0
    if not args.no_prewarm:
        for _ in range(2):                                     # (first calls: degree plan, weight images, allocator)
            step()
        torch.cuda.synchronize()
        prewarm, t_pre = 2, time.perf_counter()
        if world > 1:                                          # a step holds a collective: every rank the SAME, fixed number of them
            for _ in range(64):
                step()
            prewarm += 64
            torch.cuda.synchronize()
        else:
            while time.perf_counter() - t_pre < 0.15:
                for _ in range(16):
                    step()
                prewarm += 16
                torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    sync()
    if os.environ.get("PNA_BENCH_STEP_TIMES") == "1":            # diagnosis only: every step synchronised and timed on its own (to stderr)
        ts = []
        for _ in range(3 * args.steps):
            t_ = time.perf_counter(); step(); sync(); ts.append(round((time.perf_counter() - t_) * 1e3, 3))
        print("[bench] per-step ms:", ts, file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = E / (dt / args.steps)

    # ---- the timed kernels did the work: sampled rows of one more step's output against the oracle --------------------
    check = None
    if rank == 0 or world > 1:
        try:
            with torch.no_grad():
                y_chk = layer(g, h)
                h_ext_chk = g.source_features(h)
            check = sampled_check(g, h_ext_chk, y_chk, layer_sd, avg_log, args.check_rows)
        except Exception as ex:                                     # reported, never silently dropped
            check = {"ok": False, "error": repr(ex)}
        if not check.get("ok"):
            print(f"[bench] rank {rank}: sampled parity check FAILED: {check}", file=sys.stderr, flush=True)
    # N > 1: every rank's verdict in the line (rank 0 prints it; a failed check on ANY rank is visible and never costs the line)
    check_ranks = None
    if world > 1:
        mine = torch.tensor([1.0 if (check or {}).get("ok") else 0.0, float((check or {}).get("worst_err_over_tolerance", float("nan")))],
                            device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        check_ranks = [{"rank": i, "ok": bool(t[0].item() == 1.0), "worst_err_over_tolerance": float(t[1].item())} for i, t in enumerate(allr)]

    # ---- cold-cache leg: 3 rotating copies of the feature table and of the graph's index arrays, so that no
    # step finds its 300 MB of features / 40 MB of source ids in the 256 MiB Infinity Cache from the previous step --------
    ms_per_step_cold = None
    if world == 1 and not args.no_cold:
        copies = []
        for i in range(3):
            gi = Graph(src, dst, V)
            gi._csr = type(csr0 := g.csr)(csr0.rowptr.clone(), csr0.col.clone(), csr0.eid, csr0.row, csr0.max_degree)
            gi.degree_scalers(float(avg_log)); gi.heavy_schedule(); gi.work_items()
            hb = torch.zeros(hi - lo, max(args.x_pitch, F), device=dev)
            hb[:, :F].copy_(h)
            copies.append((gi, hb[:, :F]))
        with torch.no_grad():
            for i in range(3):
                layer(*copies[i])
            sync()
            t1 = time.perf_counter()
            for i in range(args.steps):
                layer(*copies[i % 3])
            sync()
        ms_per_step_cold = (time.perf_counter() - t1) / args.steps * 1e3
        del copies

    # ---- the same step over a CONTIGUOUS (V, F) feature tensor (the headline must be reachable through the
    # reference API: PNASimpleLayer.forward(g, h) with h = torch.randn(V, 75)) -------------------------------------------------
    ms_per_step_contig, contig_one_kernel, contig_same_bits = None, None, None
    if world == 1:
        from pna_amd import degree_groups as _DG
        h_c = torch.empty(n_local * F + 8, device=dev)[:n_local * F].view(n_local, F)     # (storage covers the last row's rounded-up strip)
        h_c.copy_(h)
        with torch.no_grad():
            contig_one_kernel = bool(_DG.FUSED and hasattr(layer, "_degree_grouped_path") and layer._degree_grouped_path(g, h_c)
                                     and _DG.fused_applies(g, h_c, F, F))
            for _ in range(max(args.warmup, 3)):
                layer(g, h_c)
            sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                layer(g, h_c)
            sync()
            ms_per_step_contig = (time.perf_counter() - t1) / args.steps * 1e3
            y_c = layer(g, h_c)
            contig_same_bits = bool(torch.equal(y_c, layer(g, h)))
        del h_c
    # ---- the same step replayed from a hipGraph (pna_amd.capture.GraphedForward: the kernel, the fork / join to the second stream and
    # the rest rows' chain as ONE graph launch): what the host side of the eager step costs.  Informative; `value` stays the eager step.
    ms_per_step_hipgraph = None
    if world == 1:
        try:
            from pna_amd.capture import GraphedForward
            with torch.no_grad():
                gf = GraphedForward(lambda x: layer(g, x), h, alias_inputs=True)
                for _ in range(max(args.warmup, 10)):
                    gf.graph.replay()
                sync()
                ms_per_step_hipgraph = float("inf")
                for _ in range(2):                         # (a diagnostic leg: the better of two passes of K replays)
                    t1 = time.perf_counter()
                    for _ in range(args.steps):
                        gf.graph.replay()
                    sync()
                    ms_per_step_hipgraph = min(ms_per_step_hipgraph, (time.perf_counter() - t1) / args.steps * 1e3)
                if not torch.equal(gf.static_out, layer(g, h)):
                    ms_per_step_hipgraph = None
            del gf
        except Exception as ex:   # noqa: BLE001  (never sinks the line)
            print(f"[bench] hipGraph leg skipped: {ex}", file=sys.stderr)
            ms_per_step_hipgraph = None

    # the same step with the contraction forced onto the exact f32-input MFMA (reported beside the headline number)
    from pna_amd import ops as _ops
    arith = "bf16x3" if (_ops.POSTTRANS_ARITH == "bf16x3" or (_ops.POSTTRANS_ARITH == "auto" and n_local >= _ops.X3_MIN_ROWS)) else "f32"
    ms_per_step_f32 = ms_per_step
    if arith != "f32":
        keep = _ops.POSTTRANS_ARITH
        _ops.POSTTRANS_ARITH = "f32"
        for _ in range(2):
            step()
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        ms_per_step_f32 = (time.perf_counter() - t1) / args.steps * 1e3
        _ops.POSTTRANS_ARITH = keep

    # the same step under the other arithmetics of the one-kernel layer: bf16 x 3 everywhere (rounds 3-4, what
    # PNA_AMD_POSTTRANS=bf16x3 selects) and round 5's unguarded fp16 x 2 -- and what the guard of the default did in a step
    arith_legs, guard_info = {}, None
    try:
        from pna_amd import degree_groups as _DG
        with torch.no_grad():
            one_kernel = hasattr(layer, "_degree_grouped_path") and layer._degree_grouped_path(g, h) and _DG.fused_applies(g, g.source_features(h), F, F)
        if one_kernel:
            keep_arith = _DG.FUSED_ARITH
            plan_ = _DG.plan_of(g)
            if _DG.fused_arith() == 0:
                _DG.guard_stats(plan_, dev, reset=True)
                step()
                sync()
                handed, calls_ = _DG.guard_stats(plan_, dev)
                guard_info = {"tiles_handed_over_to_bf16x3_per_step": handed / max(calls_, 1), "tiles": plan_.NV // 64,
                              "note": "64-row tiles of the one-kernel layer whose outputs the fp16 x 2 floor-error bound did not certify: computed again "
                                      "in bf16 x 3 by the second launch of the same call (pna_amd/csrc/pna_x3_split.h, DESIGN.md 4.8.17)"}
            for name in ("bf16x3", "fp16x2"):
                if name == keep_arith:
                    continue
                _DG.FUSED_ARITH = name
                try:
                    for _ in range(3):
                        step()
                    sync()
                    t1 = time.perf_counter()
                    for _ in range(args.steps):
                        step()
                    sync()
                    arith_legs[name] = (time.perf_counter() - t1) / args.steps * 1e3
                finally:
                    _DG.FUSED_ARITH = keep_arith
    except Exception as ex:   # noqa: BLE001  (never sinks the line)
        print(f"[bench] arithmetic legs skipped: {ex}", file=sys.stderr)

    # ---- per-kernel timing of the dominant kernels (HIP events, this rank) ------------------------------
    csr = g.csr
    with torch.no_grad():
        x_ext = g.source_features(h)                                   # (synchronous form: the halo has landed)
        # (the two collectives of this section first, on every rank: whatever the rank-local diagnostics below do -- they are caught --
        # no rank can leave another waiting in an all-to-all, and rank 0 always prints its line: VERDICT r4 item 5a)
        t_halo = event_time_ms(lambda: g.source_features(h), args.kernel_iters) if world > 1 else 0.0
    diag_error = None
    t_seg = t_post = t_post_f32 = float("inf")
    grouped = fused = power = halo_rate = None
    try:
      with torch.no_grad():
        t_seg = event_time_ms(lambda: PF.aggregate(g, x_ext, F, AGGREGATORS.split()), args.kernel_iters)
        agg = PF.aggregate(g, x_ext, F, AGGREGATORS.split())
        lin = layer.posttrans.fully_connected[0].linear
        amp, att = g.degree_scalers(float(avg_log))
        t_post = event_time_ms(lambda: PF.posttrans(agg, 4 * F, lin.weight, lin.bias, [None, amp, att]), args.kernel_iters)
        t_post_f32 = event_time_ms(lambda: _ops.posttrans(agg, 4 * F, lin.weight, [None, amp, att], lin.bias, arith="f32"), args.kernel_iters)
        # what the timed step actually launches when the layer groups its rows by in-degree (pna_amd/degree_groups.py): the same
        # gather writing in degree order, then one combined scaler block per degree tile + the three-block rest
        grouped = None
        if hasattr(layer, "_degree_grouped_path") and layer._degree_grouped_path(g, h):
            from pna_amd import degree_groups as DG
            plan = DG.plan_of(g)
            agg_g = PF.degree_grouped_aggregate(layer, g, h, plan)
            y_g = torch.empty(n_local, F, device=dev)
            t_seg_plain, t_post_plain = t_seg, t_post
            t_seg = event_time_ms(lambda: PF.degree_grouped_aggregate(layer, g, h, plan, out=agg_g, x=x_ext), args.kernel_iters)
            t_post = event_time_ms(lambda: PF.degree_grouped_posttrans(layer, g, h, agg_g, plan, out=y_g), args.kernel_iters)
            grouped = {"degree_groups": plan.G, "rows_in_groups": int((plan.perm >= 0).sum().item()), "padded_rows": plan.NV,
                       "rest_rows": plan.NR, "tile_rows": DG.TILE,
                       "segreduce_natural_order_ms": t_seg_plain, "three_block_contraction_ms": t_post_plain}
        # ... and when the group rows run in ONE kernel (pna_fused_degree_f32): that launch, and the two-kernel path of the rest rows
        fused = None
        if grouped is not None:
            from pna_amd import degree_groups as DG
            if DG.fused_applies(g, x_ext, F, F):
                call = PF.FusedDegreeCall(layer, g, h, x=x_ext)
                # (the kernel as the step launches it: on a large graph it leaves DG.FUSED_SPARE_WGS workgroups out and the rest
                # rows' launches run beside it on a second stream -- functional.run_fused_call; timed alone here, each on its own)
                beside = plan.rest_overlap_applies(F)
                call.set_spare(beside)                    # (also binds the tile list balanced for THAT grid: DegreePlan.fused_balance)
                t_fused = event_time_ms(call.group_rows, args.kernel_iters)
                t_rest = event_time_ms(call.rest_rows, args.kernel_iters)
                t_fused_full = t_fused
                if beside:                                # ... and on the whole device (what the step launched before ABI 17)
                    call.set_spare(False)
                    t_fused_full = event_time_ms(call.group_rows, args.kernel_iters)
                    call.set_spare(True)
                rows_g = grouped["rows_in_groups"]
                deg_l = (csr.rowptr[1:] - csr.rowptr[:-1]).long()
                e_g = int(deg_l[plan.perm[plan.perm >= 0].long()].sum().item())
                from pna_amd import _lib as _plib
                fused = {"ms_group_rows_kernel": t_fused, "ms_rest_rows_two_kernel_path": t_rest, "rows": rows_g, "edges": e_g,
                         "arith": _plib.FD_ARITH_NAMES[call.arith],
                         "padded_rows": plan.NV, "id_records": plan.fused_tables()[2],
                         "rest_rows_beside_kernel": bool(beside), "spare_workgroups": int(call.args.spare_workgroups),
                         "tile_order": DG.FUSED_BALANCE if plan.fused_balance(PF._fused_grid(dev, int(call.args.spare_workgroups), plan.NV // 64)) is not None else "plan order",
                         "ms_group_rows_kernel_full_grid": t_fused_full}
        # N > 1: the exchange priced against the links it crosses.  Rank 0's received + sent halo bytes (whole rows of the resident
        # table, pitch ldx floats) over the standalone exchange time, beside ONE xGMI link's ~153 GB/s per direction: the all-to-all
        # spreads over world - 1 peers (one link each on an 8-GPU node), so the per-link rate is the total / (world - 1).
        halo_rate = None
        if world > 1 and t_halo > 0:
            pitch_b = int(x_ext.stride(0)) * 4
            rb, sb = int(sum(g.recv_splits)) * pitch_b, int(sum(g.send_splits)) * pitch_b
            halo_rate = {"ms_rank0": t_halo, "recv_bytes_rank0": rb, "send_bytes_rank0": sb, "row_bytes": pitch_b,
                         "recv_GB_per_s_rank0": rb / (t_halo * 1e-3) / 1e9, "send_GB_per_s_rank0": sb / (t_halo * 1e-3) / 1e9,
                         "recv_GB_per_s_per_peer_link": rb / (t_halo * 1e-3) / 1e9 / (world - 1),
                         "xgmi_link_peak_GB_per_s_per_direction": 153.0,
                         # the link arithmetic a first hardware run is judged against: every peer is one point-to-point
                         # xGMI link, the all-to-all crosses world - 1 of them at once, so the exchange cannot take less than
                         # halo bytes / ((world - 1) x 153 GB/s) -- and a step cannot take less than that when nothing hides the exchange,
                         # nor less than max(exchange, compute) when everything does
                         "halo_bytes_per_rank": rb,
                         "exchange_ms_at_link_peak": rb / ((world - 1) * 153.0e9) * 1e3,
                         "max_speedup_by_link_bound": None,      # (filled in below, from the step's compute time)
                         "share_of_step": t_halo / ms_per_step if ms_per_step else None,
                         "note": "standalone, synchronous exchange (pack + all_to_all_single); inside the step it runs beside the interior rows"}
        # socket power and shader clock while each kernel runs alone (rocm-smi; best effort, N = 1 only): the bf16x3 contraction
        # runs at the package power cap and the firmware lowers the clock to hold it -- the dense MFMA peak at THAT clock is the
        # ceiling the kernel can be priced against (DESIGN.md 4.2c point 6, profiles/r02_power_probe.txt)
        power = None
        if world == 1 and not args.no_power_probe:
            power = {"contraction": power_probe(lambda: PF.posttrans(agg, 4 * F, lin.weight, lin.bias, [None, amp, att])),
                     "segment_reduce": power_probe(lambda: PF.aggregate(g, x_ext, F, AGGREGATORS.split()))}
            if grouped is not None:
                power["contraction_degree_grouped"] = power_probe(lambda: PF.degree_grouped_posttrans(layer, g, h, agg_g, plan, out=y_g))
    except Exception as ex:                                         # noqa: BLE001  (reported in the line, never instead of it)
        diag_error = repr(ex)
        print(f"[bench] rank {rank}: per-kernel diagnostics failed: {diag_error}", file=sys.stderr, flush=True)
        if not (t_seg < float("inf")):
            grouped = fused = None
    alg_read = e_local * (4 * F + 4) + 4 * (n_local + 1)
    alg_write = n_local * 16 * F
    alg_bytes = alg_read + alg_write
    # `traffic` is NOT measured in this run (PMC counters need their own rocprofv3 passes): it is the committed result of
    # the passes named in traffic_source, for the C3 shape at N=1 only
    traffic = traffic_post = traffic_source = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath) and args.workload == "c3" and world == 1 and args.nodes_per_gpu == V_PER_GPU and args.edges_per_gpu == E_PER_GPU:
        try:
            tj = json.load(open(tpath))
            traffic = tj.get("pna_segreduce_c3", {}).get("hbm_bytes_per_launch")
            traffic_post = tj.get("pna_posttrans_x3_c3" if arith == "bf16x3" else "pna_posttrans_f32_c3", {}).get("hbm_bytes_per_launch")
            traffic_source = ("profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/prof_kernels.py, "
                              + tj.get("collected", "round 1") + "; NOT measured in this run")
        except Exception:
            traffic = traffic_post = traffic_source = None
    roofline = {"bound": "hbm", "kernel": "k_segreduce_fast<4> + k_heavy_finalize (pna_segreduce_fwd_f32)" + (", aggregate written in degree order" if grouped else ""),
                "achieved": alg_bytes / (t_seg * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": alg_bytes / (t_seg * 1e-3) / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_source,
                "ms_per_launch": t_seg, "algorithmic_bytes_per_launch": alg_bytes,
                "read_only_frac": alg_read / (t_seg * 1e-3) / HBM_PEAK,
                "edges_per_s_kernel_only": e_local / (t_seg * 1e-3),
                "heavy_rows": hs.n_heavy, "heavy_segments": hs.n_seg}
    roofline_segreduce = None
    if fused is not None:
        # the dominant kernel of the step is the one-kernel layer over the group rows.  Algorithmic bytes of ITS units (SURVEY 8d,
        # the aggregate never written): per edge 4F + 4 (source row + its id), per 16 rows a 16-byte descriptor, per row 4F read
        # (residual) + 4F written (y) + 4 (perm).  The north star's "HBM-read roofline of the segment-reduce" is the gather part.
        e_g, rows_g = fused["edges"], fused["rows"]
        gather_read = e_g * (4 * F + 4) + plan.NV
        fused_bytes = gather_read + rows_g * (8 * F + 4)
        tf = fused["ms_group_rows_kernel"] * 1e-3
        roofline_segreduce = roofline
        roofline_segreduce["note"] = "the standalone gather kernel (still shipped: rest rows, tower layers, training); not on this step's path for the group rows"
        shape = "two gather passes of 2 full feature blocks, two panels of 64 output columns" if F > 96 else "2 full + half feature blocks"
        roofline = {"bound": "hbm", "kernel": "k_fused_degree<" + shape + "> (pna_fused_degree_f32): gather + 4 aggregators + combined scaler block + posttrans "
                                              "contraction + BN / ReLU / residual; the 4F aggregate never reaches HBM",
                    "achieved": fused_bytes / tf / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": fused_bytes / tf / HBM_PEAK,
                    "traffic": None, "traffic_source": None,
                    "ms_per_launch": fused["ms_group_rows_kernel"], "algorithmic_bytes_per_launch": fused_bytes,
                    "gather_read_bytes": gather_read, "read_only_frac": gather_read / tf / HBM_PEAK,
                    "edges_per_s_kernel_only": e_g / tf, "rows": rows_g, "edges": e_g,
                    "rest_rows_two_kernel_path_ms": fused["ms_rest_rows_two_kernel_path"],
                    # (three fp16 partial products per multiply since round 5 -- six bf16 ones before; the f16 pipe's dense peak is the bf16 pipe's)
                    "mfma_frac_executed_flops": 2.0 * plan.NV * (4 * F) * F / tf / (MFMA_BF16_PEAK / 3),
                    "contraction_arith": ARITH_TEXT[fused["arith"]],
                    "rest_rows_beside_kernel": fused["rest_rows_beside_kernel"], "spare_workgroups": fused["spare_workgroups"],
                    "tile_order": fused["tile_order"],
                    "full_grid": {"ms_per_launch": fused["ms_group_rows_kernel_full_grid"],
                                  "frac": fused_bytes / (fused["ms_group_rows_kernel_full_grid"] * 1e-3) / HBM_PEAK,
                                  "note": "the same kernel on every workgroup slot of the device (spare_workgroups = 0): its own ceiling; the step "
                                          "trades this for the rest rows running beside it"},
                    "note": "units of this launch: the rows of the degree groups (their edges); hub rows and rare degrees take the two-kernel rest path"
                            + (" -- in the step on a second stream BESIDE this kernel, which leaves spare_workgroups of its 2-per-CU workgroups out for them"
                               " (both timed alone here)" if fused["rest_rows_beside_kernel"] else "")}
        ftp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        default_c3 = args.workload == "c3" and args.nodes_per_gpu == V_PER_GPU and args.edges_per_gpu == E_PER_GPU
        default_c5 = args.workload == "c5" and args.nodes_per_gpu == 2_000_000 and args.edges_per_gpu == 20_000_000
        if os.path.exists(ftp) and world == 1 and (default_c3 or default_c5):
            try:
                tj = json.load(open(ftp)).get("pna_fused_degree_c3" if default_c3 else "pna_fused_degree_c5")
                if tj:
                    roofline["traffic"] = tj.get("hbm_bytes_per_launch")
                    roofline["traffic_source"] = ("profiles/hbm_traffic.json (pna_fused_degree_" + ("c3" if default_c3 else "c5") + "): " + tj.get("collected", "")
                                                  + "; NOT measured in this run")
                    # the counters' bytes over THIS run's launch time: what the memory system delivered.  traffic > algorithmic here is line
                    # granularity, not re-reads (a 300-byte row lies on three 128-byte lines: hbm_traffic.json line_granular_floor)
                    roofline["traffic_over_algorithmic"] = roofline["traffic"] / fused_bytes
                    roofline["traffic_GB_per_s"] = roofline["traffic"] / tf / 1e9
                    roofline["traffic_frac_of_peak"] = roofline["traffic"] / tf / HBM_PEAK
            except Exception:
                pass
    flops = 2.0 * n_local * (12 * F) * F
    if grouped is not None:
        # one combined block per degree tile: a third of the multiply-adds, and the kernel is no longer matrix-pipe bound -- it
        # streams the aggregate (4 * 4F per row), the residual and y (4F each): priced against HBM, the pipe fractions beside it
        flops_exec = 2.0 * (grouped["padded_rows"] * (4 * F) + (grouped["rest_rows"]) * (12 * F)) * F
        post_bytes = n_local * (16 * F + 4 * F + 4 * F)
        k3 = ("k_posttrans_x3<S=3,NT=5,RT=1,12 wavefronts, 3 weight buffers>" if F <= 80
              else "k_posttrans_x3<S=3,NT=8 (one 128-column block),RT=1,8 wavefronts, 2 weight buffers>")
        roofline_post = {"bound": "hbm", "kernel": ("k_posttrans_x3<S=1,...,GRP> over the degree tiles + k_posttrans_x3<S=3,...,GRP> over the rest "
                                                    "(pna_posttrans_x3_f32, row_perm / tile_image)") if F <= 80 else
                                                   ("k_posttrans_x3<S=1,128-column block,GRP> over the degree tiles (pna_posttrans_x3_f32, row_perm / tile_image)"
                                                    " + the ordinary three-block kernel over the compact rest list and an index scatter"),
                         "achieved": post_bytes / (t_post * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": post_bytes / (t_post * 1e-3) / HBM_PEAK, "ms_per_launch": t_post, "algorithmic_bytes_per_launch": post_bytes,
                         "mfma_frac_executed_flops": flops_exec / (t_post * 1e-3) / (MFMA_BF16_PEAK / 6),
                         "mfma_frac_reference_flops": flops / (t_post * 1e-3) / (MFMA_BF16_PEAK / 6),
                         "fp32_equivalent_tflops_reference_formulation": flops / (t_post * 1e-3) / 1e12,
                         "degree_grouping": grouped,
                         "three_block_kernel": {"kernel": k3, "ms_per_launch": grouped["three_block_contraction_ms"],
                                                "frac": flops / (grouped["three_block_contraction_ms"] * 1e-3) / (MFMA_BF16_PEAK / 6),
                                                "frac_at_sustained_clock": (flops / (grouped["three_block_contraction_ms"] * 1e-3) / (MFMA_BF16_PEAK / 6 * power["contraction"]["sclk_mhz"] / 2400.0)
                                                                            if power and power.get("contraction") else None)},
                         "exact_f32_mfma_kernel": {"kernel": "k_posttrans<3,false,5> (pna_posttrans_f32)", "ms_per_launch": t_post_f32,
                                                   "achieved": flops / (t_post_f32 * 1e-3) / 1e12, "peak": MFMA_F32_PEAK / 1e12,
                                                   "frac": flops / (t_post_f32 * 1e-3) / MFMA_F32_PEAK}}
    elif arith == "bf16x3":     # fp32-equivalent FLOP/s against the bf16 pipe's peak / 6 (six bf16 partial products per multiply)
        roofline_post = {"bound": "mfma", "kernel": "k_posttrans_x3<S=3,NT=5,RT=1,12 wavefronts, 3 weight buffers> (pna_posttrans_x3_f32)",
                         "achieved": flops / (t_post * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK / 6 / 1e12, "unit": "TFLOP/s (fp32-equivalent)",
                         "frac": flops / (t_post * 1e-3) / (MFMA_BF16_PEAK / 6), "ms_per_launch": t_post, "traffic": traffic_post,
                         "traffic_source": traffic_source,
                         "bf16_tflops_issued": 6 * flops / (t_post * 1e-3) / 1e12,
                         "frac_at_sustained_clock": (flops / (t_post * 1e-3) / (MFMA_BF16_PEAK / 6 * power["contraction"]["sclk_mhz"] / 2400.0)
                                                     if power and power.get("contraction") else None),
                         "exact_f32_mfma_kernel": {"kernel": "k_posttrans<3,false,5> (pna_posttrans_f32)", "ms_per_launch": t_post_f32,
                                                   "achieved": flops / (t_post_f32 * 1e-3) / 1e12, "peak": MFMA_F32_PEAK / 1e12,
                                                   "frac": flops / (t_post_f32 * 1e-3) / MFMA_F32_PEAK}}
    else:
        roofline_post = {"bound": "mfma", "kernel": "k_posttrans<3,false,5> (pna_posttrans_f32)", "achieved": flops / (t_post * 1e-3) / 1e12,
                         "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s", "frac": flops / (t_post * 1e-3) / MFMA_F32_PEAK,
                         "ms_per_launch": t_post, "traffic": traffic_post}

    # the LAYER against its own compulsory traffic: gathers + ids + rowptr, residual read, y written -- no 4F aggregate
    layer_bytes = e_local * (4 * F + 4) + 4 * (n_local + 1) + n_local * 4 * F * 2
    roofline_layer = {"bound": "hbm", "kernel": "whole PNASimpleLayer step" + (" (one kernel over the degree groups + the two-kernel rest path)" if fused else " (segreduce + posttrans)"), "algorithmic_bytes_per_step": layer_bytes,
                      "achieved": layer_bytes / (ms_per_step * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                      "frac": layer_bytes / (ms_per_step * 1e-3) / HBM_PEAK,
                      "note": "E(4F+4) + 4(V+1) + 2*V*4F: what a layer that never materialised the 4F aggregate would have to move"}
    wl = ("BASELINE.json configs[2]: synthetic power-law graph |V|=1M |E|=10M per GPU, F=75, " if args.workload == "c3" else
          "BASELINE.json configs[4]: synthetic power-law graph |V|=2M |E|=20M per GPU (16M / 160M over 8 GPUs), F=128, ")
    if halo_rate is not None:
        # whole-job speed-up over one GPU running the same per-GPU shape = world x t_1 / t_world; t_world >= max(compute, exchange at link peak)
        # when the exchange is perfectly hidden (and >= their sum when it is not hidden at all).  compute: this rank's kernels without the exchange.
        t_comp = max(ms_per_step - t_halo, 1e-6)
        ex = halo_rate["exchange_ms_at_link_peak"]
        halo_rate["compute_ms_without_exchange_rank0"] = t_comp
        halo_rate["max_speedup_by_link_bound"] = {"exchange_hidden": world * t_comp / max(t_comp, ex), "exchange_exposed": world * t_comp / (t_comp + ex),
                                                  "note": "weak scaling, against one GPU at this rank's compute time; the Chung-Lu generator has no locality -- "
                                                          "every rank references ~70 % of every peer's rows -- so at 8 GPUs the bound sits below the north "
                                                          "star's 6 x (DESIGN.md 7)"}
    rec = {
        "metric": f"PNA-layer fwd edges/sec (F={F}, 4 aggr x 3 scalers)", "value": value, "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" + (f" (fp32 in / out / accumulation; posttrans contraction: {fused['arith']}, see config.posttrans_arith)" if fused else ""),
        "data": "synthetic",
        "prewarm_steps_untimed": prewarm,
        "config": {"workload": wl + f"single PNA (simple) layer fwd: 4 aggregators x 3 scalers + posttrans Linear({12 * F}->{F}) + BN + ReLU + residual",
                   "V": V, "E": E, "F": F, "aggregators": AGGREGATORS, "scalers": SCALERS,
                   "parallelism": f"dst-range shard x{world}, halo all-to-all" if world > 1 else "single GPU",
                   "x_row_pitch_floats": max(args.x_pitch, F),
                   # (the arithmetic of the path the timed step TOOK: the one-kernel layer's for its group rows, the two-kernel path's otherwise)
                   "posttrans_arith": (ARITH_TEXT[fused["arith"]] + " -- the group rows of the one-kernel layer; the rest rows (two-kernel path): bf16x3") if fused
                                      else ARITH_TEXT["bf16x3"] if arith == "bf16x3" else "f32 (v_mfma_f32_16x16x4_f32)",
                   "guard": guard_info,
                   "max_in_degree": int(csr.max_degree), "halo_rows_rank0": getattr(g, "n_halo", 0),
                   "interior_rows_rank0": int(g.interior_mask().sum().item()) if world > 1 else None,
                   "local_rows_rank0": n_local, "local_edges_rank0": e_local, "partition_balance": args.balance if world > 1 else None,
                   "halo_exchange_in_step": ("every timed step runs the halo all-to-all of the source rows (pack + all_to_all_single over RCCL) before its "
                                             "kernel, although this benchmark's features do not change between steps: it is the cost a multi-layer net "
                                             "pays once per layer; nothing is prefetched or reused across steps") if world > 1 else None},
        "roofline": roofline, "roofline_segreduce_standalone": roofline_segreduce, "roofline_posttrans": roofline_post, "roofline_layer": roofline_layer,
        "power_probe": power,
        "ms_per_step_cold": ms_per_step_cold, "value_cold": (E / (ms_per_step_cold * 1e-3)) if ms_per_step_cold else None,
        "parity_check": check, "parity_check_all_ranks": check_ranks,
        "kernel_ms": {"fused_degree_group_rows": fused["ms_group_rows_kernel"] if fused else None,
                      "fused_degree_rest_rows": fused["ms_rest_rows_two_kernel_path"] if fused else None,
                      "segreduce": t_seg, "posttrans": t_post, "posttrans_exact_f32_mfma": t_post_f32, "halo_all_to_all": t_halo,
                      "csr_build_once_per_graph": csr_build_ms},
        "halo_exchange": halo_rate,
        "ms_per_step_exact_f32_mfma": ms_per_step_f32, "value_exact_f32_mfma": E / (ms_per_step_f32 * 1e-3),
        "ms_per_step_bf16x3": arith_legs.get("bf16x3"), "value_bf16x3": (E / (arith_legs["bf16x3"] * 1e-3)) if arith_legs.get("bf16x3") else None,
        "ms_per_step_fp16x2_unguarded": arith_legs.get("fp16x2"),
        "ms_per_step_contiguous_input": ms_per_step_contig,
        "value_contiguous_input": (E / (ms_per_step_contig * 1e-3)) if ms_per_step_contig else None,
        "per_graph_setup": setup,
        "ms_per_step_hipgraph_replay": ms_per_step_hipgraph,
        "contiguous_input": ({"what": "the same K steps with h a CONTIGUOUS (V, F) tensor (row pitch F floats = 300 bytes at F = 75): the reference API's input",
                              "takes_the_one_kernel_layer": contig_one_kernel, "output_bits_equal_the_aligned_step": contig_same_bits,
                              "value": E / (ms_per_step_contig * 1e-3), "ratio_to_headline": ms_per_step_contig / ms_per_step}
                             if ms_per_step_contig else None),
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            rec["cpu_baseline"] = cpu_baseline(src, dst, V, h_all, layer_sd, avg_log, args.cpu_sample_rows)
        except Exception as ex:                                     # the baseline must never sink the GPU number
            rec["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"failed: {ex!r}"}
    rpath = os.path.join(ROOT, "profiles", "cpu_reference_source.json")
    if rank == 0 and world == 1 and os.path.exists(rpath):
        try:      # the reference's OWN source (DGL stand-in) cannot travel to the GPU box: timed once in the build container, recorded
            rec["cpu_baseline_reference_source"] = json.load(open(rpath))
        except Exception:
            pass
    if (rank == 0 and world == 1 and args.workload == "c3" and not args.no_c5_leg and args.layers == 1
            and args.nodes_per_gpu == V_PER_GPU and args.edges_per_gpu == E_PER_GPU):
        # BASELINE configs[4] (V = 16 M, E = 160 M, F = 128 over 8 GPUs) is the driver's to launch; its PER-GPU shape on this one GPU
        # rides along with every default call, so that the number is driver-run too
        rec["configs4_per_gpu_shape"] = other_workload_leg(["--workload", "c5", "--no-cpu-baseline", "--no-cold", "--no-power-probe",
                                                            "--steps", "10", "--warmup", "3"])
    if rank == 0:
        rec["diagnostics_error"] = diag_error
        print(json.dumps(_finite(rec)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
