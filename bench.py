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
  roofline     -- the fused segment-reduce kernel vs the 8 TB/s HBM roofline, ALGORITHMIC bytes
                  E*(4F+4) + 4(V+1) + V*16F per launch / its measured mean duration
  cpu_baseline -- the oracle's C port (oracle/pna_oracle.c, OpenMP) + torch CPU Linear/BN of the same
                  layer, timed on this box's host cores on a bounded sample (rank 0, N=1 only)
"""
import argparse
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
    return p.parse_args()


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


def main():
    args = parse()
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
        g = shard_graph(src, dst, V)
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
    # node features live in a 16-byte aligned row pitch (80 floats for F=75), the layout a multi-layer net keeps
    # its activations in; the kernels accept any pitch (--x-pitch 75 = dense rows, ~3 % slower gather)
    if world > 1:
        # multi-GPU: the features live in the shard's resident [local | halo] table, so the halo exchange of the
        # timed step receives the peers' rows in place (no concatenation pass)
        h = g.alloc_features(F, pitch=max(args.x_pitch, F), device=dev)
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

    def step():
        with torch.no_grad():
            return layer(g, h)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = E / (dt / args.steps)

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

    # ---- per-kernel timing of the dominant kernels (HIP events, this rank) ------------------------------
    csr = g.csr
    with torch.no_grad():
        x_ext = g.source_features(h)
        t_seg = event_time_ms(lambda: PF.aggregate(g, x_ext, F, AGGREGATORS.split()), args.kernel_iters)
        agg = PF.aggregate(g, x_ext, F, AGGREGATORS.split())
        lin = layer.posttrans.fully_connected[0].linear
        amp, att = g.degree_scalers(float(avg_log))
        t_post = event_time_ms(lambda: PF.posttrans(agg, 4 * F, lin.weight, lin.bias, [None, amp, att]), args.kernel_iters)
        t_post_f32 = event_time_ms(lambda: _ops.posttrans(agg, 4 * F, lin.weight, [None, amp, att], lin.bias, arith="f32"), args.kernel_iters)
        t_halo = event_time_ms(lambda: g.source_features(h), args.kernel_iters) if world > 1 else 0.0
    alg_read = e_local * (4 * F + 4) + 4 * (n_local + 1)
    alg_write = n_local * 16 * F
    alg_bytes = alg_read + alg_write
    traffic = traffic_post = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")      # PMC-derived HBM bytes per launch, if collected
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get("pna_segreduce_c3", {}).get("hbm_bytes_per_launch")
            traffic_post = tj.get("pna_posttrans_x3_c3" if arith == "bf16x3" else "pna_posttrans_f32_c3", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = traffic_post = None
    roofline = {"bound": "hbm", "kernel": "k_segreduce_fast<4> + k_heavy_finalize (pna_segreduce_fwd_f32)",
                "achieved": alg_bytes / (t_seg * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": alg_bytes / (t_seg * 1e-3) / HBM_PEAK, "traffic": traffic,
                "ms_per_launch": t_seg, "algorithmic_bytes_per_launch": alg_bytes,
                "read_only_frac": alg_read / (t_seg * 1e-3) / HBM_PEAK,
                "edges_per_s_kernel_only": e_local / (t_seg * 1e-3),
                "heavy_rows": hs.n_heavy, "heavy_segments": hs.n_seg}
    flops = 2.0 * n_local * (12 * F) * F
    if arith == "bf16x3":     # fp32-equivalent FLOP/s against the bf16 pipe's peak / 6 (six bf16 partial products per multiply)
        roofline_post = {"bound": "mfma", "kernel": "k_posttrans_x3<3,false,5,1,12> (pna_posttrans_x3_f32)",
                         "achieved": flops / (t_post * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK / 6 / 1e12, "unit": "TFLOP/s (fp32-equivalent)",
                         "frac": flops / (t_post * 1e-3) / (MFMA_BF16_PEAK / 6), "ms_per_launch": t_post, "traffic": traffic_post,
                         "bf16_tflops_issued": 6 * flops / (t_post * 1e-3) / 1e12,
                         "exact_f32_mfma_kernel": {"kernel": "k_posttrans<3,false,5> (pna_posttrans_f32)", "ms_per_launch": t_post_f32,
                                                   "achieved": flops / (t_post_f32 * 1e-3) / 1e12, "peak": MFMA_F32_PEAK / 1e12,
                                                   "frac": flops / (t_post_f32 * 1e-3) / MFMA_F32_PEAK}}
    else:
        roofline_post = {"bound": "mfma", "kernel": "k_posttrans<3,false,5> (pna_posttrans_f32)", "achieved": flops / (t_post * 1e-3) / 1e12,
                         "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s", "frac": flops / (t_post * 1e-3) / MFMA_F32_PEAK,
                         "ms_per_launch": t_post, "traffic": traffic_post}

    rec = {
        "metric": "PNA-layer fwd edges/sec (F=75, 4 aggr x 3 scalers)", "value": value, "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[2]: synthetic power-law graph |V|=1M |E|=10M per GPU, F=75, "
                               "single PNA (simple) layer fwd: 4 aggregators x 3 scalers + posttrans Linear(900->75) + BN + ReLU + residual",
                   "V": V, "E": E, "F": F, "aggregators": AGGREGATORS, "scalers": SCALERS,
                   "parallelism": f"dst-range shard x{world}, halo all-to-all" if world > 1 else "single GPU",
                   "x_row_pitch_floats": max(args.x_pitch, F),
                   "posttrans_arith": ("bf16x3: fp32 in/out; each fp32 operand cut exactly into 3 bf16 terms, 6 partial products per multiply "
                                       "on the bf16 MFMA pipe, fp32 accumulate; error vs float64 at the exact-f32 kernel's level "
                                       "(tests/test_gpu_posttrans_x3.py)") if arith == "bf16x3" else "f32 (v_mfma_f32_16x16x4_f32)",
                   "max_in_degree": int(csr.max_degree), "halo_rows_rank0": getattr(g, "n_halo", 0)},
        "roofline": roofline, "roofline_posttrans": roofline_post,
        "kernel_ms": {"segreduce": t_seg, "posttrans": t_post, "posttrans_exact_f32_mfma": t_post_f32, "halo_all_to_all": t_halo,
                      "csr_build_once_per_graph": csr_build_ms},
        "ms_per_step_exact_f32_mfma": ms_per_step_f32, "value_exact_f32_mfma": E / (ms_per_step_f32 * 1e-3),
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            rec["cpu_baseline"] = cpu_baseline(src, dst, V, h_all, layer_sd, avg_log, args.cpu_sample_rows)
        except Exception as ex:                                     # the baseline must never sink the GPU number
            rec["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"failed: {ex!r}"}
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
