#!/bin/bash
# One GPU call that re-checks everything the round-end driver runs: GPU test suite, smoke(), the bench line.
#   gpurun --timeout 2400 -- 'bash tools/gpu_check.sh'
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench rc=$?"
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"frac": [0-9.]*\|"cpu_baseline": {[^}]*}' gpurun_out/bench.log
# the N=2 code path (shard, halo exchange into the resident table, max-over-ranks timing) on this one GPU over gloo
PNA_BENCH_ONE_DEVICE=1 PNA_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 3 --warmup 1 --nodes-per-gpu 200000 --edges-per-gpu 2000000 --no-cpu-baseline > gpurun_out/bench_n2_smoke.log 2>gpurun_out/bench_n2_smoke.err; echo "bench N=2 smoke rc=$?"
grep -o '"n_gpus": [0-9]*\|"value": [0-9.e+]*\|"halo_rows_rank0": [0-9]*\|"halo_all_to_all": [0-9.]*' gpurun_out/bench_n2_smoke.log | tr '\n' ' '; echo
