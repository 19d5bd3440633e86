#!/bin/bash
# Round-2 closing pass: the fixed test file, per-config timings, the rocprofv3 kernel trace of the bench command, the C5 shape.
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
mkdir -p $P/gpurun_out; rm -rf $P/gpurun_out/r02f_trace
cd $P
timeout 600 python -m pytest tests/test_gpu_tower_fused.py -q 2>&1 | tail -2
timeout 900 python tools/bench_configs.py > gpurun_out/r02f_configs.json 2>gpurun_out/r02f_configs.err; echo "configs rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02f_bench.log 2>gpurun_out/r02f_bench.err; echo "bench rc=$?"
timeout 600 python bench.py --workload c5 --no-cpu-baseline > gpurun_out/r02f_bench_c5.log 2>&1; echo "c5 rc=$?"
timeout 300 python tools/bench_train.py > gpurun_out/r02f_train.json 2>/dev/null; echo "train rc=$?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gpurun_out/r02f_trace -o bench -- python $P/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold > $P/gpurun_out/r02f_trace.log 2>&1; echo "trace rc=$?"
ls $P/gpurun_out/r02f_trace | head
