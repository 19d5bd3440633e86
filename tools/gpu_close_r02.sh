#!/bin/bash
# Closing call of round 2: everything the round-end driver runs (tools/gpu_check.sh), then the rocprofv3 kernel trace of the bench
# command (summary -> profiles/r02_bench_kernel_stats.csv).
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
cd $P
bash tools/gpu_check.sh
cp gpurun_out/bench.log gpurun_out/bench_closing.json
rm -rf $P/gpurun_out/r02c_trace
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gpurun_out/r02c_trace -o bench -- python $P/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold --no-power-probe > $P/gpurun_out/r02c_trace.log 2>&1; echo "trace rc=$?"
ls $P/gpurun_out/r02c_trace | head -5
