export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
cd /tmp; rm -rf $P/gpurun_out/dg_trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gpurun_out/dg_trace -o bench -- python $P/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold --no-power-probe > $P/gpurun_out/dg_trace.log 2>&1; echo rc=$?
head -8 $P/gpurun_out/dg_trace/bench_kernel_stats.csv | cut -c1-160
