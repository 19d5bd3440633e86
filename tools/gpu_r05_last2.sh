#!/bin/bash
# the round's last code: the bench line's rocprofv3 kernel trace and the C5-shape line (the bench line itself: tools/gpu_r05_last.sh)
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
O=$P/gpurun_out/r05last
mkdir -p $O
cd $P
timeout 300 python bench.py --workload c5 --no-cpu-baseline --no-cold > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
cd /tmp; rm -rf $O/trace
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $P/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold --no-power-probe --no-c5-leg > $O/trace.log 2>&1; echo "trace rc=$?"
find $O/trace -name "bench_kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
head -3 $O/bench_kernel_stats.csv | cut -c1-200
rm -rf $O/trace
grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $O/bench_c5.json | head -3
