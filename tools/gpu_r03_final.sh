#!/bin/bash
# Round-3 measurement set: bench line, its rocprofv3 kernel trace, C5 shard shape, training step, the other BASELINE configs.
#   gpurun --timeout 3000 -- 'bash tools/gpu_r03_final.sh'
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
O=$P/gpurun_out/r03
mkdir -p $O
cd $P
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
timeout 600 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
timeout 300 python bench.py --layers 4 --steps 10 --warmup 3 > $O/bench_layers4.json 2> $O/bench_layers4.err; echo "layers rc=$?"
timeout 600 python tools/bench_train.py > $O/train_step.json 2> $O/train_step.err; echo "train rc=$?"
timeout 900 python tools/bench_configs.py > $O/configs.json 2> $O/configs.err; echo "configs rc=$?"
timeout 300 python tools/tower_time.py $O/tower_time.json > $O/tower_time.log 2>&1; echo "tower rc=$?"
cd /tmp; rm -rf $O/trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $P/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold --no-power-probe > $O/trace.log 2>&1; echo "trace rc=$?"
cd $P
find $O/trace -name "bench_kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
head -6 $O/bench_kernel_stats.csv | cut -c1-200
rm -rf $O/trace
tail -c 300 $O/configs.err
