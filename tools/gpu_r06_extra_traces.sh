#!/bin/bash
# rocprofv3 kernel traces of the round-6 paths outside the bench line: the training step and the five-tower layer (one GPU call).
#   gpurun --timeout 900 -- 'bash tools/gpu_r06_extra_traces.sh'
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
O=$P/gpurun_out/r06x
rm -rf $O; mkdir -p $O
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t1 -o train -- python $P/tools/bench_train.py > $O/train.log 2>&1; echo "train trace rc=$?"
find $O/t1 -name "train_kernel_stats.csv" -exec cp {} $O/train_kernel_stats.csv \;
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t2 -o mt -- python $P/tools/multi_tower_time.py > $O/mt.log 2>&1; echo "multi-tower trace rc=$?"
find $O/t2 -name "mt_kernel_stats.csv" -exec cp {} $O/multi_tower_kernel_stats.csv \;
rm -rf $O/t1 $O/t2
head -12 $O/train_kernel_stats.csv | cut -c1-160
head -10 $O/multi_tower_kernel_stats.csv | cut -c1-160
