#!/bin/bash
# rocprofv3 kernel traces of the round-3 paths beside the bench line: training step, C3 tower layer, C5 shape.
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
O=$P/gpurun_out/r03
mkdir -p $O
cd /tmp
for job in "train:tools/bench_train.py" "tower:tools/tower_time.py" "c5:bench.py --workload c5 --steps 20 --warmup 5 --no-cpu-baseline --no-cold --no-power-probe"; do
  name=${job%%:*}; cmd=${job#*:}
  rm -rf $O/trace_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o t -- python $P/$cmd > $O/trace_$name.log 2>&1; echo "$name rc=$?"
  find $O/trace_$name -name "t_kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  rm -rf $O/trace_$name
  head -7 $O/${name}_kernel_stats.csv | cut -c1-170
done
