#!/bin/bash
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_tower_fused.py tests/test_gpu_layers.py -m gpu -q --timeout 600 > gpurun_out/pytest_tf.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" gpurun_out/pytest_tf.log | tail -30
timeout 120 python tools/tf_timers.py > gpurun_out/tf_timers.txt 2>&1; echo "timers rc=$?"; grep -v amdgpu.ids gpurun_out/tf_timers.txt | cut -c1-200
timeout 600 python tools/exp_small_tower.py 128 512 > gpurun_out/small_tower.json 2>gpurun_out/small_tower.err; echo "exp rc=$?"; grep "^{" gpurun_out/small_tower.err | cut -c1-330
cd /tmp
rm -rf $P/gpurun_out/zinc_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gpurun_out/zinc_stats -o z -- python $P/tools/prof_zinc.py > $P/gpurun_out/zinc_stats.log 2>&1; echo "rocprof rc=$?"
f=$(find $P/gpurun_out/zinc_stats -name "*kernel_stats.csv" | head -1); head -4 $f | cut -c1-200
