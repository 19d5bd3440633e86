#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_tower_fused.py tests/test_gpu_layers.py tests/test_gpu_kernels.py -m gpu -q --timeout 600 > gpurun_out/pytest_tf.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" gpurun_out/pytest_tf.log | tail -30
timeout 600 python tools/exp_small_tower.py > gpurun_out/small_tower.json 2>gpurun_out/small_tower.err; echo "exp rc=$?"; grep "^{" gpurun_out/small_tower.err | cut -c1-400; tail -3 gpurun_out/small_tower.err | cut -c1-300
