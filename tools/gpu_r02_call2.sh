#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/pytest_gpu.log | tail -15
timeout 900 python tools/exp_r02.py posttrans > gpurun_out/exp2.log 2>&1; echo "exp2 rc=$?"; grep -E "^(posttrans)" gpurun_out/exp2.log | cut -c1-900; tail -3 gpurun_out/exp2.log | cut -c1-600
