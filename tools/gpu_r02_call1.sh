#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|Error|error" gpurun_out/pytest_gpu.log | tail -15
timeout 900 python tools/exp_r02.py posttrans tower c5 > gpurun_out/exp1.log 2>&1; echo "exp1 rc=$?"; grep -E "^(posttrans|tower|c5)" gpurun_out/exp1.log | cut -c1-900
timeout 900 python tools/exp_r02.py occupancy > gpurun_out/exp_occ.log 2>&1; echo "occ rc=$?"; grep "^occ" gpurun_out/exp_occ.log | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench rc=$?"
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"frac": [0-9.]*' gpurun_out/bench.log
