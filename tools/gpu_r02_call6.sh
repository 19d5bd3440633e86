#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_posttrans_x3.py tests/test_gpu_layers.py -m gpu -q --timeout 600 > gpurun_out/pytest_x3.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/pytest_x3.log | tail -8
timeout 600 python tools/exp_r02.py c5 > gpurun_out/exp6.log 2>&1; echo "exp6 rc=$?"; grep -E "^c5" gpurun_out/exp6.log | cut -c1-900
timeout 600 python bench.py --workload c5 --no-cpu-baseline > gpurun_out/bench_c5.log 2>gpurun_out/bench_c5.err; echo "bench c5 rc=$?"
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"frac": [0-9.]*' gpurun_out/bench_c5.log
timeout 900 python tools/bench_configs.py > gpurun_out/configs.json 2>gpurun_out/configs.err; echo "configs rc=$?"; tail -2 gpurun_out/configs.err
grep -E "batch_build|eager_ms|hipgraph_ms|layer_ms" gpurun_out/configs.json
