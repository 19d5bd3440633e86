#!/bin/bash
# One GPU call that re-checks everything the round-end driver runs: GPU test suite, smoke(), the bench line.
#   gpurun --timeout 2400 -- 'bash tools/gpu_check.sh'
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench rc=$?"
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"frac": [0-9.]*\|"cpu_baseline": {[^}]*}' gpurun_out/bench.log
