#!/bin/bash
# the round's last code: GPU suite, smoke, the bench line
mkdir -p gpurun_out/r05last
timeout 500 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/r05last/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r05last/pytest_gpu.log | tail -5
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r05last/bench_n1.json 2> gpurun_out/r05last/bench_n1.err; echo "bench rc=$?"
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*' gpurun_out/r05last/bench_n1.json | head -6
