export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -25
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"frac": [0-9.]*' gpurun_out/bench.log
timeout 600 python tools/sweep.py --tag r01g --rounds 3 > gpurun_out/sweep.log 2>&1; echo "sweep rc=$?"
head -30 gpurun_out/sweep.log | cut -c1-200
