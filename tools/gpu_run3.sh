export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -25
timeout 600 python tools/sweep.py --tag r01b > gpurun_out/sweep.log 2>&1; echo "sweep rc=$?"
cat gpurun_out/sweep.log | tail -70
