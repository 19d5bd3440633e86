export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -25
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -2 gpurun_out/bench.log | cut -c1-2500
timeout 600 python tools/sweep.py --tag r01f --rounds 3 > gpurun_out/sweep.log 2>&1; echo "sweep rc=$?"
head -16 gpurun_out/sweep.log | cut -c1-180
