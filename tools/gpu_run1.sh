export TMPDIR=/tmp
mkdir -p gpurun_out
rocminfo | grep -m1 gfx || true
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/smoke.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/bench.log
timeout 600 python tools/sweep.py --tag r01a > gpurun_out/sweep.log 2>&1; echo "sweep rc=$?" | tee -a gpurun_out/sweep.log
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log; tail -20 gpurun_out/sweep.log
