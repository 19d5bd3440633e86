export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -5
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"frac": [0-9.]*' gpurun_out/bench.log
timeout 900 python tools/bench_configs.py > gpurun_out/configs.json 2> gpurun_out/configs.err; echo "configs rc=$?"; cat gpurun_out/configs.json; tail -5 gpurun_out/configs.err
