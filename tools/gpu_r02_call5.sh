#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/pytest_gpu.log | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print({k:r[k] for k in ("value","ms_per_step","ms_per_step_cold","ms_per_step_exact_f32_mfma","kernel_ms","parity_check")})
print(r["roofline"]["frac"], r["roofline"]["read_only_frac"], r["roofline_posttrans"]["frac"], r["roofline_layer"]["frac"], r.get("cpu_baseline"))
PY
timeout 900 python bench.py --workload c5 --no-cpu-baseline > gpurun_out/bench_c5.log 2>gpurun_out/bench_c5.err; echo "bench c5 rc=$?"; tail -3 gpurun_out/bench_c5.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/bench_c5.log').read().strip().splitlines()[-1])
print({k:r[k] for k in ("value","ms_per_step","ms_per_step_cold","kernel_ms","parity_check")}); print(r["roofline"]["frac"], r["roofline"]["read_only_frac"], r["roofline_posttrans"]["frac"])
PY
PNA_BENCH_ONE_DEVICE=1 PNA_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 3 --warmup 1 --nodes-per-gpu 200000 --edges-per-gpu 2000000 --no-cpu-baseline > gpurun_out/bench_n2_smoke.log 2>gpurun_out/bench_n2_smoke.err; echo "bench N=2 smoke rc=$?"
grep -o '"n_gpus": [0-9]*\|"value": [0-9.e+]*\|"halo_rows_rank0": [0-9]*\|"interior_rows_rank0": [0-9]*\|"halo_all_to_all": [0-9.]*\|"parity_check": {[^}]*}' gpurun_out/bench_n2_smoke.log | tr '\n' ' '; echo; tail -3 gpurun_out/bench_n2_smoke.err
timeout 900 python tools/bench_configs.py > gpurun_out/configs_r02.json 2>gpurun_out/configs_r02.err; echo "configs rc=$?"; tail -2 gpurun_out/configs_r02.err; python -c "
import json; r=json.load(open('gpurun_out/configs_r02.json')); print({k:(v if not isinstance(v,dict) else {a:b for a,b in v.items() if 'ms' in a or 'err' in a}) for k,v in r.items()})"
